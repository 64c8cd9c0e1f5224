import sys, time; sys.path.insert(0, '.')
import torch
from vidseg_diffusion_amd import ops
dev = torch.device('cuda:0')
def bench(fn, flops, name, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"{name:44s} {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TF/s")
B = 28
def conv(H, Cin, Cout, C1=0, stride=1, up=1):
    x0 = torch.randn(B, H, H, Cin, device=dev).to(ops.act_dtype()); x1 = torch.randn(B, H, H, C1, device=dev).to(ops.act_dtype()) if C1 else None
    w = (torch.randn(Cout, 9 * (Cin + C1), device=dev) * 0.02).to(ops.act_dtype()); b = torch.zeros(Cout, device=dev)
    Ho = H * up // stride
    bench(lambda: ops.conv3x3(x0, w, b, x1=x1, stride=stride, up=up), 2 * B * Ho * Ho * Cout * 9 * (Cin + C1), f"conv H{H} {Cin}+{C1}->{Cout} s{stride} u{up}")
def lin(M, K, N, act=0):
    a = torch.randn(M, K, device=dev).to(ops.act_dtype()); w = (torch.randn(N, K, device=dev) * 0.02).to(ops.act_dtype()); b = torch.zeros(N, device=dev)
    bench(lambda: ops.linear(a, w, b, act=act), 2 * M * N * K, f"linear M{M} K{K} N{N} act{act}")
conv(64, 320, 320); conv(64, 640, 320, 320); conv(64, 320, 320, 320); conv(64, 320, 640, stride=2) if False else None
conv(32, 640, 640); conv(32, 1280, 640, 640); conv(32, 640, 640, 320)
conv(16, 1280, 1280); conv(16, 1280, 1280, 1280); conv(16, 1280, 1280, 640)
conv(8, 1280, 1280); conv(8, 1280, 1280, 1280)
conv(32, 640, 640, up=2); conv(16, 1280, 1280, up=2)
lin(114688, 320, 320); lin(114688, 320, 960); lin(114688, 320, 2560, act=2); lin(114688, 1280, 320)
lin(28672, 640, 640); lin(28672, 640, 1920); lin(28672, 640, 5120, act=2); lin(28672, 2560, 640)
lin(7168, 1280, 1280); lin(7168, 1280, 3840); lin(7168, 1280, 10240, act=2); lin(7168, 5120, 1280)
lin(1792, 1280, 1280); lin(1792, 1280, 10240, act=2); lin(1792, 5120, 1280); lin(2156, 1024, 1280)
