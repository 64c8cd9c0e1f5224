import sys; sys.path.insert(0, '.')
import torch
from vidseg_diffusion_amd import ops
dev = torch.device('cuda:0')
def bench(fn, flops, name, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"{name:44s} {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TF/s")
def lin(M, K, N, act=0):
    a = torch.randn(M, K, device=dev).to(ops.act_dtype()); w = (torch.randn(N, K, device=dev) * 0.02).to(ops.act_dtype()); b = torch.zeros(N, device=dev)
    bench(lambda: ops.linear(a, w, b, act=act), 2 * M * N * K, f"linear M{M} K{K} N{N} act{act}")
lin(7168, 1280, 1280); lin(7168, 1280, 3840); lin(7168, 1280, 10240, act=2); lin(7168, 5120, 1280); lin(28672, 640, 5120, act=2); lin(114688, 1280, 320)
lin(1792, 1280, 1280); lin(1792, 1280, 10240, act=2); lin(2156, 1024, 2560); lin(28672, 2560, 640)
