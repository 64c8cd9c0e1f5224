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
def lin(M, K, N, act=0, res=False):
    a = torch.randn(M, K, device=dev).to(ops.act_dtype()); w = (torch.randn(N, K, device=dev) * 0.02).to(ops.act_dtype()); b = torch.zeros(N, device=dev)
    r = torch.randn(M, N if act != 2 else N // 2, device=dev).to(ops.act_dtype()) if res else None
    bench(lambda: ops.linear(a, w, b, act=act, residual=r), 2 * M * N * K, f"linear M{M} K{K} N{N} act{act} res{int(res)}")
lin(114688, 320, 320); lin(114688, 320, 320, res=True); lin(114688, 320, 2560, act=2); lin(28672, 640, 640); lin(114688, 320, 960)
