import sys; sys.path.insert(0, '.')
import torch
from vidseg_diffusion_amd import ops
dev = torch.device('cuda:0')
def bench(fn, flops, name, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"{name:40s} {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TF/s")
B = 28
for (N, H, Nk) in [(4096, 5, 4096), (1024, 10, 1024), (256, 20, 256), (64, 20, 64), (4096, 5, 77), (1024, 10, 77)]:
    C = H * 64
    qkv = torch.randn(B, N, 3 * C, device=dev).to(ops.act_dtype())
    kv = torch.randn(B, Nk, 2 * C, device=dev).to(ops.act_dtype())
    if Nk == N:
        bench(lambda: ops.attention(qkv[..., :C], qkv[..., C:2*C], qkv[..., 2*C:], H), 4 * B * H * N * Nk * 64, f"self N{N} H{H}")
    else:
        q = qkv[..., :C].contiguous()
        bench(lambda: ops.attention(q, kv[..., :C], kv[..., C:], H), 4 * B * H * N * Nk * 64, f"cross N{N} H{H} Nk{Nk}")
