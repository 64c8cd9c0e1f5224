import sys
sys.path.insert(0, ".")
import torch
from vidseg_diffusion_amd import ops
dev = torch.device("cuda:0")
for (M, N, K) in [(114688, 2560, 320), (28672, 5120, 640), (7168, 10240, 1280)]:
    a = torch.randn(M, K, device=dev).to(ops.act_dtype())
    w = (torch.randn(N, K, device=dev) * 0.05).to(ops.act_dtype())
    b = torch.randn(N, device=dev) * 0.1
    f = lambda: ops.linear(a, w, b, act=2)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        f()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print(f"GEGLU M{M} N{N} K{K}: {us:7.1f} us {2.0*M*N*K/us/1e6:6.0f} TF/s checksum {f().float().abs().mean().item():.6f}", flush=True)
