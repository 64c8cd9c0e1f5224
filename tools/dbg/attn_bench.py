"""Time the attention kernels (16-bit vs fp8) on the UNet's shapes."""
import sys
sys.path.insert(0, ".")
import torch
from vidseg_diffusion_amd import ops
dev = torch.device("cuda:0")
for (B, H, Nq, Nk) in [(28, 5, 4096, 4096), (28, 10, 1024, 1024), (28, 20, 256, 256), (28, 5, 4096, 77), (28, 10, 1024, 77), (28, 5, 9216, 9216)]:
    C = H * 64
    if Nq == Nk:
        qkv = torch.randn((B, Nq, 3 * C), device=dev).to(ops.act_dtype())
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    else:
        q = torch.randn((B, Nq, C), device=dev).to(ops.act_dtype())
        kv = torch.randn((B, Nk, 2 * C), device=dev).to(ops.act_dtype())
        k, v = kv[..., :C], kv[..., C:]
    for fp8 in (False, True):
        f = lambda: ops.attention(q, k, v, H, fp8=fp8)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            f()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        print(f"B{B} H{H} Nq{Nq} Nk{Nk} fp8={fp8}: {ms*1e3:8.1f} us  {4.0*B*H*Nq*Nk*64/ms/1e9:7.1f} TF/s (incl. quantisation)", flush=True)
