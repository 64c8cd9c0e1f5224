"""Exact-mode attention at the window's sizes: k_x_attention_mfma (split operands on the matrix pipe) vs k_x_attention_f32
(GPU box): python tools/lab/x_attn_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vidseg_diffusion_amd import exact as X  # noqa: E402

dev = torch.device("cuda:0")


def bench(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


tot = {"mfma": 0.0, "f32": 0.0}
# (B, heads, Nq, Nk, launches per UNet evaluation): SD 2.1 at 64x64 latents, CFG batch 28
for B, H, Nq, Nk, n in ((28, 5, 4096, 4096, 5), (28, 10, 1024, 1024, 5), (28, 20, 256, 256, 5), (28, 20, 64, 64, 1),
                        (28, 5, 4096, 77, 5), (28, 10, 1024, 77, 5), (28, 20, 256, 77, 5), (28, 20, 64, 77, 1)):
    C = H * 64
    q = torch.randn(B, Nq, C, device=dev)
    kv = torch.randn(B, Nk, 2 * C, device=dev)
    flops = 4.0 * B * H * Nq * Nk * 64
    line = f"B={B} H={H} Nq={Nq} Nk={Nk}:"
    for name, fn in (("mfma", lambda: X.attention_mfma(q, kv, H, B, Nq, Nk)), ("f32", lambda: X.attention_f32(q, kv[..., :C], kv[..., C:], H, B, Nq, Nk))):
        if name == "mfma" and Nq < 128:
            fn = lambda: X.attention_f32(q, kv[..., :C], kv[..., C:], H, B, Nq, Nk)   # noqa: E731
        us = bench(fn)
        tot[name] += us * n * 3
        line += f"  {name} {us:9.1f} us ({flops / us / 1e6:6.1f} TF/s fp32-equivalent)"
    us = bench(lambda: X.split_planes(kv))
    line += f"  [split_planes {us:.1f} us, inside mfma]"
    print(line, flush=True)
print(f"per window (3 evaluations): mfma {tot['mfma'] / 1e3:.1f} ms, f32 {tot['f32'] / 1e3:.1f} ms")
