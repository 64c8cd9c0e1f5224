"""Time ops.attention at the window's self-attention sizes (GPU box).  python tools/attn_bench.py [--fp8]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vidseg_diffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for B, H, N in ((28, 5, 4096), (28, 10, 1024), (28, 20, 256)):
    C = H * 64
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn((B, N, 3 * C), generator=g).to(ops.act_dtype()).to(dev)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    for _ in range(3):
        o = ops.attention(q, k, v, H)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        o = ops.attention(q, k, v, H)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 4.0 * B * H * N * N * 64
    print(f"attention B={B} H={H} N={N}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  (TR={os.environ.get('VIDSEG_ATTN_TR', '1')})", flush=True)
