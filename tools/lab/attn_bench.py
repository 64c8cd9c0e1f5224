"""Time ops.attention at the window's self-attention sizes (GPU box).  python tools/lab/attn_bench.py [--fp8]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    print(f"attention B={B} H={H} N={N}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  (TR={os.environ.get('VIDSEG_ATTN', 'default')})", flush=True)

# experiment builds with -DVS_ATTN_STAMPS (tools/build_exp.py): phase timeline of tiles 8..11 of one block, per wave
import ctypes  # noqa: E402
from vidseg_diffusion_amd import _lib  # noqa: E402
L = _lib.lib()
if hasattr(L, "vidseg_debug_attn_stamps"):
    B, H, N = 28, 5, 4096
    qkv = torch.randn((B, N, 3 * H * 64)).to(ops.act_dtype()).to(dev)
    ops.attention(qkv[..., :320], qkv[..., 320:640], qkv[..., 640:], H)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 256)()
    L.vidseg_debug_attn_stamps(buf)
    names = ["X0 S(t,1)|P(A)", "fix+Y0 PV(A)", "X1 S(t+1,0)|P(B)", "fix+Y1 PV(B)", "stores", "barrier", "loop edge"]
    for w in range(4):
        for t in range(4):
            st = [buf[(w * 4 + t) * 8 + i] for i in range(7)]
            nxt = buf[(w * 4 + t + 1) * 8] if t < 3 else None
            d = [st[i + 1] - st[i] for i in range(6)] + ([nxt - st[6]] if nxt else [])
            print(f"wave {w} tile {8 + t}: " + "  ".join(f"{n} {v}" for n, v in zip(names, d)) + f"  | total {sum(d[:6])}")
