"""Fixed cost per round of the big GEMM tiles: time linears of growing K on a shape that is exactly one / two rounds of 256 tiles and
fit t = a + b K (GPU box).  python tools/lab/ksweep.py   (VIDSEG_GEMM=big=2,p7=2 forces the 224x320 tile)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vidseg_diffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def bench(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


for (M, N, res) in ((28672, 640, False), (28672, 640, True), (114688, 320, False), (57344, 1280, False)):
    ks, ts = [], []
    for K in (64, 128, 320, 640, 1280, 2560, 5120):
        a = torch.randn(M, K, device=dev).to(ops.act_dtype())
        w = (torch.randn(N, K, device=dev) * 0.02).to(ops.act_dtype())
        b = torch.zeros(N, device=dev)
        r = torch.randn(M, N, device=dev).to(ops.act_dtype()) if res else None
        us = bench(lambda: ops.linear(a, w, b, residual=r))
        ks.append(K)
        ts.append(us)
    A = np.vstack([np.ones(len(ks)), np.array(ks, dtype=np.float64)]).T
    (a0, b0), *_ = np.linalg.lstsq(A[2:], np.array(ts)[2:], rcond=None)
    tiles = ((M + 223) // 224) * ((N + 319) // 320)
    print(f"M={M} N={N} res={res} ({tiles} tiles of 224x320 = {tiles / 256:.2f} rounds): " + "  ".join(f"K={k}: {t:.1f}" for k, t in zip(ks, ts))
          + f"  |  fit (K >= 320): {a0:.1f} us + {b0 * 64:.3f} us per K-tile", flush=True)
