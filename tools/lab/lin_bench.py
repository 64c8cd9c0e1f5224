"""Time the short-K linears of the window (GPU box): python tools/lab/lin_bench.py   (VIDSEG_GEMM=ws=0/1 to compare)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vidseg_diffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def bench(fn, flops, nbytes, name, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    print(f"{name:40s} {us:8.1f} us  {flops / us / 1e6:7.1f} TF/s  {nbytes / us / 1e6:6.2f} TB/s algorithmic", flush=True)


for M, K, N, res in ((114688, 320, 320, False), (114688, 320, 320, True), (114688, 320, 960, False), (28672, 640, 640, True), (28672, 640, 1920, False)):
    a = torch.randn(M, K, device=dev).to(ops.act_dtype())
    w = (torch.randn(N, K, device=dev) * 0.02).to(ops.act_dtype())
    b = torch.zeros(N, device=dev)
    r = torch.randn(M, N, device=dev).to(ops.act_dtype()) if res else None
    bench(lambda: ops.linear(a, w, b, residual=r), 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N * (2 if res else 1)),
          f"linear M{M} K{K} N{N}{' +res' if res else ''} (WS={os.environ.get('VIDSEG_GEMM', 'default')})")

# experiment builds with -DVS_WS_STAMPS: phase timeline of tiles 1..4 of block 40, per wave (the scalar counter path adds its own latency)
import ctypes  # noqa: E402
from vidseg_diffusion_amd import _lib  # noqa: E402
L = _lib.lib()
if hasattr(L, "vidseg_debug_ws_stamps"):
    M, K, N = 114688, 320, 960
    a = torch.randn(M, K, device=dev).to(ops.act_dtype())
    w = (torch.randn(N, K, device=dev) * 0.02).to(ops.act_dtype())
    ops.linear(a, w, torch.zeros(N, device=dev))
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 128)()
    L.vidseg_debug_ws_stamps(buf)
    for wv in range(8):
        parts = []
        for t in range(4):
            s0, s1, s2 = (buf[(wv * 4 + t) * 4 + i] for i in range(3))
            nxt = buf[(wv * 4 + t + 1) * 4] if t < 3 else None
            parts.append(f"tile {t + 1}: K loop {s1 - s0} epilogue {s2 - s1}" + (f" edge {nxt - s2}" if nxt else ""))
        print(f"wave {wv}: " + " | ".join(parts))
