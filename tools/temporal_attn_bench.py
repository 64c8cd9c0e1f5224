"""k_x_temporal_attention at the SVD window's sizes (CFG batch = 2 videos, T = 14): time per launch and achieved HBM rate
(16 bytes per value: q | k | v read as fp32, the result written as the (hi, lo) image), next to the path it replaces (two permuted
copies + k_x_attention_f32 + split3).        python tools/temporal_attn_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vidseg_diffusion_amd import exact as X  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for (S, H) in ((9216, 5), (2304, 10), (576, 20), (144, 20)):
    nv, T, C = 2, 14, H * 64
    qkv = torch.randn((nv * T, S, 3 * C), device=dev)

    def old():
        t = qkv.view(nv, T, S, 3 * C).permute(0, 2, 1, 3).contiguous().view(nv * S, T, 3 * C)
        a = X.attention_f32(t[..., :C], t[..., C:2 * C], t[..., 2 * C:], H, nv * S, T, T)
        return X.split3(a.view(nv, S, T, C).permute(0, 2, 1, 3).contiguous().view(nv * T, S, C))

    us_new = timeit(lambda: X.temporal_attention_x(qkv, nv, T, S, H, split_out=True))
    us_old = timeit(old, 5)
    nbytes = nv * T * S * C * 16
    a, b = X.temporal_attention_x(qkv, nv, T, S, H, split_out=True), old()
    same = float((a[..., :2 * C].float() - b[..., :2 * C].float()).abs().max())
    print(f"S={S:5d} H={H:2d} C={C:4d}: in place {us_new:8.1f} us = {nbytes / us_new / 1e6:6.2f} TB/s of {nbytes / 1e6:7.1f} MB | permute + f32 kernel + split "
          f"{us_old:8.1f} us | max |image difference| {same:.2e}")
