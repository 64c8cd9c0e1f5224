"""Time the UNet's output conv (Cout = 4) at the window's size (GPU box): python tools/lab/conv_out_bench.py  (VIDSEG_GEMM=convout=0/1)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vidseg_diffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for B, H, C in ((28, 64, 320), (14, 512, 128)):
    x = torch.randn(B, H, H, C, device=dev).to(ops.act_dtype())
    w = ops.pack_conv_out(torch.randn(4, C, 3, 3) * 0.05, dev)
    b = torch.zeros(4, device=dev)
    for _ in range(3):
        ops.conv_out4(x, w, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.conv_out4(x, w, b)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 20
    print(f"conv_out4 B={B} {H}x{H} Cin={C}: {us:8.1f} us  ({x.numel() * 2 / us / 1e6:.2f} TB/s of input)  WS={os.environ.get('VIDSEG_GEMM', 'default')}", flush=True)
