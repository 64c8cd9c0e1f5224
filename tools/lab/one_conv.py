"""A few launches of one 3x3 conv shape (for rocprofv3 --pmc passes): python tools/lab/one_conv.py B H W C0 C1 Cout [n]"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vidseg_diffusion_amd import ops
dev = torch.device("cuda:0")
B, H, W, C0, C1, Co = [int(v) for v in sys.argv[1:7]]
n = int(sys.argv[7]) if len(sys.argv) > 7 else 6
g = torch.Generator(device="cpu").manual_seed(1)
x0 = torch.randn((B, H, W, C0), generator=g).to(ops.act_dtype()).to(dev)
x1 = torch.randn((B, H, W, C1), generator=g).to(ops.act_dtype()).to(dev) if C1 else None
w = ops.pack_conv3x3(torch.randn((Co, C0 + C1, 3, 3), generator=g) * 0.03, dev)
b = torch.zeros(Co, device=dev)
for _ in range(n):
    ops.conv3x3(x0, w, b, x1=x1)
torch.cuda.synchronize()
