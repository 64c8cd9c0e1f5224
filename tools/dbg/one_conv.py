import sys; sys.path.insert(0, '.')
import torch
from vidseg_diffusion_amd import ops
dev = torch.device('cuda:0')
B, H, Cin, Cout = 28, 32, 640, 640
x0 = torch.randn(B, H, H, Cin, device=dev).to(ops.act_dtype())
w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).to(ops.act_dtype()); b = torch.zeros(Cout, device=dev)
for _ in range(5): ops.conv3x3(x0, w, b)
torch.cuda.synchronize()
