import sys; sys.path.insert(0, '.')
import torch
from vidseg_diffusion_amd import ops
dev = torch.device('cuda:0')
M, K, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (114688, 320, 320)))
a = torch.randn(M, K, device=dev).to(ops.act_dtype()); w = (torch.randn(N, K, device=dev) * 0.02).to(ops.act_dtype()); b = torch.zeros(N, device=dev)
for _ in range(5):
    ops.linear(a, w, b)
torch.cuda.synchronize()
