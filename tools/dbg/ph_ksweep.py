"""Fixed cost of the phased big tile: time of M x 320 x K linears as K grows (VIDSEG_GEMM_BIG=2 forces it)."""
import sys
sys.path.insert(0, ".")
import torch
from vidseg_diffusion_amd import ops
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 114688
for N in (320, 640):
    for K in (64, 128, 320, 640, 1280, 2560):
        a = torch.randn(M, K, device=dev).to(ops.act_dtype())
        w = (torch.randn(N, K, device=dev) * 0.02).to(ops.act_dtype())
        f = lambda: ops.linear(a, w, None)
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30):
            f()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / 30 * 1e3
        print(f"M{M} N{N} K{K}: {us:7.1f} us  {2.0*M*N*K/us/1e6:6.0f} TF/s", flush=True)
