"""HBM rate of the exact mode's fp32 glue kernels (LayerNorm / GroupNorm -> split image, split3_cat) at the parity window's sizes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vidseg_diffusion_amd import exact as X, ops
dev = torch.device("cuda:0")
ops.workspace(dev)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for B in (28, 14):
    for HW, C in ((4096, 320), (1024, 640), (256, 1280)):
        M = B * HW
        x = torch.randn(M, C, device=dev); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
        us = timeit(lambda: X.layernorm_split3(x, g, b))
        by = M * C * (4 + 4)
        print(f"layernorm_split3 B={B} M={M} C={C}: {us:7.1f} us  {by/us/1e6:6.2f} TB/s")
        x4 = x.reshape(B, HW, C)
        us = timeit(lambda: X.groupnorm_split3(x4, g, b))
        print(f"groupnorm_split3 (partial+finish+apply) B={B} HW={HW} C={C}: {us:7.1f} us  {(M*C*(4+4+4))/us/1e6:6.2f} TB/s (x read twice + image)")
        us = timeit(lambda: X.split3(x))
        print(f"split3 M={M} C={C}: {us:7.1f} us  {by/us/1e6:6.2f} TB/s")
    for HW, C0, C1 in ((4096, 320, 320), (4096, 640, 320), (1024, 640, 640), (1024, 1280, 640), (256, 1280, 1280)):
        a0 = torch.randn(B, HW, C0, device=dev); a1 = torch.randn(B, HW, C1, device=dev)
        g = torch.ones(C0 + C1, device=dev); b = torch.zeros(C0 + C1, device=dev)
        us = timeit(lambda: X.groupnorm_split3(a0, g, b, x1=a1))
        print(f"groupnorm_split3 concat B={B} HW={HW} C={C0}+{C1}: {us:7.1f} us  {(B*HW*(C0+C1)*12)/us/1e6:6.2f} TB/s")
        us = timeit(lambda: X.split3_cat(a0, a1))
        print(f"split3_cat B={B} HW={HW} C={C0}+{C1}: {us:7.1f} us  {(B*HW*(C0+C1)*8)/us/1e6:6.2f} TB/s")
