"""What the vendor library (torch -> hipBLASLt/rocBLAS) reaches on the short-K linears: a calibration point only."""
import torch
dev = torch.device('cuda:0')
def bench(fn, flops, name, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"{name:44s} {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TF/s")
for (M, K, N) in [(114688, 320, 320), (114688, 320, 2560), (28672, 640, 640), (114688, 320, 960), (28672, 640, 5120), (7168, 1280, 1280), (114688, 1280, 320), (28672, 5760, 640)]:
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.02).bfloat16(); b = torch.zeros(N, device=dev).bfloat16()
    bench(lambda: torch.nn.functional.linear(a, w, b), 2 * M * N * K, f"torch linear M{M} K{K} N{N}")
