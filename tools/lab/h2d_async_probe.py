"""Does tensor.to(device, non_blocking=True) from PAGEABLE host memory read the source at call time (CUDA's documented behaviour) or
later, when the stream gets to it (then a freed / overwritten host temporary is a race)?"""
import time
import torch

dev = torch.device("cuda:0")
s = torch.cuda.Stream()
late = 0
for n in (28, 4096, 1 << 20):
    for trial in range(20):
        with torch.cuda.stream(s):
            torch.cuda._sleep(300_000_000)                    # ~0.12 s of queued work on this stream
            v = torch.full((n,), 1.0)
            t0 = time.perf_counter()
            d = v.to(dev, non_blocking=True)
            dt = time.perf_counter() - t0
            v.fill_(2.0)                                      # what a freed-and-reused temporary would look like
        torch.cuda.synchronize()
        if float(d.max()) != 1.0 or float(d.min()) != 1.0:
            late += 1
    print(f"n = {n}: the call returned after {dt * 1e3:.2f} ms (last trial); copies that saw the LATER host contents so far: {late}")
print("pageable non_blocking H2D reads the host buffer late:", late > 0)
