"""clock / power while one short-K linear runs in a loop (see clock_probe.py).  usage: clock_probe_lin.py M N K [act]"""
import subprocess
import sys
import threading
import time

sys.path.insert(0, ".")
import torch
from vidseg_diffusion_amd import ops

dev = torch.device("cuda:0")
M, N, K = (int(v) for v in sys.argv[1:4])
act = int(sys.argv[4]) if len(sys.argv) > 4 else 0
a = torch.randn(M, K, device=dev).to(ops.act_dtype())
w = (torch.randn(N, K, device=dev) * 0.02).to(ops.act_dtype())
b = torch.zeros(N, device=dev)
samples, stop = [], False


def poll():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        s = [l.split(":")[-1].strip() for l in out.splitlines() if "sclk" in l or "Package Power" in l]
        samples.append(" | ".join(s))
        time.sleep(0.4)


th = threading.Thread(target=poll)
th.start()
t0, n = time.time(), 0
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
while time.time() - t0 < 3.5:
    for _ in range(100):
        ops.linear(a, w, b, act=act)
    n += 100
    torch.cuda.synchronize()
e.record()
torch.cuda.synchronize()
stop = True
th.join()
us = s.elapsed_time(e) / n * 1e3
print(f"M{M} N{N} K{K} act{act}: {us:.1f} us {2.0 * M * N * K / us / 1e6:.0f} TF/s; {samples[-3:]}")
