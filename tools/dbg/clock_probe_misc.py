"""Clock / power while attention or GroupNorm runs in a loop (same method as clock_probe.py)."""
import subprocess, sys, threading, time
sys.path.insert(0, ".")
import torch
from vidseg_diffusion_amd import ops
dev = torch.device("cuda:0")
which = sys.argv[1]
if which == "attn":
    B, H, N = 28, 5, 4096
    qkv = torch.randn((B, N, 3 * H * 64), device=dev).to(ops.act_dtype())
    C = H * 64
    f = lambda: ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], H)
    work = 4.0 * B * H * N * N * 64
else:
    x = torch.randn((28, 64, 64, 320), device=dev).to(ops.act_dtype())
    g, b = torch.ones(320, device=dev), torch.zeros(320, device=dev)
    f = lambda: ops.groupnorm(x, g, b, eps=1e-5, silu=True)
    work = 3.0 * x.numel() * 2
samples, stop = [], False
def poll():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        samples.append(" | ".join(l.split(":")[-1].strip() for l in out.splitlines() if "sclk" in l or "Package Power" in l))
        time.sleep(0.4)
th = threading.Thread(target=poll); th.start()
t0, n = time.time(), 0
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
while time.time() - t0 < 3.5:
    for _ in range(50): f()
    n += 50; torch.cuda.synchronize()
e.record(); torch.cuda.synchronize(); stop = True; th.join()
us = s.elapsed_time(e) / n * 1e3
print(which, f"{us:.1f} us", f"{work / us / 1e6:.0f} T(FLOP or B)/s", samples[-3:])
