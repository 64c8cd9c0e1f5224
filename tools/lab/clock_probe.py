"""Run one conv shape in a loop for a few seconds and sample rocm-smi (sclk, power) meanwhile: shows how far the chip
clocks down under each kernel variant.  usage: python tools/lab/clock_probe.py [seconds]"""
import subprocess
import sys
import threading
import time

sys.path.insert(0, ".")
import torch
from vidseg_diffusion_amd import ops

dev = torch.device("cuda:0")
B, H, W, C0, C1, Co = 28, 32, 32, 1280, 640, 640
g = torch.Generator(device="cpu").manual_seed(1)
x0 = torch.randn((B, H, W, C0), generator=g).to(ops.act_dtype()).to(dev)
x1 = torch.randn((B, H, W, C1), generator=g).to(ops.act_dtype()).to(dev)
w = ops.pack_conv3x3(torch.randn((Co, C0 + C1, 3, 3), generator=g) * 0.03, dev)
b = torch.zeros(Co, device=dev)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
samples = []
stop = False


def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            s = [l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l]
            samples.append(" | ".join(x.split(":", 1)[-1].strip() if False else x for x in s))
        except Exception as e:  # noqa: BLE001
            samples.append(repr(e))
        time.sleep(0.4)


th = threading.Thread(target=poll)
th.start()
t0 = time.time()
n = 0
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
while time.time() - t0 < secs:
    for _ in range(50):
        ops.conv3x3(x0, w, b, x1=x1)
    n += 50
    torch.cuda.synchronize()
e.record()
torch.cuda.synchronize()
stop = True
th.join()
print(f"{s.elapsed_time(e) / n * 1e3:.1f} us per launch over {n} launches")
for x in samples[-5:]:
    print(x)
