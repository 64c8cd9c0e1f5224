"""One 3x3 conv shape in a loop for a few seconds with rocm-smi sampled meanwhile: time per launch, TFLOP/s, shader clock, socket
power.  The tile-selection knobs of the library are read once per process, so A/B two kernels with two invocations:

    VIDSEG_GEMM=p7=0 python tools/lab/power_probe.py 28 64 64 320 0 320      # B H W C0 C1 Cout [up] [seconds]
    VIDSEG_GEMM=p7=1 python tools/lab/power_probe.py 28 64 64 320 0 320
"""
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, ".")
import torch
from vidseg_diffusion_amd import ops

dev = torch.device("cuda:0")
a = [int(v) for v in sys.argv[1:7]]
B, H, W, C0, C1, Co = a
up = int(sys.argv[7]) if len(sys.argv) > 7 else 1
secs = float(sys.argv[8]) if len(sys.argv) > 8 else 3.0
g = torch.Generator(device="cpu").manual_seed(1)
x0 = torch.randn((B, H, W, C0), generator=g).to(ops.act_dtype()).to(dev)
x1 = torch.randn((B, H, W, C1), generator=g).to(ops.act_dtype()).to(dev) if C1 else None
w = ops.pack_conv3x3(torch.randn((Co, C0 + C1, 3, 3), generator=g) * 0.03, dev)
b = torch.zeros(Co, device=dev)
clk, pwr = [], []
stop = False


def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            for l in out.splitlines():
                m = re.search(r"sclk.*\((\d+)Mhz\)", l)
                if m:
                    clk.append(int(m.group(1)))
                m = re.search(r"Power.*?:\s*([\d.]+)", l)
                if m and "Power" in l:
                    pwr.append(float(m.group(1)))
        except Exception:  # noqa: BLE001
            pass
        time.sleep(0.3)


for _ in range(20):
    ops.conv3x3(x0, w, b, x1=x1, up=up)
torch.cuda.synchronize()
th = threading.Thread(target=poll)
th.start()
t0 = time.time()
n = 0
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
while time.time() - t0 < secs:
    for _ in range(50):
        ops.conv3x3(x0, w, b, x1=x1, up=up)
    n += 50
    torch.cuda.synchronize()
e.record()
torch.cuda.synchronize()
stop = True
th.join()
us = s.elapsed_time(e) / n * 1e3
fl = 2.0 * B * H * up * W * up * Co * 9 * (C0 + C1)
med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
print(f"conv B{B} {H}x{W} up{up} {C0}+{C1}->{Co}: {us:.1f} us/launch, {fl / us / 1e6:.0f} TFLOP/s, sclk median {med(clk[2:])} MHz, "
      f"power median {med(pwr[2:])} W ({len(clk)} samples)")
