"""Why is a conv slower inside the UNet than in a hot loop?  Times ONE 3x3 conv shape (HIP events around the conv only) when
  hot     the same input / output buffers every launch (what tools/lab/power_probe.py measures)
  rotate  NSETS different input / output buffer sets in turn (working set >> the 256 MB Infinity Cache)
  gn+conv GroupNorm+SiLU writes the conv's input right before it (the UNet's pattern), rotating sets
usage: python tools/lab/temp_probe.py B H W C0 Cout [nsets]"""
import sys

sys.path.insert(0, ".")
import torch
from vidseg_diffusion_amd import ops

dev = torch.device("cuda:0")
B, H, W, C0, Co = [int(v) for v in sys.argv[1:6]]
NS = int(sys.argv[6]) if len(sys.argv) > 6 else 8
g = torch.Generator(device="cpu").manual_seed(1)
xs = [torch.randn((B, H, W, C0), generator=g).to(ops.act_dtype()).to(dev) for _ in range(NS)]
w = ops.pack_conv3x3(torch.randn((Co, C0, 3, 3), generator=g) * 0.03, dev)
b = torch.zeros(Co, device=dev)
gam, bet = torch.ones(C0, device=dev), torch.zeros(C0, device=dev)
fl = 2.0 * B * H * W * Co * 9 * C0


def run(mode, iters=160):
    evs = []
    for it in range(iters + 20):
        x = xs[0] if mode == "hot" else xs[it % NS]
        if mode == "gn+conv":
            x = ops.groupnorm(x, gam, bet, silu=True)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.conv3x3(x, w, b)
        e.record()
        if it >= 20:
            evs.append((s, e))
    torch.cuda.synchronize()
    us = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
    med = us[len(us) // 2]
    print(f"{mode:8s} conv B{B} {H}x{W} {C0}->{Co}: median {med:.1f} us ({fl / med / 1e6:.0f} TFLOP/s), p10 {us[len(us) // 10]:.1f}, p90 {us[9 * len(us) // 10]:.1f}")


for mode in ("hot", "rotate", "gn+conv", "hot"):
    run(mode)
