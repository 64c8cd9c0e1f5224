"""Guard-band test of one full-size exact-mode UNet evaluation: every tensor the evaluation allocates sits between two NaN-filled guard
regions.  Out-of-bounds WRITES show up as damaged guards; out-of-bounds READS that matter show up as a changed (NaN-poisoned) output.
(lab probe for profiles/r06_e_sweep_lanes_race.txt)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

GUARD = int(os.environ.get("GUARD_BYTES", str(1 << 20)))


def main():
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    eng, cfg, _sd, _n = bench.build(False, False, dev)
    net = eng.model.diffusion_model
    net.set_precision("exact")
    net.tap_mode = os.environ.get("TAPS", "none")
    net._set_taps()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(28, 4, 64, 64, generator=g).to(dev)
    t = torch.full((28,), 958.0, device=dev)
    ctx = torch.randn(28, 77, 1024, generator=g).to(dev)
    ref = net(x, timesteps=t, context=ctx).clone()
    torch.cuda.synchronize()
    orig_empty = torch.empty
    guards = []

    def guarded_empty(*size, dtype=None, device=None, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        dt = dtype or torch.float32
        if device is None or torch.device(device).type != "cuda" or dt not in (torch.float16, torch.float32, torch.float64):
            return orig_empty(*size, dtype=dtype, device=device, **kw)
        es = orig_empty((), dtype=dt).element_size()
        n = 1
        for s_ in size:
            n *= int(s_)
        ge = GUARD // es
        pad = (-n) % (256 // es)                                          # keep the next guard 256-byte aligned
        big = torch.full((ge + n + pad + ge,), float("nan"), dtype=dt, device=device)
        guards.append((big, ge, n, pad, tuple(size), dt))
        return big[ge:ge + n].view(size)

    torch.empty = guarded_empty
    try:
        out = net(x, timesteps=t, context=ctx)
        torch.cuda.synchronize()
    finally:
        torch.empty = orig_empty
    print(f"{len(guards)} guarded allocations, {GUARD} bytes each side")
    print("output bit-identical to the unguarded evaluation:", torch.equal(out, ref), "| NaN in output:", bool(torch.isnan(out).any()))
    damaged = 0
    for i, (big, ge, n, pad, size, dt) in enumerate(guards):
        lo, hi = big[:ge], big[ge + n + pad:]
        wl, wh = ~torch.isnan(lo), ~torch.isnan(hi)
        nlo, nhi = int(wl.sum()), int(wh.sum())
        if nlo or nhi:
            damaged += 1
            if damaged <= 12:
                il = torch.nonzero(wl).reshape(-1)
                ih = torch.nonzero(wh).reshape(-1)
                print(f"  allocation #{i} {size} {dt}: {nlo} elements written BEFORE (nearest {ge - int(il.max()) if nlo else 0} elements before the start), "
                      f"{nhi} written AFTER (elements +{int(ih.min()) + pad if nhi else 0}..+{int(ih.max()) + pad if nhi else 0} past the end)", flush=True)
    print("allocations with damaged guards:", damaged)


if __name__ == "__main__":
    main()
