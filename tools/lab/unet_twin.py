"""Two full-size exact-mode UNet evaluations on two HIP streams at once: is the victim's output bit-stable, and if not, which block's output
diverges first?  (lab probe for profiles/r06_e_sweep_lanes_race.txt)"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from vidseg_diffusion_amd import exact as X
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    eng, cfg, _sd, _n = bench.build(False, False, dev)
    net = eng.model.diffusion_model
    net.set_precision("exact")
    net.tap_mode = "none"
    net._set_taps()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(28, 4, 64, 64, generator=g).to(dev)
    t = torch.full((28,), 958.0, device=dev)
    ctx = torch.randn(28, 77, 1024, generator=g).to(dev)
    x2 = torch.randn(28, 4, 64, 64, generator=g).to(dev)
    rec = None
    names = []

    def wrap(name):
        orig = getattr(X, name)

        def f(*a, **k):
            out = orig(*a, **k)
            if rec is not None:
                for o in (out if isinstance(out, tuple) else (out,)):
                    for oo in (o if isinstance(o, tuple) else (o,)):
                        if torch.is_tensor(oo):
                            rec.append(oo)
                            if len(names) < len(rec):
                                names.append(f"{name} {tuple(oo.shape)}")
            return out
        setattr(X, name, f)

    for nm in ("split3", "split3_cat", "geglu_split3", "groupnorm_split3", "layernorm_split3", "attention_x", "linear_x", "linear_qkv_x", "conv3x3_x",
               "geglu_linear_x", "split_planes"):
        wrap(nm)
    net(x, timesteps=t, context=ctx)
    torch.cuda.synchronize()
    rec = []
    ref_out = net(x, timesteps=t, context=ctx)
    torch.cuda.synchronize()
    ref = list(rec)
    rec = None
    side = torch.cuda.Stream()
    iters = int(os.environ.get("ITERS", "150"))
    bad_runs = 0
    random.seed(1)
    for it in range(iters):
        mode = os.environ.get("MODE", "late")
        with torch.cuda.stream(side):
            if mode == "late":                      # the aggressor begins while the victim is somewhere in its evaluation
                torch.cuda._sleep(random.randrange(1, 150_000_000))
            agg = net(x2, timesteps=t, context=ctx)
            if mode == "late":
                torch.cuda._sleep(random.randrange(1, 60_000_000))
                agg = net(x2, timesteps=t, context=ctx)
        rec = []
        out = net(x, timesteps=t, context=ctx)
        got = list(rec)
        rec = None
        torch.cuda.synchronize()
        def live(tn):
            c3 = tn.shape[-1]
            return tn[..., :2 * (c3 // 3)] if tn.dtype == torch.float16 and c3 % 192 == 0 else tn
        bads = [i for i, (a, b) in enumerate(zip(got, ref)) if not torch.equal(live(a), live(b))]
        if bads or not torch.equal(out, ref_out):
            bad_runs += 1
            if not bads:
                print(f"iter {it}: every op output identical, final output differs", flush=True)
            else:
                first = bads[0]
                d = (live(got[first]).float() - live(ref[first]).float()).abs()
                rows = d.reshape(-1, d.shape[-1])
                badrows = torch.nonzero(rows.amax(1) > 0).reshape(-1)
                badcols = torch.nonzero(rows.amax(0) > 0).reshape(-1)
                print(f"iter {it}: first divergent op output #{first} of {len(ref)}: {names[first]} ({len(bads)} outputs differ in all): {int((d > 0).sum())} elements, "
                      f"max {float(d.max()):.3e}; rows {int(badrows.min())}..{int(badrows.max())} ({badrows.numel()} rows), cols {int(badcols.min())}..{int(badcols.max())} "
                      f"({badcols.numel()} cols); previous op: {names[first - 1]}; next: {names[min(first + 1, len(names) - 1)]}", flush=True)
        del agg
    print(f"{bad_runs} of {iters} victim evaluations differ", flush=True)


if __name__ == "__main__":
    main()
