"""Which block output first diverges in a racy lanes + shared-prefix pass?  Device-side checksums (no host sync) after every ExactRunner.block."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from vidseg_diffusion_amd import exact as X
    from vidseg_diffusion_amd.pipeline import modulation_sweep, segment_window
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    eng, cfg, _sd, _n = bench.build(False, False, dev)
    eng.model.diffusion_model.set_precision("exact")
    lat, c, uc, noise = bench.make_inputs(dev, 0, cfg)
    base, exp = "/nonexistent/lanes_check", "w0"
    lab, _ = segment_window(eng, lat, c, uc, num_masks=20, num_steps=25, t_start=22, seed=17, noise=noise, feature_folder=base, exp_name=exp, keep_all_steps=True)
    folder = os.path.join(base, exp, "match_gt_mask", "output_block_8_output_block_7_output_block_6_spatial_self_attn_q_masks_20")
    labels = [int(v) for v in np.unique(lab)]
    kw = dict(t_start=22, num_steps=25, feature_folder=base, exp_name=exp, noise=noise, seed=17)
    rec = []
    orig = X.ExactRunner.block
    names = {}

    def block(self, blk, x, *a, **k):
        out = orig(self, blk, x, *a, **k)
        rec.append((names.setdefault(id(blk), len(names)), out.double().abs().sum()))
        return out

    X.ExactRunner.block = block
    modulation_sweep(eng, lat, c, uc, labels, folder, share_prefix=True, lanes=1, **kw)
    torch.cuda.synchronize()
    ref = [(n, float(v)) for n, v in rec]
    print("block calls per sweep:", len(ref), flush=True)
    for trial in range(int(os.environ.get("TRIALS", "6"))):
        del rec[:]
        modulation_sweep(eng, lat, c, uc, labels, folder, share_prefix=True, lanes=2, **kw)
        torch.cuda.synchronize()
        got = [(n, float(v)) for n, v in rec]
        assert len(got) == len(ref) and all(a[0] == b[0] for a, b in zip(got, ref))
        bad = [i for i, (a, b) in enumerate(zip(got, ref)) if a[1] != b[1]]
        # a pass = 72 block calls (job 0) or 53 (resumed: 5 + 24 + 24)
        def where(i):
            if i < 72:
                return 0, i // 24, ref[i][0]
            j, r = divmod(i - 72, 53)
            ev = 0 if r < 5 else (1 if r < 29 else 2)
            return j + 1, ev, ref[i][0]
        firsts = {}
        for i in bad:
            p, ev, blk = where(i)
            firsts.setdefault(p, (ev, blk, i))
        print(f"trial {trial}: {len(firsts)} racy passes; first divergent block call of each (pass, evaluation 0/1/2, block id in call order): "
              f"{[(p,) + v[:2] for p, v in sorted(firsts.items())]}", flush=True)


if __name__ == "__main__":
    main()
