"""Full-size SD Step-4 sweep: which of (shared prefix, lanes) changes bits, and is it run-to-run stable?  (lab probe, round 6)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import modulation_sweep, segment_window
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    eng, cfg, _sd, _n = bench.build(False, False, dev)
    eng.model.diffusion_model.set_precision("exact")
    lat, c, uc, noise = bench.make_inputs(dev, 0, cfg)
    base, exp = "/nonexistent/lanes_check", "w0"
    lab, _ = segment_window(eng, lat, c, uc, num_masks=20, num_steps=25, t_start=22, seed=17, noise=noise, feature_folder=base, exp_name=exp, keep_all_steps=True)
    folder = os.path.join(base, exp, "match_gt_mask", "output_block_8_output_block_7_output_block_6_spatial_self_attn_q_masks_20")
    labels = [int(v) for v in np.unique(lab)][:int(os.environ.get("NLAB", "4"))]
    kw = dict(t_start=22, num_steps=25, feature_folder=base, exp_name=exp, noise=noise, seed=17)
    ref = modulation_sweep(eng, lat, c, uc, labels, folder, share_prefix=False, lanes=1, **kw)
    torch.cuda.synchronize()
    plans = ((False, 1), (True, 1), (False, 2), (False, 2), (True, 2), (True, 2), (False, 3)) if len(labels) < 8 else ((True, 1), (False, 2), (True, 2), (True, 2))
    if os.environ.get("PLANS"):
        plans = tuple((True, 2) for _ in range(int(os.environ["PLANS"])))
    for share, lanes in plans:
        got = modulation_sweep(eng, lat, c, uc, labels, folder, share_prefix=share, lanes=lanes, **kw)
        torch.cuda.synchronize()
        bad = {k: float((got[k] - ref[k]).abs().max()) for k in ref if not torch.equal(got[k], ref[k])}
        print(f"share_prefix={share} lanes={lanes}: {len(ref) - len(bad)} of {len(ref)} latents bit-identical to the plain sweep; differing: "
              f"{ {k: f'{v:.2e}' for k, v in bad.items()} }", flush=True)


if __name__ == "__main__":
    main()
