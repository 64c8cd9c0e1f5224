"""Where does a racy lanes+shared-prefix pass first diverge?  x after every sampler step of every pass, racy sweep vs plain sweep."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from vidseg_diffusion_amd import sampling
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
    if os.environ.get("NOINJECT") == "1":
        kw["is_injected_features"] = False
    if os.environ.get("NOTAPS") == "1":
        net = eng.model.diffusion_model
        net.tap_mode = "none"
        net._set_taps()
    trace = []
    orig = sampling.EDMSampler.__call__

    def patched(self, denoiser, x, cond, **k):
        if k.get("is_modulate"):
            rec = []
            trace.append(rec)
            k["img_callback"] = lambda xt, i: rec.append(xt.clone())
        return orig(self, denoiser, x, cond, **k)

    sampling.EDMSampler.__call__ = patched
    modulation_sweep(eng, lat, c, uc, labels, folder, share_prefix=False, lanes=1, **kw)
    torch.cuda.synchronize()
    ref = [list(r) for r in trace]
    for trial in range(int(os.environ.get("TRIALS", "6"))):
        del trace[:]
        modulation_sweep(eng, lat, c, uc, labels, folder, share_prefix=os.environ.get('SHARE', '1') == '1', lanes=2, **kw)
        torch.cuda.synchronize()
        msgs = []
        for j, (a, b) in enumerate(zip(trace, ref)):
            d = [float((x - y).abs().max()) for x, y in zip(a, b)]
            if any(v > 0 for v in d):
                msgs.append(f"pass {j} (lane {'main' if j == 0 else (j - 1) % 2}): max |dx| after steps 22/23/24 = {['%.2e' % v for v in d]}; "
                            f"elements differing after the first bad step: {int((a[[v > 0 for v in d].index(True)] != b[[v > 0 for v in d].index(True)]).sum())}")
        print(f"trial {trial}: {len(msgs)} racy passes" + ("".join("\n    " + m for m in msgs)), flush=True)


if __name__ == "__main__":
    main()
