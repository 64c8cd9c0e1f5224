"""Where does the 1.2e-3 error of the step-24 Q taps come from?  (CPU, build container.)

Runs the oracle on window 0 of the headline clip with selected classes of values kept in fp32 while the rest is rounded to fp16
(`UNetOracle(round_bf16="f16", round_spec={...})`) and reports the normalised rms of the block-6/7/8 taps against the REFERENCE's
(tests/golden/c2_window.npz) plus the Step-3 mask IoU.  What it is for: deciding which stores of the HIP path are worth keeping
wider (LayerNorm folded into the consuming GEMM = "ln": False; fp32 residual stream = "res": False; ...).

    python tools/tap_error_study.py [--window 0] [--variants f16 ln res ln+res w f32]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from mask_rounding_study import Chunked  # noqa: E402
from oracle import pipeline as OP  # noqa: E402
from tools_metrics import matched_iou  # noqa: E402
from vidseg_diffusion_amd import synthetic  # noqa: E402

F, LAT, K = 14, 64, 20
VARIANTS = {"f16": {}, "ln": {"ln": False}, "res": {"res": False}, "ln+res": {"ln": False, "res": False}, "w": {"w": False},
            "ln+res+w": {"ln": False, "res": False, "w": False}, "f32": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--variants", nargs="*", default=["f16", "ln", "res", "ln+res", "w"])
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    from vidseg_diffusion_amd.unet import UNetModel
    cfg = dict(synthetic.SD21_FULL)
    shapes = {k: tuple(v.shape) for k, v in UNetModel(**cfg).state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1234, zero_gain=synthetic.HEADLINE["zero_gain"]).items()}
    c, uc = synthetic.sd_conditioning(F, context_dim=cfg["context_dim"], seq=77, seed=1)
    w = args.window
    g = np.load(os.path.join(ROOT, "tests", "golden", "c2_window.npz" if w == 0 else f"c2_window_w{w}.npz"))
    cache = os.path.join(os.environ.get("C2_TAP_CACHE", "/tmp/vidseg_taps"), f"taps_w{w}.npz")
    full = np.load(cache) if os.path.exists(cache) else None
    lat = synthetic.headline_latent(F, LAT, LAT, window_id=w)
    noise = torch.randn((F, 4, LAT, LAT), generator=torch.Generator().manual_seed(100 + w))
    out = []
    for name in args.variants:
        t0 = time.time()
        spec = VARIANTS[name]
        net = Chunked(sd, round_bf16=False if spec is None else "f16")
        if spec:
            net.rb_w, net.rb_ln, net.rb_res = (spec.get(k, "f16") for k in ("w", "ln", "res"))
        res = OP.segment_window(net, torch.from_numpy(lat), torch.from_numpy(c), torch.from_numpy(uc), noise, num_masks=K, t_start=22, seed=17)
        errs = []
        for b in (6, 7, 8):
            q = res["q_taps"][b][F:].astype(np.float64)
            ref = full[f"q{b}"].astype(np.float64) if full is not None else g[f"q{b}_sub"].astype(np.float64)
            if full is None:
                q = q[:, ::16, ::2]
            errs.append(float(np.linalg.norm(q - ref) / np.linalg.norm(ref)))
        iou, ex = matched_iou(res["match_labels"], g["match_labels"].astype(np.int64), K)
        line = f"window {w} variant {name:9s}: tap nrms blocks 6/7/8 {errs[0]:.2e} {errs[1]:.2e} {errs[2]:.2e}; Step-3 IoU {iou:.4f} identical {ex:.4f} ({time.time() - t0:.0f} s)"
        print(line, flush=True)
        out.append(line)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"tap_error_study_w{w}.txt"), "w") as fh:
        fh.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
