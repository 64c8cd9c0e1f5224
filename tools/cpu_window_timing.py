"""One FULL configs[1] window through the CPU oracle on this host's cores, timed: the measurement bench.py's `cpu_baseline` extrapolates to
(`3 * (14 / 4) * t_unet(4 frames) + t_analysis`).

    python tools/cpu_window_timing.py [--window 0] [--threads N]        # prints one JSON line

SD 2.1 full-width UNet (synthetic near-init weights), 14 x 512 x 512 (latent 14 x 4 x 64 x 64), CFG batch 28, three Euler steps
(t_start 22), Q taps of decoder blocks 6/7/8, 3-block mean, K-means K = 20 n_init = 10, 4-NN -- oracle.pipeline.segment_window, fp32,
the whole CFG batch per evaluation; the masks are scored against the reference's fixture for that window.  Uses `oracle/` (test
infrastructure) as the thing measured on purpose: this is the CPU baseline leg, never the product path.  ~35 GB of host memory.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    torch.set_grad_enabled(False)
    from oracle.pipeline import segment_window
    from oracle.unet import UNetOracle
    from tools_metrics import matched_iou
    from vidseg_diffusion_amd import synthetic
    from vidseg_diffusion_amd.unet import UNetModel
    F, LAT, K = 14, 64, 20
    cfg = dict(synthetic.SD21_FULL)
    net = UNetModel(**cfg)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1234, zero_gain=synthetic.HEADLINE["zero_gain"]).items()}
    del net
    w = args.window
    lat = torch.from_numpy(synthetic.headline_latent(F, LAT, LAT, window_id=w))
    c, uc = synthetic.sd_conditioning(F, context_dim=cfg["context_dim"], seq=77, seed=1)
    noise = torch.randn((F, 4, LAT, LAT), generator=torch.Generator().manual_seed(100 + w))
    o = UNetOracle(sd)
    t0 = time.time()
    out = segment_window(o, lat, torch.from_numpy(c), torch.from_numpy(uc), noise, num_masks=K, seed=17)
    dt = time.time() - t0
    rec = {"window": w, "seconds_per_window": round(dt, 1), "frames_per_s": round(F / dt, 5), "threads": torch.get_num_threads(),
           "what": "oracle.pipeline.segment_window: 3 full CFG evaluations (batch 28, fp32) + 3-block mean + K-means (n_init 10) + 4-NN on the host"}
    gpath = os.path.join(ROOT, "tests", "golden", "c2_window.npz" if w == 0 else f"c2_window_w{w}.npz")
    if os.path.exists(gpath):
        g = np.load(gpath)
        iou, ident = matched_iou(out["labels"].reshape(-1), g["match_labels"].astype(np.int64).reshape(-1), K)
        rec.update(mask_iou_vs_reference=round(float(iou), 4), identical_fraction=round(float(ident), 4))
    try:
        with open("/proc/cpuinfo") as fh:
            rec["cpu"] = next(line.split(":", 1)[1].strip() for line in fh if line.startswith("model name"))
    except Exception:                                                    # noqa: BLE001
        pass
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
