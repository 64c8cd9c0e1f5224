"""Headline window (BASELINE configs[1]) segmented under several fp32-summation-order variants of the same arithmetic, each compared
with the REFERENCE's fp32 masks (tests/golden/c2_window.npz) and with the default variant: how much of the mask agreement is the
kernels' accuracy and how much is the K-means / best-of-10 selection reacting to rounding-level changes of its input.

    python tools/lab/mask_sensitivity.py            (GPU box; VIDSEG_ACT=bf16 for the bf16 build)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tools_metrics import matched_iou  # noqa: E402


def main():
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd import ops, synthetic
    from vidseg_diffusion_amd.pipeline import build_sd_engine, segment_window
    from vidseg_diffusion_amd.unet import UNetModel
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(ROOT, "tests", "golden", "c2_window.npz"))
    F, LAT, K = int(g["F"]), int(g["lat"]), int(g["K"])
    cfg = dict(synthetic.SD21_FULL)
    net = UNetModel(**cfg)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1234, zero_gain=float(g["zero_gain"])).items()})
    net.pack(dev)
    eng = build_sd_engine(net, num_steps=25, scale=5.0)
    lat = torch.from_numpy(synthetic.headline_latent(F, LAT, LAT, 0)).to(dev)
    c, uc = synthetic.sd_conditioning(F, context_dim=cfg["context_dim"], seq=77, seed=1)
    c, uc = {"crossattn": torch.from_numpy(c).to(dev)}, {"crossattn": torch.from_numpy(uc).to(dev)}
    noise = torch.randn((F, 4, LAT, LAT), generator=torch.Generator().manual_seed(100)).to(dev)
    ref = g["match_labels"].astype(np.int64).reshape(F, -1)
    res = {}
    for name, rpc, mo in (("default", 0, False), ("gn chunks of 64 rows", 64, False), ("gn chunks of 16 rows", 16, False),
                          ("last step on the cond half (masks_only)", 0, True)):
        ops._GN_RPC_FORCE = rpc
        FE.FeatureStore.clear()
        FE.MaskStore.clear()
        lab, _ = segment_window(eng, lat, c, uc, num_masks=K, num_steps=25, t_start=22, seed=17, noise=noise, keep_all_steps=False,
                                masks_only=mo, feature_folder="/nonexistent/sens", exp_name="s")
        res[name] = np.asarray(lab).reshape(F, -1)
        i1, e1 = matched_iou(res[name], ref, K)
        i2, e2 = matched_iou(res[name], res["default"], K)
        print(f"{name:42s} vs reference: IoU {i1:.4f} identical {e1:.4f} | vs default: IoU {i2:.4f} identical {e2:.4f}", flush=True)
    ops._GN_RPC_FORCE = 0


if __name__ == "__main__":
    main()
