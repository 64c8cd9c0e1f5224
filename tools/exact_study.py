"""GPU box: the exact precision mode on the fixture windows of the headline clip -- tap error (window 0), masks vs the reference,
time per window.  Writes gpurun_out/exact_study.txt.      python tools/exact_study.py [--windows 0-15]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tools_metrics import matched_iou  # noqa: E402
from vidseg_diffusion_amd import synthetic  # noqa: E402

F, LAT, K = 14, 64, 20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", default="0-15")
    ap.add_argument("--refine", action="store_true")
    args = ap.parse_args()
    a, _, b = args.windows.partition("-")
    wids = list(range(int(a), int(b or a) + 1))
    from vidseg_diffusion_amd import analysis as A
    A.KEEP_LAST = True
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import build_sd_engine, segment_window
    from vidseg_diffusion_amd.unet import UNetModel
    dev = torch.device("cuda:0")
    cfg = dict(synthetic.SD21_FULL)
    net = UNetModel(**cfg)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1234, zero_gain=synthetic.HEADLINE["zero_gain"]).items()})
    net.set_precision("exact")
    eng = build_sd_engine(net, num_steps=25, scale=5.0)
    c, uc = synthetic.sd_conditioning(F, context_dim=cfg["context_dim"], seq=77, seed=1)
    cc, ucc = {"crossattn": torch.from_numpy(c).to(dev)}, {"crossattn": torch.from_numpy(uc).to(dev)}
    lines = []
    for w in wids:
        path = os.path.join(ROOT, "tests", "golden", "c2_window.npz" if w == 0 else f"c2_window_w{w}.npz")
        if not os.path.exists(path):
            continue
        g = np.load(path)
        lat = synthetic.headline_latent(F, LAT, LAT, window_id=w)
        noise = torch.randn((F, 4, LAT, LAT), generator=torch.Generator().manual_seed(100 + w))
        FE.FeatureStore.clear()
        FE.MaskStore.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        labels, _ = segment_window(eng, torch.from_numpy(lat).to(dev), cc, ucc, num_masks=K, num_steps=25, t_start=22, seed=17,
                                   noise=noise.to(dev), feature_folder="/nonexistent/ex", exp_name=f"w{w}", keep_all_steps=False,
                                   is_refine_mask=args.refine)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        iou, ex = matched_iou(labels, g["corrected_labels" if args.refine else "match_labels"].astype(np.int64), K)
        extra = ""
        if "q7_sub" in g.files:
            st = FE.FeatureStore.folder("/nonexistent/ex", f"w{w}")
            errs = []
            for bk in (6, 7, 8):
                q = st[f"output_block_{bk}_spatial_self_attn_q_time_24"].cpu().numpy()[F:, ::16, ::2].astype(np.float64)
                ref = g[f"q{bk}_sub"].astype(np.float64)
                errs.append(np.linalg.norm(q - ref) / np.linalg.norm(ref))
                extra += f" q{bk}: nrms {errs[-1]:.2e}, fp16 values differing {np.mean(q != ref):.4f};"
        km = A.LAST_KMEANS
        same = 0
        if "restart_labels" in g.files:
            hip_runs = km.all_labels.cpu().numpy().astype(np.int64)
            same = sum(matched_iou(hip_runs[r], g["restart_labels"][r].astype(np.int64), K)[0] >= 0.99 for r in range(10))
        line = f"window {w:2d} exact mode: IoU {iou:.4f} identical {ex:.4f}; restarts in place {same}/10; {dt * 1e3:.0f} ms;{extra}"
        print(line, flush=True)
        lines.append(line)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "exact_study.txt"), "w") as fh:
            fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
