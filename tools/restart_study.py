"""GPU box: for every fixture window of the headline clip (tests/golden/c2_window*.npz with the reference's ten K-means restarts)
run the HIP path and record how its ten restarts relate to the reference's -- the data behind tests/test_gpu_c2_window.py's
restart-equivalence assertion.  Writes gpurun_out/restart_study.{txt,npz}.

    python tools/restart_study.py [--windows 0-15]
"""
import argparse
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tools_metrics import clustering_objective, matched_iou  # noqa: E402
from vidseg_diffusion_amd import synthetic  # noqa: E402

F, LAT, K = 14, 64, 20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", default="0-15")
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
    net.pack(dev)
    eng = build_sd_engine(net, num_steps=25, scale=5.0)
    c, uc = synthetic.sd_conditioning(F, context_dim=cfg["context_dim"], seq=77, seed=1)
    cc, ucc = {"crossattn": torch.from_numpy(c).to(dev)}, {"crossattn": torch.from_numpy(uc).to(dev)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    lines, rec = [], {}
    for w in wids:
        path = os.path.join(ROOT, "tests", "golden", "c2_window.npz" if w == 0 else f"c2_window_w{w}.npz")
        if not os.path.exists(path):
            continue
        g = np.load(path)
        if "restart_labels" not in g.files:
            continue
        lat = synthetic.headline_latent(F, LAT, LAT, window_id=w)
        noise = torch.randn((F, 4, LAT, LAT), generator=torch.Generator().manual_seed(100 + w))
        assert synthetic.sha256_of(lat) == str(g["latent_sha256"])
        FE.FeatureStore.clear()
        FE.MaskStore.clear()
        labels, _ = segment_window(eng, torch.from_numpy(lat).to(dev), cc, ucc, num_masks=K, num_steps=25, t_start=22, seed=17,
                                   noise=noise.to(dev), feature_folder="/nonexistent/rs", exp_name=f"w{w}", keep_all_steps=False)
        km = A.LAST_KMEANS
        hip_runs = km.all_labels.cpu().numpy().astype(np.int64)
        hip_j = np.asarray(km.all_inertia, dtype=np.float64)
        hb = int(km.best_restart)
        ref_runs, ref_j, rb = g["restart_labels"].astype(np.int64), g["restart_inertia"], int(g["restart_best"])
        iou_final = matched_iou(labels, g["match_labels"].astype(np.int64), K)[0]
        same_idx = np.array([matched_iou(hip_runs[r], ref_runs[r], K)[0] for r in range(10)])
        win_vs_ref = np.array([matched_iou(hip_runs[hb], ref_runs[r], K)[0] for r in range(10)])
        st = FE.FeatureStore.folder("/nonexistent/rs", f"w{w}")
        from oracle import analysis as OA
        taps = {bk: st[f"output_block_{bk}_spatial_self_attn_q_time_24"].cpu().numpy() for bk in (6, 7, 8)}
        flat = OA.normalize_tokens(OA.aggregate_blocks([taps[8], taps[7], taps[6]])[F:]).reshape(-1, 640)
        j_hip_on_hip = clustering_objective(flat, hip_runs[hb])
        j_refbest_on_hip = clustering_objective(flat, ref_runs[rb])
        r2 = int(np.argmax(win_vs_ref))
        line = (f"window {w:2d}: final IoU {iou_final:.4f} | HIP best r{hb} J {hip_j[hb]:.1f}; ref best r{rb} J {ref_j[rb]:.1f} | HIP winner ~ ref restart r{r2} "
                f"(IoU {win_vs_ref[r2]:.4f}, ref gap (J_r-J_best)/J_best {(ref_j[r2] - ref_j[rb]) / ref_j[rb]:.2e}) | restarts in place "
                f"{int(np.sum(same_idx >= 0.99))}/10 | on HIP features: J(HIP winner) / J(ref winner's labels) = {j_hip_on_hip / j_refbest_on_hip:.5f}")
        print(line, flush=True)
        lines.append(line)
        lines.append("           HIP J/J_ref per restart: " + " ".join(f"{hip_j[r] / ref_j[r]:.5f}" for r in range(10)))
        lines.append("           same-index IoU:          " + " ".join(f"{v:.3f}" for v in same_idx))
        rec[f"w{w}_hip_j"], rec[f"w{w}_same_idx"], rec[f"w{w}_win_vs_ref"] = hip_j, same_idx, win_vs_ref
        rec[f"w{w}_iou_final"], rec[f"w{w}_hb"], rec[f"w{w}_jratio_on_hip"] = iou_final, hb, j_hip_on_hip / j_refbest_on_hip
        with open(os.path.join(ROOT, "gpurun_out", "restart_study.txt"), "w") as fh:
            fh.write("\n".join(lines) + "\n")
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "restart_study.npz"), **rec)


if __name__ == "__main__":
    main()
