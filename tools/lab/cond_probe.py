"""How well-conditioned is the Step-3 clustering of a synthetic window?  (GPU box.)

For every candidate input design the full-size SD window (BASELINE configs[1]) is segmented three ways on the HIP path:
  full   -- the reference's schedule (3 CFG evaluations, batch 28)
  half   -- masks_only: same arithmetic, another fp32 summation order (~1e-3 normalised rms on the taps)
and the label maps are written to gpurun_out/cond_probe_<act>.npz; run once per build (VIDSEG_ACT=f16 / bf16) and compare
with `--compare` (bf16 is a ~1e-2 perturbation of the taps).  Prints matched IoU / identical fraction between the variants
and the agreement of `full` with the generating partition (region designs).

    python tools/lab/cond_probe.py            # writes gpurun_out/cond_probe_<act>.npz
    python tools/lab/cond_probe.py --compare  # f16 vs bf16 files
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tools_metrics import matched_iou  # noqa: E402

DESIGNS = [
    ("lat20_a2.0", dict(kind="region", R=20, amp=2.0, noise=0.05, protos="lattice")),
    ("blk20_a2.0", dict(kind="region", R=20, amp=2.0, noise=0.05, protos="lattice", layout="blocks")),
    ("blk20_a3.0", dict(kind="region", R=20, amp=3.0, noise=0.05, protos="lattice", layout="blocks")),
    ("blk20_a2.0_n0", dict(kind="region", R=20, amp=2.0, noise=0.0, protos="lattice", layout="blocks")),
]
ZERO_GAINS = [float(v) for v in os.environ.get("PROBE_ZERO_GAINS", "0.3").split(",")]
TAG = os.environ.get("PROBE_TAG", "")


def make_latent(d, F, lat):
    from vidseg_diffusion_amd import synthetic
    if d["kind"] == "blob":
        return synthetic.latent_clip(F, lat, lat, seed=1), None
    if d["kind"] == "scene":
        x = synthetic.scene_clip(F, lat, lat, num_objects=d["R"], cells=d["cells"], seed=1, amp=d["amp"], noise=d["noise"])
        gt = synthetic.scene_labels(F, lat, lat, d["R"], d["cells"], 1 + 4000)
    else:
        x = synthetic.region_clip(F, lat, lat, num_regions=d["R"], seed=1, amp=d["amp"], noise=d["noise"], protos=d.get("protos", "random"),
                                  layout=d.get("layout", "voronoi"))
        gt = (synthetic.block_labels if d.get("layout") == "blocks" else synthetic.region_labels)(F, lat, lat, d["R"], 1 + 4000)
    return x, gt[:, ::2, ::2].reshape(F, -1)                       # token grid = latent / 2 (top-left latent pixel of each token)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--compare", action="store_true")
    ap.add_argument("--masks", type=int, default=20)
    args = ap.parse_args()
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    if args.compare:
        import glob
        files = sorted(glob.glob(os.path.join(out_dir, "cond_probe_*.npz")))
        data = {os.path.basename(f)[11:-4]: np.load(f) for f in files}
        names = sorted(data)
        for key in data[names[0]].files:
            if not key.endswith("_full"):
                continue
            worst = (2.0, 2.0, "")
            for i in range(len(names)):
                for j in range(i + 1, len(names)):
                    if key in data[names[j]].files:
                        iou, ex = matched_iou(data[names[i]][key], data[names[j]][key], args.masks)
                        worst = min(worst, (iou, ex, f"{names[i]} vs {names[j]}"))
            print(f"{key:30s} worst pair over {names}: IoU {worst[0]:.4f} identical {worst[1]:.4f} ({worst[2]})")
        return
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd import ops, synthetic
    from vidseg_diffusion_amd.pipeline import build_sd_engine, segment_window
    from vidseg_diffusion_amd.unet import UNetModel
    dev = torch.device("cuda:0")
    F, LAT, K = 14, 64, args.masks
    cfg = dict(synthetic.SD21_FULL)
    c, uc = synthetic.sd_conditioning(F, context_dim=cfg["context_dim"], seq=77, seed=1)
    c, uc = {"crossattn": torch.from_numpy(c).to(dev)}, {"crossattn": torch.from_numpy(uc).to(dev)}
    noise = torch.randn((F, 4, LAT, LAT), generator=torch.Generator().manual_seed(100)).to(dev)
    act = "f16" if ops.act_dtype() == torch.float16 else "bf16"
    rec = {}
    for zg in ZERO_GAINS:
        net = UNetModel(**cfg)
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1234, zero_gain=zg).items()})
        net.pack(dev)
        eng = build_sd_engine(net, num_steps=25, scale=5.0)
        for name, d in DESIGNS:
            lat, gt = make_latent(d, F, LAT)
            lat = torch.from_numpy(lat).to(dev)
            res = {}
            for variant, mo in (("full", False), ("half", True), ("gn64", False)):
                FE.FeatureStore.clear()
                FE.MaskStore.clear()
                ops._GN_RPC_FORCE = 64 if variant == "gn64" else 0     # another summation order of the GroupNorm statistics
                labels, _ = segment_window(eng, lat, c, uc, num_masks=K, num_steps=25, t_start=22, seed=17, noise=noise,
                                           keep_all_steps=False, masks_only=mo, feature_folder="/nonexistent/probe", exp_name="p")
                res[variant] = np.asarray(labels).reshape(F, -1)
                rec[f"zg{zg}_{name}_{variant}"] = res[variant].astype(np.int16)
            ops._GN_RPC_FORCE = 0
            iou3, ex3 = matched_iou(res["full"], res["gn64"], K)
            iou, ex = matched_iou(res["full"], res["half"], K)
            sizes = np.sort(np.bincount(res["full"].reshape(-1), minlength=K))[::-1]
            line = f"zg={zg:<4} {name:12s} [{act}] full vs half: IoU {iou:.4f} identical {ex:.4f}; vs gn64: IoU {iou3:.4f} identical {ex3:.4f}; sizes {sizes[:3]}..{sizes[-3:]}"
            if gt is not None:
                giou, gex = matched_iou(res["full"], gt, max(K, d["R"]))
                line += f"; vs generating partition IoU {giou:.3f} identical {gex:.3f}"
            print(line, flush=True)
        del net, eng
        torch.cuda.empty_cache()
    np.savez_compressed(os.path.join(out_dir, f"cond_probe_{act}{TAG}.npz"), **rec)


if __name__ == "__main__":
    main()
