"""Does a per-step precision plan keep the reference's masks?  (CPU, build container)

The all-exact mode (every layer of every step on split operands) reproduces the reference's masks; the 16-bit mode does not.
This study runs the CPU oracle on the 16 fixture windows with a precision PER SAMPLER STEP -- e.g. steps 22 and 23 with every
matmul operand and stored activation rounded to fp16 (what the 16-bit HIP mode does) and step 24 in fp32 -- and reports the
step-24 tap error against the REFERENCE's taps (tests/golden/c2_window*.npz) and the Step-3 / Step-3b mask IoU.

    python tools/mixed_plan_study.py --plan f16 f16 f32 [--windows 0 1 2]     -> gpurun_out/mixed_plan_<plan>.txt
"""
import argparse
import glob
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


class Planned(Chunked):
    """The oracle with its rounding mode switched per network evaluation (one evaluation per sampler step)."""

    def __init__(self, sd, plan):
        super().__init__(sd, round_bf16=False)
        self.plan, self.calls = list(plan), 0

    def forward(self, x, timesteps, context, y=None, num_video_frames=None):
        mode = self.plan[self.calls]
        self.calls += 1
        m = False if mode == "f32" else mode
        self.rb = self.rb_w = self.rb_ln = self.rb_res = m
        return super().forward(x, timesteps, context)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, nargs="*", default=None)
    ap.add_argument("--plan", nargs=3, default=["f16", "f16", "f32"])
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    torch.set_grad_enabled(False)
    from vidseg_diffusion_amd.unet import UNetModel
    cfg = dict(synthetic.SD21_FULL)
    shapes = {k: tuple(v.shape) for k, v in UNetModel(**cfg).state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1234, zero_gain=synthetic.HEADLINE["zero_gain"]).items()}
    c, uc = synthetic.sd_conditioning(F, context_dim=cfg["context_dim"], seq=77, seed=1)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    tag = "_".join(args.plan)
    lines, ious = [], []
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "c2_window*.npz"))):
        g = np.load(path)
        w = int(g["window_id"]) if "window_id" in g.files else 0
        if args.windows is not None and w not in args.windows:
            continue
        lat = synthetic.headline_latent(F, LAT, LAT, window_id=w)
        noise = torch.randn((F, 4, LAT, LAT), generator=torch.Generator().manual_seed(100 + w))
        t0 = time.time()
        net = Planned(sd, args.plan)
        res = OP.segment_window(net, torch.from_numpy(lat), torch.from_numpy(c), torch.from_numpy(uc), noise, num_masks=K, t_start=22,
                                seed=17, is_refine_mask=True)
        errs = []
        if "q6_sub" in g.files:
            for b in (6, 7, 8):
                q = res["q_taps"][b][F:].astype(np.float64)[:, ::16, ::2]
                ref = g[f"q{b}_sub"].astype(np.float64)
                errs.append(float(np.linalg.norm(q - ref) / np.linalg.norm(ref)))
        xerr = float(np.linalg.norm(res["x_final"].numpy().astype(np.float64) - g["x_final"]) / np.linalg.norm(g["x_final"])) if "x_final" in g.files else -1
        iou, ex = matched_iou(res["match_labels"], g["match_labels"].astype(np.int64), K)
        iou2, ex2 = matched_iou(res["labels"], g["corrected_labels"].astype(np.int64), K)
        ious.append(iou2)
        line = (f"window {w} plan {tag}: taps 6/7/8 " + " ".join(f"{e:.2e}" for e in errs) + f" x_final {xerr:.2e}; Step 3 IoU {iou:.4f} identical {ex:.4f}; "
                f"Step 3b IoU {iou2:.4f} identical {ex2:.4f} ({time.time() - t0:.0f} s)")
        print(line, flush=True)
        lines.append(line)
        summ = f"plan {tag}: {len(ious)} windows, mean IoU {np.mean(ious):.4f}, at >= 0.99: {sum(i >= 0.99 for i in ious)}"
        with open(os.path.join(out_dir, f"mixed_plan_{tag}.txt"), "w") as fh:
            fh.write("\n".join(lines + [summ]) + "\n")
    print(summ)


if __name__ == "__main__":
    main()
