"""How much of the mask disagreement with the reference is fp16 itself?  (CPU, build container or GPU box host.)

For the fixture windows of the headline clip (tests/golden/c2_window*.npz = the REFERENCE's fp32 labels) run the CPU oracle
twice on the same inputs: in fp32 (must reproduce the reference's taps to 1e-5 and, K-means being what it is, usually its
labels) and with every matmul operand and every stored activation rounded to fp16 (`UNetOracle(round_bf16="f16")`: what ANY
fp16 evaluation of the network -- the reference's own CUDA autocast included -- does to the features), and compare the masks
with the reference's.  The fp16-rounded oracle shares no code with the HIP path; if its masks scatter like the HIP path's, the
scatter is a property of best-of-10 K-means on 1e-3-perturbed features, not of the kernels.

    python tools/mask_rounding_study.py [--windows 0 1 2] [--modes f16 f32]     -> gpurun_out/mask_rounding_study.txt
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
from oracle import pipeline as OP  # noqa: E402
from oracle.unet import UNetOracle  # noqa: E402
from tools_metrics import matched_iou  # noqa: E402
from vidseg_diffusion_amd import synthetic  # noqa: E402

F, LAT, K = 14, 64, 20


class Chunked(UNetOracle):
    """The same network evaluated 7 samples at a time (every operator is per-sample; bounds the attention matrices)."""

    def forward(self, x, timesteps, context, y=None, num_video_frames=None):
        outs, taps = [], {}
        for i in range(0, x.shape[0], 7):
            outs.append(super().forward(x[i:i + 7], timesteps[i:i + 7], context[i:i + 7]))
            for k, v in self.taps.items():
                taps.setdefault(k, []).append(v)
        self.taps = {k: torch.cat(v, 0) for k, v in taps.items()}
        return torch.cat(outs, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, nargs="*", default=None)
    ap.add_argument("--modes", nargs="*", default=["f16"])
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    from vidseg_diffusion_amd.unet import UNetModel
    cfg = dict(synthetic.SD21_FULL)
    shapes = {k: tuple(v.shape) for k, v in UNetModel(**cfg).state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1234, zero_gain=synthetic.HEADLINE["zero_gain"]).items()}
    c, uc = synthetic.sd_conditioning(F, context_dim=cfg["context_dim"], seq=77, seed=1)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    lines = []
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "c2_window*.npz"))):
        g = np.load(path)
        w = int(g["window_id"]) if "window_id" in g.files else 0
        if args.windows is not None and w not in args.windows:
            continue
        lat = synthetic.headline_latent(F, LAT, LAT, window_id=w)
        noise = torch.randn((F, 4, LAT, LAT), generator=torch.Generator().manual_seed(100 + w))
        assert synthetic.sha256_of(lat) == str(g["latent_sha256"])
        for mode in args.modes:
            t0 = time.time()
            net = Chunked(sd, round_bf16=False if mode == "f32" else mode)
            res = OP.segment_window(net, torch.from_numpy(lat), torch.from_numpy(c), torch.from_numpy(uc), noise, num_masks=K, t_start=22,
                                    seed=17, is_refine_mask=True)
            iou, ex = matched_iou(res["match_labels"], g["match_labels"].astype(np.int64), K)
            iou2, ex2 = matched_iou(res["labels"], g["corrected_labels"].astype(np.int64), K)
            line = (f"window {w} oracle[{mode}] vs reference fp32: Step 3 IoU {iou:.4f} identical {ex:.4f}; Step 3b IoU {iou2:.4f} identical {ex2:.4f} "
                    f"({time.time() - t0:.0f} s)")
            print(line, flush=True)
            lines.append(line)
            with open(os.path.join(out_dir, "mask_rounding_study.txt"), "w") as fh:
                fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
