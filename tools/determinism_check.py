"""Run-to-run determinism of the headline window (GPU box): sha256 of the step-24 Q taps of blocks 6 / 7 / 8, of the final latent and of
the masks, for a few fixture windows, each run twice in this process; run the script twice and diff the outputs for process-to-process
determinism.     python tools/determinism_check.py [--windows 0-3] [--precision exact]"""
import argparse
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vidseg_diffusion_amd import synthetic  # noqa: E402

F, LAT, K = 14, 64, 20


def h(t):
    a = t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", default="0-3")
    ap.add_argument("--precision", default="fp16")
    ap.add_argument("--overlap", action="store_true", help="through WindowPipeline (analysis on the second stream), as bench.py runs it")
    ap.add_argument("--prof", action="store_true", help="with the GEMM timing events on, as in bench.py's timed region (VIDSEG_GEMM=ext=0/1 picks the mechanism)")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--lanes", type=int, default=1, help="--overlap: feature passes in flight (bench.py's headline runs 2)")
    ap.add_argument("--masks-only", action="store_true", help="the parity mode's pruned last step (bench.py's headline)")
    ap.add_argument("--refine", action="store_true", help="with Step 3b (dense tracking + trajectory vote)")
    ap.add_argument("--chain", action="store_true", help="windows chained as one clip (4-NN label propagation instead of K-means after window 0)")
    args = ap.parse_args()
    a, _, b = args.windows.partition("-")
    wids = list(range(int(a), int(b or a) + 1))
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import WindowPipeline, build_sd_engine, segment_window
    from vidseg_diffusion_amd.unet import UNetModel
    dev = torch.device("cuda:0")
    cfg = dict(synthetic.SD21_FULL)
    net = UNetModel(**cfg)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1234, zero_gain=synthetic.HEADLINE["zero_gain"]).items()})
    if args.precision == "exact":
        net.set_precision("exact")
    eng = build_sd_engine(net, num_steps=25, scale=5.0)
    c, uc = synthetic.sd_conditioning(F, context_dim=cfg["context_dim"], seq=77, seed=1)
    cc, ucc = {"crossattn": torch.from_numpy(c).to(dev)}, {"crossattn": torch.from_numpy(uc).to(dev)}
    from vidseg_diffusion_amd import ops
    seen = {}
    for rep in range(args.reps):
        if args.prof:
            ops.gemm_profile_begin()
        if args.overlap:
            pipe = WindowPipeline(eng, chain=args.chain, lanes=args.lanes, num_masks=K, is_aggre_attn=True, is_refine_mask=args.refine)
            outs = []
            for w in wids:
                lat = torch.from_numpy(synthetic.headline_latent(F, LAT, LAT, window_id=w)).to(dev)
                noise = torch.randn((F, 4, LAT, LAT), generator=torch.Generator().manual_seed(100 + w)).to(dev)
                got = pipe.push(lat, cc, ucc, keep_all_steps=False, exp_name=f"w{w}", noise=noise, num_steps=25, t_start=22, seed=17,
                                feature_folder="/nonexistent/det", masks_only=args.masks_only)
                if got is not None:
                    outs.append(got)
            outs += pipe.drain()
            for w, lab in zip(wids, outs):
                line = f"window {w}: masks {h(lab)}"
                flag = "" if seen.setdefault(w, line) == line else "   <-- DIFFERS from rep 0"
                print(f"rep {rep} {line}{flag}", flush=True)
            if args.prof:
                ops.gemm_profile_end()
            continue
        for w in wids:
            lat = torch.from_numpy(synthetic.headline_latent(F, LAT, LAT, window_id=w)).to(dev)
            noise = torch.randn((F, 4, LAT, LAT), generator=torch.Generator().manual_seed(100 + w)).to(dev)
            FE.FeatureStore.clear()
            FE.MaskStore.clear()
            labels, _ = segment_window(eng, lat, cc, ucc, num_masks=K, num_steps=25, t_start=22, seed=17, noise=noise,
                                       feature_folder="/nonexistent/det", exp_name=f"w{w}", keep_all_steps=False)
            st = FE.FeatureStore.folder("/nonexistent/det", f"w{w}")
            taps = " ".join(f"q{bk} {h(st[f'output_block_{bk}_spatial_self_attn_q_time_24'])}" for bk in (6, 7, 8))
            line = f"window {w}: {taps} masks {h(labels)}"
            flag = "" if seen.setdefault(w, line) == line else "   <-- DIFFERS from rep 0"
            print(f"rep {rep} {line}{flag}", flush=True)
        if args.prof:
            ops.gemm_profile_end()


if __name__ == "__main__":
    main()
