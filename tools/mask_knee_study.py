"""How small must the feature error be for best-of-10 K-means to return the reference's masks?  (CPU, build container only.)

Input: the REFERENCE's own fp32 step-24 Q taps of decoder blocks 6/7/8 for the fixture windows (written by
`C2_TAP_CACHE=/tmp/vidseg_taps python tools/gen_golden_c2_window.py --windows 0-15`; 110 MB per window, never committed) and
the committed fixtures tests/golden/c2_window*.npz (the reference's labels and all ten restarts).

For every window and every noise level eps: add white noise of normalised rms eps to the fp32 taps (per block: sigma = eps *
rms(block)), round to fp16 like the dump does, and run Steps 3 of the reference's analysis -- fp16 3-block mean, max-abs
normalise, sklearn KMeans(20, n_init=10) under np.random.seed(17), predict(frame 0), 4-NN propagation (FE:546-613; sklearn is the
reference's own arithmetic) -- then compare with the reference's labels.  Also recorded per (window, eps): for each of the ten
restarts, whether the perturbed restart still lands in the reference restart's clustering (matched IoU >= 0.99) and the relative
change of its inertia; that is the "measured effect of an eps perturbation" the restart-equivalence test's threshold comes from.

    python tools/mask_knee_study.py [--windows 0-15] [--eps 0 1e-5 1e-4 3e-4 1e-3] [--seeds 2]   -> profiles/r03_mask_knee_study.txt
"""
import argparse
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from tools_metrics import matched_iou  # noqa: E402

F, N, C, K = 14, 1024, 640, 20


def analyse(taps16):
    """FE:739-748 + FE:546-613 on fp16 cond-half taps {6,7,8: [F,N,C]} with sklearn, all restarts observed."""
    import torch
    from sklearn.cluster import KMeans
    from sklearn.neighbors import KNeighborsClassifier
    from gen_golden_c2_window import RestartRecorder
    agg = torch.mean(torch.stack([torch.from_numpy(taps16[b]) for b in (8, 7, 6)]), dim=0).numpy()          # fp16 mean, FE:745
    feat = agg / np.max(np.abs(agg), axis=-1, keepdims=True)                                                  # FE:554-555 (fp16)
    flat = feat.reshape(-1, C)
    np.random.seed(17)
    with RestartRecorder() as rec:
        km = KMeans(n_clusters=K, n_init=10).fit(flat)                                                         # FE:562-570
    fake = km.predict(feat[0])                                                                                 # FE:572
    knn = KNeighborsClassifier(n_neighbors=4).fit(feat[0], fake)                                               # FE:608-612 (identity label map)
    return knn.predict(flat), rec.runs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", default="0-15")
    ap.add_argument("--eps", type=float, nargs="*", default=[0.0, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3])
    ap.add_argument("--seeds", type=int, default=2)
    ap.add_argument("--cache", default=os.environ.get("C2_TAP_CACHE", "/tmp/vidseg_taps"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_mask_knee_study.txt"))
    args = ap.parse_args()
    a, _, b = args.windows.partition("-")
    wids = list(range(int(a), int(b or a) + 1))
    lines = ["# tools/mask_knee_study.py: white noise of normalised rms eps added to the REFERENCE's fp32 step-24 Q taps (blocks 6/7/8), then the",
             "# reference's Step 3 (fp16 dump, 3-block mean, normalise, sklearn KMeans n_init=10 seed 17, predict, 4-NN) vs the reference's labels",
             "# per line: window, eps, noise seed, final-mask IoU / identical fraction; restarts that stay in their reference clustering (IoU>=0.99)",
             "#           of 10; max relative inertia change among those; does the best restart change (index old->new)"]
    summary = {}
    for w in wids:
        gpath = os.path.join(ROOT, "tests", "golden", "c2_window.npz" if w == 0 else f"c2_window_w{w}.npz")
        tpath = os.path.join(args.cache, f"taps_w{w}.npz")
        if not (os.path.exists(gpath) and os.path.exists(tpath)):
            continue
        g, t = np.load(gpath), np.load(tpath)
        taps32 = {b: t[f"q{b}"] for b in (6, 7, 8)}
        ref_labels, ref_runs = g["match_labels"].astype(np.int64).reshape(-1), g["restart_labels"].astype(np.int64)
        ref_inertia, ref_best = g["restart_inertia"], int(g["restart_best"])
        for eps in args.eps:
            for s in range(1 if eps == 0 else args.seeds):
                t0 = time.time()
                rng = np.random.Generator(np.random.PCG64(1000 * w + s))
                taps16 = {}
                for b, q in taps32.items():
                    sig = eps * float(np.sqrt(np.mean(q.astype(np.float64) ** 2)))
                    taps16[b] = (q + (rng.standard_normal(q.shape, dtype=np.float32) * np.float32(sig) if eps else 0)).astype(np.float16)
                labels, runs = analyse(taps16)
                iou, exact = matched_iou(labels, ref_labels, K)
                same, drel = 0, 0.0
                for r in range(10):
                    if matched_iou(runs[r][0].astype(np.int64), ref_runs[r], K)[0] >= 0.99:
                        same += 1
                        drel = max(drel, abs(runs[r][1] / ref_inertia[r] - 1.0))
                best = int(np.argmin([r[1] for r in runs]))
                line = (f"window {w:2d} eps {eps:.0e} seed {s}: IoU {iou:.4f} identical {exact:.4f}; restarts in place {same}/10, "
                        f"max |dJ/J| {drel:.2e}; best {ref_best}->{best}  ({time.time() - t0:.0f} s)")
                print(line, flush=True)
                lines.append(line)
                summary.setdefault(eps, []).append((iou, same, drel))
        with open(args.out, "w") as fh:
            fh.write("\n".join(lines) + "\n")
    lines.append("# summary: eps -> windows x seeds, mean / median / min IoU, share with IoU >= 0.99, mean restarts in place, max |dJ/J| (in-place restarts)")
    for eps, v in sorted(summary.items()):
        ious = np.array([x[0] for x in v])
        lines.append(f"eps {eps:.0e}: n {len(v)}, IoU mean {ious.mean():.4f} median {np.median(ious):.4f} min {ious.min():.4f}, >=0.99: "
                     f"{np.mean(ious >= 0.99):.2f}, restarts in place {np.mean([x[1] for x in v]):.1f}/10, max |dJ/J| {max(x[2] for x in v):.2e}")
    with open(args.out, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines[-len(summary) - 1:]))


if __name__ == "__main__":
    main()
