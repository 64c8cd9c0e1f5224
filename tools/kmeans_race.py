"""Is kmeans_fit bit-stable while another stream keeps the chip busy?  (GPU box)  python tools/kmeans_race.py [--iters 12] [--quiet]"""
import argparse
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vidseg_diffusion_amd import analysis as A, ops, synthetic  # noqa: E402
A.KEEP_LAST = True

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=12)
ap.add_argument("--quiet", action="store_true")
ap.add_argument("--same-stream", action="store_true", help="background work on the SAME stream (ordering only, no concurrency)")
args = ap.parse_args()
dev = torch.device("cuda:0")
F, h, w, C, K = 14, 32, 32, 640, 20
blocks, _ = synthetic.attention_q_dumps(F, h, w, C, num_blocks=3, seed=7)
_, feat = A.mean_normalize([torch.from_numpy(b).to(dev) for b in blocks], F * h * w, F * h * w)
ad = ops.act_dtype()
a = torch.randn(114688, 320, device=dev).to(ad)
wt = ops.pack_linear(torch.randn(320, 2880) * 0.02, dev) if False else None
x0 = torch.randn(28, 64, 64, 320, device=dev).to(ad)
wc = ops.pack_conv3x3(torch.randn(320, 320, 3, 3) * 0.02, dev)
bc = torch.zeros(320, device=dev)
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
seen = None
for it in range(args.iters):
    if not args.quiet:
        for _ in range(120):                                   # ~25 ms of full-chip GEMMs queued on the main stream
            ops.conv3x3(x0, wc, bc)
    st = main if args.same_stream else side
    with torch.cuda.stream(st):
        if not args.same_stream:
            st.wait_stream(main) if it % 2 == 0 else None      # every other run: the K-means is queued BEHIND the GEMMs' completion
        np.random.seed(17)
        km = A.kmeans_fit(feat, K)
        sig = (tuple(km.all_n_iter), hashlib.sha256(km.all_labels.cpu().numpy().tobytes()).hexdigest()[:10],
               tuple(f"{v:.17g}" for v in km.all_inertia))
    torch.cuda.synchronize()
    if seen is None:
        seen = sig
    print(f"run {it}: n_iter {list(sig[0])} labels {sig[1]} seeds {hashlib.sha256(A.LAST_CENTER_IDS.cpu().numpy().tobytes()).hexdigest()[:8]} per restart {[hashlib.sha256(r.tobytes()).hexdigest()[:4] for r in A.LAST_CENTER_IDS.cpu().numpy().reshape(10, -1)]} {'same' if sig == seen else 'DIFFERS'}", flush=True)
