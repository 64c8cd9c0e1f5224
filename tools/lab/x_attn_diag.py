"""Where does k_x_attention_mfma lose accuracy?  (GPU box)  python tools/lab/x_attn_diag.py
1. the split GEMM on all-positive operands of growing K (does a long MFMA accumulation chain into a large accumulator keep fp32 accuracy?)
2. the attention error against float64 by key count (round 3 ran it with the PV products accumulated across tiles inside the MFMA
   and with the per-tile flush that the kernel kept)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vidseg_diffusion_amd import exact as X, ops  # noqa: E402

dev = torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal(shape).astype(np.float32)) * scale


def rel(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / np.abs(ref).max()), float(np.sqrt(((got - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean()))


for K in (64, 320, 1280, 5120, 20480):
    for positive in (False, True):
        a, w = rnd((256, K), 3), rnd((320, K), 4)
        if positive:
            a, w = a.abs(), w.abs()
        out = X.linear_x(X.split3(a.to(dev)), X.pack_linear_x(w, dev)).cpu()
        ref = a.double() @ w.double().t()
        f32 = (a.to(dev) @ w.to(dev).t()).cpu()
        print(f"split GEMM K={K} positive={positive}: max/rms err {rel(out, ref)[0]:.2e} {rel(out, ref)[1]:.2e}   (torch fp32 matmul {rel(f32, ref)[1]:.2e})", flush=True)

for (B, H, Nq, Nk) in ((1, 2, 128, 64), (1, 2, 128, 128), (1, 2, 128, 256), (1, 2, 128, 512), (1, 2, 128, 1024), (1, 2, 128, 2048), (1, 2, 128, 4096), (1, 10, 1024, 1024)):
    C = H * 64
    q, k, vv = rnd((B, Nq, C), 19), rnd((B, Nk, C), 20), rnd((B, Nk, C), 21)
    kv = torch.cat([k, vv], -1).to(dev)
    out = X.attention_mfma(q.to(dev), kv, H, B, Nq, Nk).cpu()
    hd = lambda t, n: t.double().view(B, n, H, 64).transpose(1, 2)            # noqa: E731
    ref = TF.scaled_dot_product_attention(hd(q, Nq), hd(k, Nk), hd(vv, Nk)).transpose(1, 2).reshape(B, Nq, C)
    f32 = X.attention_f32(q.to(dev), kv[..., :C], kv[..., C:], H, B, Nq, Nk).cpu()
    d = (out.double() - ref).abs()
    idx = np.unravel_index(int(d.argmax()), d.shape)
    print(f"attention B={B} H={H} Nq={Nq} Nk={Nk}: mfma max/rms {rel(out, ref)[0]:.2e} {rel(out, ref)[1]:.2e}  f32 kernel {rel(f32, ref)[0]:.2e} {rel(f32, ref)[1]:.2e}"
          f"  worst at {idx}, fraction of elements over 2e-6*max: {float((d > 2e-6 * ref.abs().max()).double().mean()):.4f}", flush=True)
