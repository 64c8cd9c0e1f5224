"""The exact mode's self-attention (k_x_attention_mfma) at the window's sizes: time per launch, executed MFMA rate (three fp16 products per
contraction), error against float64 on one head, and a digest of the output for same-box A/Bs between two builds of the library
(VIDSEG_LIB=libvidseg_ab_NAME.so, tools/build_exp.py).        python tools/xattn_bench.py [--reps 20]"""
import argparse
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vidseg_diffusion_amd import exact  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--first", action="store_true", help="the 64 x 64 level only (PMC passes)")
args = ap.parse_args()
dev = torch.device("cuda:0")
tag = os.environ.get("VIDSEG_LIB", "libvidseg_hip.so") + " " + os.environ.get("VIDSEG_ATTN", "")
SHAPES = ((28, 5, 4096), (14, 5, 4096), (28, 10, 1024), (28, 20, 256), (2, 5, 4000))
for B, H, N in SHAPES[:1] if args.first else SHAPES:
    C = H * 64
    g = torch.Generator().manual_seed(1)
    qkv = (torch.randn((B, N, 3 * C), generator=g) * 1.5).to(dev)
    q, kv = qkv[..., :C], qkv[..., C:]
    for _ in range(3):
        o = exact.attention_mfma(q, kv, H, B, N, N, split_out=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        o = exact.attention_mfma(q, kv, H, B, N, N, split_out=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.reps
    hi, lo = exact.split_planes(kv)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.reps):
        exact.split_planes(kv)
    e1.record()
    torch.cuda.synchronize()
    us_split = e0.elapsed_time(e1) * 1e3 / args.reps
    f32 = exact.attention_mfma(q, kv, H, B, N, N, split_out=False)
    qd, kd, vd = (t[0, :, :64].double() for t in (q, kv[..., :C], kv[..., C:]))
    ref = torch.softmax(qd @ kd.T * 0.125, dim=-1) @ vd
    err = float((f32[0, :, :64].double() - ref).norm() / ref.norm())
    img = o[0, :, :].float()
    rec = img[:, :64] + img[:, C:C + 64]
    err3 = float((rec.double() - ref).norm() / ref.norm())
    dig = hashlib.sha1(o.cpu().numpy().tobytes()).hexdigest()[:12]
    fl = 3 * 4.0 * B * H * N * N * 64
    print(f"x_attention B={B} H={H} N={N}: {us:8.1f} us (of which K/V plane split {us_split:6.1f})  {fl / (us - us_split) / 1e6:7.1f} TFLOP/s executed  "
          f"nrms vs f64 {err:.2e} (split image {err3:.2e})  sha1 {dig}  [{tag.strip()}]", flush=True)
