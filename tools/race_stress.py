"""Bitwise stability of the GEMM / attention / norm kernels while another stream keeps the memory system and some CUs busy (GPU box).
Hand-counted vmcnt schedules read an LDS buffer as soon as their count says the DMA has landed: a count that is one short passes every
quiet run and fails when a load arrives late.  Every op is run `--iters` times beside a stream of large copies and small f64 kernels;
each result is compared bit for bit with the first.        python tools/race_stress.py [--iters 200]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vidseg_diffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--quiet", action="store_true", help="no background stream")
ap.add_argument("--exact", action="store_true", help="the exact mode's launches instead: split-operand conv / linear (k_gemm_p7x), the GEGLU projection on the "
                                                   "split tile (k_gemm_p7x<4, true>), the split-operand attention")
ap.add_argument("--twin", action="store_true", help="instead of the background stream: the SAME op on the SAME inputs on two HIP streams at once "
                                                  "(two window lanes / two sweep passes running in lockstep), both results compared with the first")
ap.add_argument("--unet-bg", action="store_true", help="instead of the background stream: full-size exact-mode UNet evaluations (batch 28, 64x64 latents) on "
                                                     "a second HIP stream while every op runs -- the aggressor mix of two window lanes")
ap.add_argument("--cross", action="store_true", help="instead of the background stream: while an op runs, a second HIP stream runs the OTHER ops of the "
                                                   "list (different kernels co-resident on the CUs: what two window lanes / two sweep passes do)")
args = ap.parse_args()
ad = ops.act_dtype()
B = 28
g = torch.Generator(device="cpu").manual_seed(3)


def rn(*shape, s=1.0):
    return (torch.randn(*shape, generator=g) * s).to(dev)


cases = []


def conv(H, Cin, Cout, C1=0, up=1, res=False):
    x0 = rn(B, H, H, Cin).to(ad)
    x1 = rn(B, H, H, C1).to(ad) if C1 else None
    w = ops.pack_conv3x3(rn(Cout, Cin + C1, 3, 3, s=0.02).cpu(), dev)
    b = rn(Cout)
    r = rn(B, H * up, H * up, Cout).to(ad) if res else None
    cases.append((f"conv H{H} {Cin}+{C1}->{Cout} up{up} res{int(res)}", lambda: ops.conv3x3(x0, w, b, x1=x1, up=up, residual=r)))


def lin(M, K, N, act=0, res=False):
    a = rn(M, K).to(ad)
    w = (ops.pack_geglu(rn(N, K, s=0.02).cpu(), torch.zeros(N), dev)[0] if act == 2 else ops.pack_linear(rn(N, K, s=0.02).cpu(), dev))
    b = rn(N)
    r = rn(M, N).to(ad) if res else None
    cases.append((f"linear M{M} K{K} N{N} act{act} res{int(res)}", lambda: ops.linear(a, w, b, act=act, residual=r)))


def attn(N, H):
    q, k, v = (rn(B, N, H * 64).to(ad) for _ in range(3))
    cases.append((f"attention N{N} H{H}", lambda: ops.attention(q, k, v, H)))


def xconv(H, Cin, Cout, up=1, stride=1, res=False, Bx=B):
    from vidseg_diffusion_amd import exact as X
    x3 = X.split3(rn(Bx, H, H, Cin))
    w3 = X.pack_conv3x3_x(rn(Cout, Cin, 3, 3, s=0.02).cpu(), dev)
    b = rn(Cout)
    Ho = (H * up + 2 - 3) // stride + 1
    r = rn(Bx, Ho, Ho, Cout) if res else None
    cases.append((f"exact conv B{Bx} H{H} {Cin}->{Cout} up{up} s{stride} res{int(res)}", lambda: X.conv3x3_x(x3, w3, b, up=up, stride=stride, residual=r)))


def xlin(M, K, N, res=False, geglu=False):
    from vidseg_diffusion_amd import exact as X
    a3 = X.split3(rn(M, K))
    if geglu:
        w3g, bg, grp = X.pack_geglu_x(rn(N, K, s=0.02).cpu(), rn(N).cpu(), dev)
        cases.append((f"exact GEGLU M{M} K{K} N{N} (groups of {grp})", lambda: X.geglu_linear_x(a3, w3g, bg, grp)))
        return
    w3 = X.pack_linear_x(rn(N, K, s=0.02).cpu(), dev)
    b = rn(N)
    r = rn(M, N) if res else None
    cases.append((f"exact linear M{M} K{K} N{N} res{int(res)}", lambda: X.linear_x(a3, w3, b, residual=r)))


def xattn(N, H, Bx=B):
    from vidseg_diffusion_amd import exact as X
    q, kv = rn(Bx, N, H * 64), rn(Bx, N, 2 * H * 64)
    cases.append((f"exact attention B{Bx} N{N} H{H}", lambda: X.attention_x(q, kv, H, Bx, N, N, split_out=True)))


def xtemporal(nv, S, H, T=14):
    from vidseg_diffusion_amd import exact as X
    qkv = rn(nv * T, S, 3 * H * 64)
    cases.append((f"exact temporal attention videos{nv} T{T} S{S} H{H}", lambda: X.temporal_attention_x(qkv, nv, T, S, H, split_out=True)))


def xblend(M, K, N):
    from vidseg_diffusion_amd import exact as X
    a3, w3, b, r, sp = X.split3(rn(M, K)), X.pack_linear_x(rn(N, K, s=0.02).cpu(), dev), rn(N), rn(M, N), rn(M, N)
    cases.append((f"exact linear + blend M{M} K{K} N{N}", lambda: X.linear_blend_x(a3, w3, b, r, sp, 0.4)))


def xrowvec(M, K, N, rows):
    from vidseg_diffusion_amd import exact as X
    a3, w3, b, r, rv = X.split3(rn(M, K)), X.pack_linear_x(rn(N, K, s=0.02).cpu(), dev), rn(N), rn(M, N), rn(M // rows, N)
    cases.append((f"exact linear + row vector + residual M{M} K{K} N{N}", lambda: X.linear_x(a3, w3, b, residual=r, rowvec=rv, rows_per_sample=rows)))


def xglue():
    from vidseg_diffusion_amd import exact as X
    for (Bx, H, C) in ((B, 8, 1280), (B, 16, 640), (B, 32, 640), (B, 64, 320), (B, 32, 1280), (B, 16, 1280)):
        x, gm, bt = rn(Bx, H, H, C), rn(C), rn(C)
        cases.append((f"exact GroupNorm+SiLU -> image B{Bx} H{H} C{C}", lambda x=x, gm=gm, bt=bt: X.groupnorm_split3(x, gm, bt, eps=1e-5, silu=True)))
        x1 = rn(Bx, H, H, C)
        g2, b2 = rn(2 * C), rn(2 * C)
        cases.append((f"exact GroupNorm(concat) -> image B{Bx} H{H} C{C}+{C}", lambda x=x, x1=x1, g2=g2, b2=b2: X.groupnorm_split3(x, g2, b2, x1=x1, eps=1e-5, silu=True)))
        t = x.view(Bx, H * H, C)
        cases.append((f"exact LayerNorm -> image M{Bx * H * H} C{C}", lambda t=t, gm=gm, bt=bt: X.layernorm_split3(t, gm, bt)))
        cases.append((f"exact split3 M{Bx * H * H} C{C}", lambda t=t: X.split3(t)))
        cases.append((f"exact split3_cat M{Bx * H * H} C{C}+{C}", lambda x=x, x1=x1: X.split3_cat(x, x1)))


def xcross(N, H, L=77, Bx=B):
    from vidseg_diffusion_amd import exact as X
    q, kv = rn(Bx, N, H * 64), rn(Bx, L, 2 * H * 64)
    cases.append((f"exact cross-attention B{Bx} N{N} L{L} H{H}", lambda: X.attention_x(q, kv, H, Bx, N, L, split_out=True)))


def xsmall():
    """the shapes of the 8 x 8 / 16 x 16 levels, the step-constant projections and the embedding GEMMs (round 6: the twin-stream
    instability of profiles/r06_e first showed in an 8 x 8 ResBlock)"""
    from vidseg_diffusion_amd import exact as X
    xconv(8, 2560, 1280); xconv(16, 1920, 1280); xconv(32, 1280, 640); xconv(32, 960, 640); xconv(64, 640, 320); xconv(16, 640, 1280); xconv(8, 1280, 1280, stride=1)
    xlin(1792, 1280, 1280, res=True); xlin(1792, 1280, 3840); xlin(1792, 5120, 1280, res=True); xlin(2156, 1024, 2560); xlin(2156, 1024, 640)
    xlin(28, 320, 1280); xlin(28, 1280, 1280); xlin(28, 1280, 21760)
    xattn(64, 20); xattn(256, 20)
    xcross(4096, 5); xcross(1024, 10); xcross(256, 20); xcross(64, 20)
    x, gm, bt = rn(B, 8, 8, 1280), rn(1280), rn(1280)
    t = x.view(B, 64, 1280)
    cases.append(("exact LayerNorm -> image M1792 C1280", lambda: X.layernorm_split3(t, gm, bt)))
    y = rn(1792, 10240)
    cases.append(("exact geglu_split3 M1792", lambda: X.geglu_split3(y)))


def xattn_raw(N, H, L=None, Bx=B):
    from vidseg_diffusion_amd import exact as X
    L = L or N
    q, kv = rn(Bx, N, H * 64), rn(Bx, L, 2 * H * 64)
    cases.append((f"exact attention, fp32 result (no split pass) B{Bx} N{N} L{L} H{H}", lambda: X.attention_x(q, kv, H, Bx, N, L, split_out=False)))


if args.exact and os.environ.get("ONLY_SMALL_ATTN") == "1":
    xattn_raw(64, 20); xattn_raw(64, 20, 77); xattn(64, 20); xcross(64, 20); xattn_raw(64, 20); xattn_raw(64, 20, 77)
    xconv = xlin = xtemporal = xblend = xrowvec = xattn = lambda *a, **k: None
    xglue = xsmall = lambda: None
if args.exact:
    xglue()
    xsmall()
    xtemporal(2, 2304, 5); xtemporal(2, 576, 10); xtemporal(1, 2304, 5); xblend(64512, 2560, 640); xblend(16128, 5120, 1280); xrowvec(64512, 640, 640, 2304)
    xattn(9216, 5, Bx=4)
    xconv(64, 320, 320, res=True); xconv(64, 960, 320); xconv(32, 640, 640, res=True); xconv(32, 1920, 640); xconv(16, 1280, 1280, res=True)
    xconv(16, 2560, 1280); xconv(8, 1280, 1280, res=True); xconv(32, 640, 640, up=2); xconv(64, 320, 320, stride=2); xconv(32, 640, 640, res=True, Bx=14)
    xlin(114688, 320, 320, res=True); xlin(114688, 320, 960); xlin(114688, 1280, 320, res=True); xlin(28672, 640, 640, res=True); xlin(28672, 2560, 640, res=True)
    xlin(7168, 1280, 1280, res=True); xlin(7168, 5120, 1280, res=True); xlin(3584, 1280, 1280, res=True); xlin(896, 1280, 1280, res=True)
    xlin(114688, 320, 2560, geglu=True); xlin(28672, 640, 5120, geglu=True); xlin(7168, 1280, 10240, geglu=True); xlin(1792, 1280, 10240, geglu=True)
    xlin(14336, 640, 5120, geglu=True); xlin(896, 1280, 10240, geglu=True)
    xconv(18, 1280, 1280, res=True); xconv(18, 2560, 1280); xlin(16128, 1280, 1280, res=True); xlin(16128, 5120, 1280, res=True)   # k_gemm_phx (256-row tile)
    xattn(4096, 5); xattn(1024, 10); xattn(1024, 10, Bx=14)
    conv = lin = attn = lambda *a, **k: None                  # the 16-bit cases below are skipped
conv(64, 320, 320, res=True); conv(64, 640, 320, 320); conv(32, 640, 640, res=True); conv(32, 1280, 640, 640); conv(16, 1280, 1280, res=True)
conv(16, 1280, 1280, 1280); conv(8, 1280, 1280, res=True); conv(32, 640, 640, up=2); conv(16, 1280, 1280, up=2)
lin(114688, 320, 320); lin(114688, 320, 320, res=True); lin(114688, 320, 960); lin(114688, 320, 2560, act=2); lin(114688, 1280, 320, res=True)
lin(28672, 640, 640, res=True); lin(28672, 640, 1920); lin(28672, 640, 5120, act=2); lin(28672, 2560, 640, res=True)
lin(7168, 1280, 1280, res=True); lin(7168, 1280, 3840); lin(7168, 1280, 10240, act=2); lin(7168, 5120, 1280, res=True)
lin(1792, 1280, 1280, res=True); lin(1792, 1280, 10240, act=2); lin(1792, 5120, 1280, res=True)
attn(4096, 5); attn(1024, 10); attn(256, 20)

side = torch.cuda.Stream()
big_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
big_b = torch.empty_like(big_a)
small = torch.randn(4096, 64, device=dev, dtype=torch.float64)
stop = [False]


def background(n):
    with torch.cuda.stream(side):
        for _ in range(n):
            big_b.copy_(big_a)                                # 512 MB of HBM traffic
            for _ in range(4):
                (small @ small.t()).sum()                     # small f64 kernels: a few CUs at a time


def live(t):
    """the planes a split operand image carries: [hi | lo] of [.., 3 C] fp16 -- the third plane is unwritten for C % 64 == 0 (csrc/common.h)"""
    c3 = t.shape[-1]
    return t[..., :2 * (c3 // 3)] if args.exact and t.dtype == torch.float16 and c3 % 192 == 0 else t   # exact mode: fp16 results are images


bad_total = 0
twin = torch.cuda.Stream()
keep = []
if args.unet_bg:
    import bench
    torch.set_grad_enabled(False)
    eng, cfg, _sd, _n = bench.build(False, False, dev)
    unet = eng.model.diffusion_model
    unet.set_precision("exact")
    unet.tap_mode = "none"
    unet._set_taps()
    ux = rn(28, 4, 64, 64)
    ut = torch.full((28,), 958.0, device=dev)
    uctx = rn(28, 77, 1024)
    unet(ux, timesteps=ut, context=uctx)
    torch.cuda.synchronize()
for ci, (name, fn) in enumerate(cases):
    ref = live(fn()).clone()
    torch.cuda.synchronize()
    bad = 0
    for it in range(args.iters):
        if args.unet_bg:
            if it % 8 == 0:
                with torch.cuda.stream(twin):
                    keep.append(unet(ux, timesteps=ut, context=uctx))
                    del keep[:-2]
            out = live(fn())
        elif args.cross:
            with torch.cuda.stream(twin):
                for k in range(3):
                    keep.append(cases[(ci + 1 + (3 * it + k) % (len(cases) - 1)) % len(cases)][1]())
                del keep[:-6]
            out = live(fn())
        elif args.twin:
            twin.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(twin):
                out2 = live(fn())
            out = live(fn())
            torch.cuda.current_stream().wait_stream(twin)
            out2.record_stream(torch.cuda.current_stream())
            if not torch.equal(out2, ref):
                out = out2
        else:
            if not args.quiet and it % 4 == 0:
                background(2)
            out = live(fn())
        if not torch.equal(out, ref):
            bad += 1
            if bad <= 2:
                d = (out.float() - ref.float()).abs()
                print(f"   iter {it}: {int((d > 0).sum())} elements differ, max {float(d.max()):.4g}", flush=True)
                if os.environ.get("ROW_DETAIL") == "1" and out.dim() == 3:
                    idx = torch.nonzero(d > 0)
                    b_, n_ = int(idx[0, 0]), int(idx[0, 1])
                    c0, c1 = int(idx[:, 2].min()), int(idx[:, 2].max())
                    g, r = out[b_, n_, c0:c1 + 1].float().cpu(), ref[b_, n_, c0:c1 + 1].float().cpu()
                    ratio = g / r
                    print(f"      sample {b_} query {n_} columns {c0}..{c1}; rows/samples touched: {sorted(set(idx[:, 0].tolist()))} / {sorted(set(idx[:, 1].tolist()))}; "
                          f"got/ref ratio min {float(ratio.min()):.6f} max {float(ratio.max()):.6f}; got[:4] {g[:4].tolist()} ref[:4] {r[:4].tolist()}", flush=True)
    torch.cuda.synchronize()
    bad_total += bad
    print(f"{name:48s} {'OK' if bad == 0 else f'{bad} of {args.iters} runs DIFFER'}", flush=True)
print("unstable results:", bad_total)
