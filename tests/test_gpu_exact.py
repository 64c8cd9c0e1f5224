"""The "exact" precision mode (vidseg_diffusion_amd/exact.py, csrc/exact_ops.hip): fp32-accurate evaluation of the SD UNet on the
16-bit MFMA kernels over split (hi, lo) operands.

Bars (fp16 build only; floating point, stated against float64 evaluations of the same fp32 inputs):
    glue operators (split3, GroupNorm, LayerNorm, GEGLU, attention)   |err| <= 2e-6 * max|ref|   (fp32 arithmetic)
    split GEMM / conv (one MFMA GEMM over the 3-fold K axis)             |err| <= 5e-6 * max|ref|   (22-bit operands, fp32 accumulation)
    narrow UNet forward + taps vs the REFERENCE golden                   normalised rms <= 5e-5 (output), fp16 taps: at most 1 ulp apart
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from conftest import act_mode
from vidseg_diffusion_amd import synthetic

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden", "unet_sd_narrow.npz")


@pytest.fixture(scope="module")
def X():
    assert torch.cuda.is_available()
    from vidseg_diffusion_amd import _lib, exact
    _lib.lib()
    if act_mode()[0] != "f16":
        pytest.skip("the exact mode exists in the fp16 build only")
    return exact


def rel(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / np.abs(ref).max())


def rnd(shape, seed, scale=1.0):
    return (torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal(shape).astype(np.float32)) * scale)


def test_split3_reconstructs_22_bits(X):
    dev = torch.device("cuda:0")
    for scale in (1.0, 1e-2, 30.0):                              # incl. values whose lo part is an fp16 subnormal
        for C in (320, 100):          # C % 64 == 0 (every width a GEMM takes): two planes, the third left unwritten; other widths [hi | lo | hi]
            x = rnd((37, C), 1, scale)
            s = X.split3(x.to(dev)).cpu()
            assert s.dtype == torch.float16 and tuple(s.shape) == (37, 3 * C)
            hi, lo, hi2 = s[:, :C], s[:, C:2 * C], s[:, 2 * C:]
            assert torch.equal(hi, x.half()) and (torch.equal(hi, hi2) if C % 64 else bool(torch.isnan(hi2).all()))   # conftest poisons the unwritten plane
            rec = hi.double() + lo.double()
            assert float((rec - x.double()).abs().max()) <= max(2.0 ** -21 * scale * 6, 2.0 ** -24), scale
    a, b = rnd((9, 7, 128), 3), rnd((9, 7, 64), 4, 20.0)
    assert torch.equal(X.split3_cat(a.to(dev), b.to(dev)).cpu()[..., :2 * 192], X.split3(torch.cat([a, b], -1).to(dev)).cpu()[..., :2 * 192])
    y = X.split3(rnd((5, 64), 2).to(dev), silu=True).cpu()
    ref = TF.silu(rnd((5, 64), 2).double())
    assert rel(y[:, :64].double() + y[:, 64:128].double(), ref) <= 2e-6


def test_split_gemm_is_fp32_accurate(X):
    """linear_x / conv3x3_x = ONE call of the 16-bit MFMA GEMM over [hi|lo|hi] x [hi|hi|lo]: compare with float64."""
    from vidseg_diffusion_amd import ops
    dev = torch.device("cuda:0")
    for (M, K, N), sa, sw in (((300, 320, 640), 1.0, 0.02), ((4096, 1280, 320), 3.0, 0.02), ((28, 320, 1280), 1e-2, 1e-3)):
        a, w, b = rnd((M, K), 3, sa), rnd((N, K), 4, sw), rnd((N,), 5)
        out = X.linear_x(X.split3(a.to(dev)), X.pack_linear_x(w, dev), ops.f32(b, dev)).cpu()
        ref = a.double() @ w.double().t() + b.double()
        e = rel(out, ref)
        e16 = rel(a.half().double() @ w.half().double().t() + b.double(), ref)      # what plain fp16 operands cost on the same data
        print(f"split GEMM {M}x{N}x{K} (scales {sa}, {sw}): max err {e:.2e} of max|ref| (fp16 operands alone: {e16:.2e})")
        assert e <= 5e-6, (M, K, N, e)
    x, w, b = rnd((2, 64, 12, 20), 6), rnd((128, 64, 3, 3), 7, 0.05), rnd((128,), 8)
    xn = x.permute(0, 2, 3, 1).contiguous().to(dev)
    for kw, ref in ((dict(), TF.conv2d(x.double(), w.double(), b.double(), padding=1)),
                    (dict(stride=2), TF.conv2d(x.double(), w.double(), b.double(), padding=1, stride=2)),
                    (dict(up=2), TF.conv2d(TF.interpolate(x.double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1))):
        out = X.conv3x3_x(X.split3(xn), X.pack_conv3x3_x(w, dev), ops.f32(b, dev), **kw).cpu().permute(0, 3, 1, 2)
        e = rel(out, ref)
        print(f"split conv3x3 {kw}: max err {e:.2e}")
        assert e <= 5e-6, (kw, e)


def test_split_tile_stages_each_plane_once(X):
    """k_gemm_p7x (csrc/gemm_conv.hip): where the 224 x 320 tile is chosen, the split-operand entry points stage a_hi, a_lo, w_hi, w_lo
    once per 64 original channels and form a_lo w_hi + a_hi w_hi + a_hi w_lo from them, reading planes 0 / 1 of the [hi | lo | hi]
    activation image and planes 0 / 2 of the [hi | hi | lo] weight image in place.  Shapes chosen so that the tile IS selected (a full
    round of 256 CUs) while the float64 reference stays cheap: 3x3 conv (nine taps per 64-channel chunk), stride 2, nearest-2x
    upsample, per-sample vector + fp32 residual, taps, a split-K linear, the [3,1,1] temporal conv.  Same bar as the 3K walk."""
    from vidseg_diffusion_amd import ops
    dev = torch.device("cuda:0")
    ops.gemm_profile_begin()
    # 3x3 conv, stride 1, + bias + per-sample vector + fp32 residual: B = 10 at 64 x 64 -> M = 40960 (183 tiles of 224 rows)
    x, w, b = rnd((10, 64, 64, 64), 61), rnd((320, 64, 3, 3), 62, 0.05), rnd((320,), 63)
    rv, r = rnd((10, 320), 64), rnd((10, 64, 64, 320), 65, 2.0)
    out = X.conv3x3_x(X.split3(x.to(dev)), X.pack_conv3x3_x(w, dev), ops.f32(b, dev), rowvec=rv.to(dev), residual=r.to(dev)).cpu()
    ref = TF.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1) + rv.double()[:, None, None, :] + r.double()
    e = rel(out, ref)
    print(f"split tile conv3x3 10x64x64 64->320 + vector + residual: max err {e:.2e}")
    assert e <= 5e-6, e
    # stride 2 (B = 40: M = 40960) and nearest-2x upsample (B = 10 at 32 x 32 -> 64 x 64), two 64-channel chunks
    x, w, b = rnd((40, 64, 64, 128), 66), rnd((320, 128, 3, 3), 67, 0.05), rnd((320,), 68)
    out = X.conv3x3_x(X.split3(x.to(dev)), X.pack_conv3x3_x(w, dev), ops.f32(b, dev), stride=2).cpu()
    ref = TF.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1, stride=2).permute(0, 2, 3, 1)
    e = rel(out, ref)
    print(f"split tile conv3x3 stride 2: max err {e:.2e}")
    assert e <= 5e-6, e
    x = rnd((10, 32, 32, 128), 69)
    out = X.conv3x3_x(X.split3(x.to(dev)), X.pack_conv3x3_x(w, dev), ops.f32(b, dev), up=2).cpu()
    ref = TF.conv2d(TF.interpolate(x.double().permute(0, 3, 1, 2), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    e = rel(out, ref)
    print(f"split tile conv3x3 up 2: max err {e:.2e}")
    assert e <= 5e-6, e
    # linears: K = 320 with residual + fp16 taps of the first columns; split-K shape
    M, K, N = 40960, 320, 960
    a, w, b = rnd((M, K), 70), rnd((N, K), 71, 0.05), rnd((N,), 72)
    t1, t2 = torch.empty((M, 320), dtype=torch.float16, device=dev), torch.empty((M, 320), dtype=torch.float16, device=dev)
    out = X.linear_x(X.split3(a.to(dev)), X.pack_linear_x(w, dev), ops.f32(b, dev), tap=t1, tap2=t2, tap_cols=320).cpu()
    ref = a.double() @ w.double().t() + b.double()
    e = rel(out, ref)
    print(f"split tile linear {M}x{N}x{K} + taps: max err {e:.2e}")
    assert e <= 5e-6, e
    assert torch.equal(t1.cpu(), out[:, :320].half()) and torch.equal(t2.cpu(), out[:, 320:640].half())
    for (M, K, N) in ((7168, 5120, 1280), (28672, 640, 640)):
        a, w, b, r = rnd((M, K), 73), rnd((N, K), 74, 0.02), rnd((N,), 75), rnd((M, N), 76, 2.0)
        out = X.linear_x(X.split3(a.to(dev)), X.pack_linear_x(w, dev), ops.f32(b, dev), residual=r.to(dev)).cpu()
        e = rel(out, a.double() @ w.double().t() + b.double() + r.double())
        print(f"split tile linear {M}x{N}x{K} + fp32 residual: max err {e:.2e}")
        assert e <= 5e-6, (M, K, N, e)
    # [3,1,1] temporal conv over T = 14 frames of two videos, 64 x 64 locations, 128 -> 320 channels
    T, BT, H, W, C, Co = 14, 28, 64, 64, 128, 320
    x, w, b = rnd((BT, H, W, C), 77), rnd((Co, C, 3, 1, 1), 78, 0.05), rnd((Co,), 79)
    rv = rnd((BT, Co), 80)
    out = X.conv_temporal3_x(X.split3(x.to(dev)), X.pack_conv_temporal3_x(w, dev), ops.f32(b, dev), T, rowvec=rv.to(dev)).cpu()
    x5 = x.double().view(2, T, H, W, C).permute(0, 4, 1, 2, 3)                               # (b t) h w c -> b c t h w
    ref = TF.conv3d(x5, w.double(), b.double(), padding=(1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(BT, H, W, Co) + rv.double()[:, None, None, :]
    e = rel(out, ref)
    print(f"split tile temporal conv T=14 128->320: max err {e:.2e}")
    assert e <= 5e-6, e
    # k_gemm_phx: the 256 x 320 tile with the same four-tile staging, for shapes whose M fills a round of 256-row tiles (M = 28 * 18 * 32
    # = 63 tile rows x 4 tile columns = 252 of 256 CUs: the SVD window's 18x32 level): 3x3 conv + residual and a long-K linear
    x, w, b = rnd((28, 18, 32, 128), 81), rnd((1280, 128, 3, 3), 82, 0.05), rnd((1280,), 83)
    r = rnd((28, 18, 32, 1280), 84, 2.0)
    out = X.conv3x3_x(X.split3(x.to(dev)), X.pack_conv3x3_x(w, dev), ops.f32(b, dev), residual=r.to(dev)).cpu()
    ref = TF.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1) + r.double()
    e = rel(out, ref)
    print(f"split 256-row tile conv3x3 28x18x32 128->1280 + residual: max err {e:.2e}")
    assert e <= 5e-6, e
    M, K, N = 16128, 1280, 1280
    a, w, b, r = rnd((M, K), 85), rnd((N, K), 86, 0.02), rnd((N,), 87), rnd((M, N), 88, 2.0)
    out = X.linear_x(X.split3(a.to(dev)), X.pack_linear_x(w, dev), ops.f32(b, dev), residual=r.to(dev)).cpu()
    e = rel(out, a.double() @ w.double().t() + b.double() + r.double())
    print(f"split 256-row tile linear {M}x{N}x{K} + fp32 residual: max err {e:.2e}")
    assert e <= 5e-6, e
    ops.gemm_profile_end()
    kinds = {n.split(" (")[0]: ln for (n, ms, fl, ln, ab) in ops.gemm_profile_kinds()}
    print("launches by kernel:", kinds)
    assert kinds.get("k_gemm_p7x<5, false>", 0) >= 4 and kinds.get("k_gemm_ph<NJ>", 0) >= 3, kinds                                           # the convolutions above must exercise the new tile (the N = 960 / split-K linears go to other tiles)


_FOLD_SCRIPT = r"""
import sys, numpy as np, torch, torch.nn.functional as TF
sys.path.insert(0, sys.argv[1])
from vidseg_diffusion_amd import exact as X, ops
dev = torch.device("cuda:0")
def rnd(shape, seed, scale=1.0):
    return torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal(shape).astype(np.float32)) * scale
def rel(got, ref):
    return float((got.double() - ref).abs().max() / ref.abs().max())
ops.gemm_profile_begin()
worst = 0.0
for (M, K, N) in ((40960, 320, 960), (7168, 5120, 1280), (28672, 640, 640), (300, 320, 640), (1792, 1280, 1280)):
    a, w, b = rnd((M, K), 1), rnd((N, K), 2, 0.03), rnd((N,), 3)
    a3 = X.split3(a.to(dev))
    assert bool(torch.isnan(a3[:, 2 * K:]).all())                      # the poisoned, never-written third plane
    out = X.linear_x(a3, X.pack_linear_x(w, dev), ops.f32(b, dev)).cpu()
    worst = max(worst, rel(out, a.double() @ w.double().t() + b.double()))
x, w, b = rnd((10, 64, 64, 64), 4), rnd((320, 64, 3, 3), 5, 0.05), rnd((320,), 6)
out = X.conv3x3_x(X.split3(x.to(dev)), X.pack_conv3x3_x(w, dev), ops.f32(b, dev)).cpu()
worst = max(worst, rel(out, TF.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)))
a, w, b = rnd((7168, 320), 7, 1.5), rnd((2560, 320), 8, 0.03), rnd((2560,), 9, 0.5)
w3g, bg, grp = X.pack_geglu_x(w, b, dev)
g = X.geglu_linear_x(X.split3(a.to(dev)), w3g, bg, grp).cpu()
y = a.double() @ w.double().t() + b.double()
worst = max(worst, rel(g[:, :1280].double() + g[:, 1280:2560].double(), y[:, :1280] * TF.gelu(y[:, 1280:])))
ops.gemm_profile_end()
print("KINDS", sorted(n.split(" (")[0] for (n, ms, fl, ln, ab) in ops.gemm_profile_kinds() if ln), "WORST", worst)
assert worst <= 5e-6, worst
"""


def test_no_kernel_reads_the_third_plane(X):
    """Operand images of GEMM widths carry TWO planes (csrc/common.h: VS_THIRD_PLANE): k_gemm_p7x / k_gemm_phx address planes 0, 1 and
    every other GEMM kernel folds the last third of its 3 C walk back onto plane 0 (GemmParams::a_fold).  With the third plane
    poisoned (conftest: VIDSEG_X_POISON_PLANE3) the same shapes must stay fp32-accurate on every kernel family the VIDSEG_GEMM
    override can route split operands to -- one process each, the override is read once."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = set()
    for knob, tile in (("", "p7x"), ("p7x=0,phx=0", "p7x"), ("big=0", "ph"), ("big=0,mid=2", "ph"), ("big=0,mid=0,dma=0", "ph"), ("big=2,p7=0", "ph"), ("ph=0", "ph")):
        env = dict(os.environ, VIDSEG_GEMM=knob, VIDSEG_X_POISON_PLANE3="1", VIDSEG_X_GEGLU_TILE=tile)
        r = subprocess.run([sys.executable, "-c", _FOLD_SCRIPT, root], env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("KINDS")]
        print(f"VIDSEG_GEMM='{knob}' GEGLU tile {tile}: {line[0] if line else r.stderr[-400:]}")
        assert r.returncode == 0 and line, (knob, r.stderr[-800:])
        seen.update(eval(line[0].split("KINDS ")[1].split(" WORST")[0]))
    print("kernel families exercised on poisoned images:", sorted(seen))
    assert len(seen) >= 6, seen


def test_layernorm_rows_per_wave_is_a_schedule_not_an_arithmetic(X):
    """LayerNorm -> split image with several rows per wave (csrc/exact_ops.hip: k_x_layernorm_split3_rows, taken for M >= 4096) against
    the one-row-per-wave kernel the masks were pinned with: the same bits (a second process with VIDSEG_X_LN_ROWS=1 runs the old schedule)."""
    import subprocess
    import sys
    import tempfile
    dev = torch.device("cuda:0")
    code = (
        "import sys, torch, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from vidseg_diffusion_amd import exact as X\n"
        "out = {}\n"
        "for M, C in ((4096, 320), (4099, 320), (114688, 320), (28672, 640), (5000, 1024), (7168, 1280), (4100, 64), (4097, 512)):\n"
        "    g = np.random.Generator(np.random.PCG64(M + C))\n"
        "    x = torch.from_numpy((g.standard_normal((M, C)) * 2.0 + 0.3).astype(np.float32)).cuda()\n"
        "    ga = torch.from_numpy((1 + 0.1 * g.standard_normal(C)).astype(np.float32)).cuda()\n"
        "    be = torch.from_numpy((0.05 * g.standard_normal(C)).astype(np.float32)).cuda()\n"
        "    y = X.layernorm_split3(x, ga, be)\n"
        "    out[f'{M}x{C}'] = y[:, :2 * C].contiguous().cpu().numpy().view(np.uint16)\n"
        "    vec = torch.from_numpy(g.standard_normal((14, C)).astype(np.float32)).cuda()\n"
        "    xs, y2 = X.layernorm_rowvec_split3(x, vec, 37, ga, be)\n"
        "    out[f'{M}x{C}rv'] = y2[:, :2 * C].contiguous().cpu().numpy().view(np.uint16)\n"
        "    out[f'{M}x{C}sum'] = xs.cpu().numpy().view(np.uint32)\n"
        "np.savez(sys.argv[1], **out)\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        res = {}
        for knob in ("0", "1"):
            path = os.path.join(d, f"ln{knob}.npz")
            env = dict(os.environ, VIDSEG_X_LN_ROWS=knob, VIDSEG_X_POISON_PLANE3="0")
            subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=300)
            res[knob] = np.load(path)
        assert sorted(res["0"].files) == sorted(res["1"].files) and len(res["0"].files) == 24
        for k in res["0"].files:
            assert np.array_equal(res["0"][k], res["1"][k]), k
    _ = X, dev


def test_fp32_glue_operators(X):
    from vidseg_diffusion_amd import ops
    dev = torch.device("cuda:0")

    def join(s, C):
        s = s.cpu().double()
        return s[..., :C] + s[..., C:2 * C]

    x0, x1 = rnd((3, 8, 10, 64), 11) + 0.7, rnd((3, 8, 10, 128), 12, 2.0)
    g, b = rnd((192,), 13) * 0.3 + 1.0, rnd((192,), 14) * 0.2
    for silu in (True, False):
        for two in (True, False):
            xs = torch.cat([x0, x1], -1) if two else x0
            C = xs.shape[-1]
            out = join(X.groupnorm_split3(x0.to(dev), ops.f32(g[:C], dev), ops.f32(b[:C], dev), x1=x1.to(dev) if two else None, eps=1e-5,
                                          silu=silu), C)
            ref = TF.group_norm(xs.double().permute(0, 3, 1, 2), 32, g[:C].double(), b[:C].double(), 1e-5).permute(0, 2, 3, 1)
            ref = TF.silu(ref) if silu else ref
            assert rel(out, ref) <= 2e-6, (silu, two, rel(out, ref))
    for C in (320, 640, 1280):
        x, gg, bb = rnd((77, C), 15) * 2 + 0.3, rnd((C,), 16) * 0.3 + 1.0, rnd((C,), 17) * 0.1
        out = join(X.layernorm_split3(x.to(dev), ops.f32(gg, dev), ops.f32(bb, dev)), C)
        assert rel(out, TF.layer_norm(x.double(), (C,), gg.double(), bb.double(), 1e-5)) <= 2e-6, C
    # the frame-embedding add folded into norm_in's pass: the bits of LayerNorm(add_rowvec(x, emb))
    x, emb, gg, bb = rnd((6, 35, 320), 25), rnd((3, 320), 26), rnd((320,), 27) * 0.3 + 1.0, rnd((320,), 28) * 0.1
    xs, img = X.layernorm_rowvec_split3(x.to(dev), emb.to(dev), 35, ops.f32(gg, dev), ops.f32(bb, dev))
    ref_sum = X.add_rowvec(x.to(dev), emb.to(dev), 35)
    assert torch.equal(xs, ref_sum) and torch.equal(img[..., :640], X.layernorm_split3(ref_sum, ops.f32(gg, dev), ops.f32(bb, dev))[..., :640])
    y = rnd((50, 2 * 256), 18, 1.5)
    out = join(X.geglu_split3(y.to(dev)), 256)
    v, gt = y.double().chunk(2, dim=-1)
    assert rel(out, v * TF.gelu(gt)) <= 2e-6
    for (B, H, Nq, Nk) in ((2, 5, 200, 200), (3, 2, 64, 77), (1, 10, 1024, 1024)):
        q, k, vv = rnd((B, Nq, H * 64), 19), rnd((B, Nk, H * 64), 20), rnd((B, Nk, H * 64), 21)
        qkv = torch.cat([q, q, q], -1).to(dev)                                   # q as a column slice of a wider buffer (stride 3C)
        out = X.attention_f32(qkv[..., H * 64:2 * H * 64], k.to(dev), vv.to(dev), H, B, Nq, Nk).cpu()
        hd = lambda t, n: t.double().view(B, n, H, 64).transpose(1, 2)            # noqa: E731
        ref = TF.scaled_dot_product_attention(hd(q, Nq), hd(k, Nk), hd(vv, Nk)).transpose(1, 2).reshape(B, Nq, H * 64)
        e = rel(out, ref)
        print(f"attention_f32 B={B} H={H} Nq={Nq} Nk={Nk}: max err {e:.2e}")
        assert e <= 2e-6, (B, H, Nq, Nk, e)


def test_attention_mfma_split_operands(X):
    """k_x_attention_mfma (both contractions as three fp16 MFMA products of hi / lo operands, P split in registers) against float64
    attention of the same fp32 inputs: self-attention sizes, the ragged 77-key context, a query count that is no multiple of the
    128-query block, large score ranges (the running maximum moves and the accumulators are rescaled) and inputs of small magnitude
    (lo parts in the fp16 subnormal range)."""
    dev = torch.device("cuda:0")
    cases = ((2, 5, 256, 256, 1.0), (3, 2, 200, 77, 1.0), (1, 10, 1024, 1024, 1.0), (1, 5, 4096, 4096, 1.0), (2, 3, 384, 320, 4.0),
             (2, 3, 128, 192, 0.02))
    for (B, H, Nq, Nk, s) in cases:
        C = H * 64
        q, k, vv = rnd((B, Nq, C), 19, s), rnd((B, Nk, C), 20, s), rnd((B, Nk, C), 21, s)
        qq = torch.cat([q, q], -1).to(dev)                                       # q as a column slice of a wider buffer
        kv = torch.cat([torch.full((B, Nk, C), 7.0), k, vv], -1).to(dev)          # k | v as columns [C, 3C) of a [.., 3C] buffer
        out = X.attention_mfma(qq[..., C:], kv[..., C:], H, B, Nq, Nk).cpu()
        hd = lambda t, n: t.double().view(B, n, H, 64).transpose(1, 2)            # noqa: E731
        ref = TF.scaled_dot_product_attention(hd(q, Nq), hd(k, Nk), hd(vv, Nk)).transpose(1, 2).reshape(B, Nq, C)
        e = rel(out, ref)
        f32 = X.attention_f32(qq[..., C:], kv[..., C:2 * C], kv[..., 2 * C:], H, B, Nq, Nk).cpu()
        print(f"attention_mfma B={B} H={H} Nq={Nq} Nk={Nk} scale={s}: max err {e:.2e} (fp32 vector kernel {rel(f32, ref):.2e})")
        # scores grow with s^2 and with them the fp32 spacing of the exponent's argument: the bar follows (the fp32 vector kernel
        # reads 5.3e-6 at s = 4 where this one reads 2.8e-6)
        assert e <= 2e-6 * max(1.0, s * s / 4), (B, H, Nq, Nk, s, e)


def test_fused_residual_and_split_outputs(X):
    """The exact mode's fused forms: fp32 residual inside the GEMM / conv epilogue (incl. a split-K shape, whose finish kernel adds
    it), the attention kernel writing the consumer's [hi | lo | hi] image, GroupNorm statistics over the coalesced two-pass kernels
    at the window's channel counts (80 / 640 four-channel columns, two sources)."""
    from vidseg_diffusion_amd import ops
    dev = torch.device("cuda:0")
    for (M, K, N) in ((300, 320, 640), (7168, 5120, 1280), (1792, 1280, 1280)):
        a, w, b, r = rnd((M, K), 41), rnd((N, K), 42, 0.02), rnd((N,), 43), rnd((M, N), 44, 2.0)
        out = X.linear_x(X.split3(a.to(dev)), X.pack_linear_x(w, dev), ops.f32(b, dev), residual=r.to(dev)).cpu()
        ref = a.double() @ w.double().t() + b.double() + r.double()
        e = rel(out, ref)
        print(f"linear_x + fp32 residual {M}x{N}x{K}: max err {e:.2e}")
        assert e <= 5e-6, (M, K, N, e)
    # the same linear writing the next GEMM's operand image from its epilogue (vidseg_linear_a16_rf32_x3): the bits of split3(linear_x(..))
    # on the 224 x 320 split tile, on a split-K shape (k_splitk_finish writes the image), on the small tile, with and without the residual
    for (M, K, N, res) in ((28672, 1280, 320, True), (7168, 5120, 1280, True), (1792, 5120, 1280, True), (300, 320, 640, False), (114688, 1280, 320, True)):
        a, w, b = rnd((M, K), 71), rnd((N, K), 72, 0.02), rnd((N,), 73)
        r = rnd((M, N), 74, 2.0).to(dev) if res else None
        a3, w3, bd = X.split3(a.to(dev)), X.pack_linear_x(w, dev), ops.f32(b, dev)
        two = X.split3(X.linear_x(a3, w3, bd, residual=r))
        one = X.linear_x(a3, w3, bd, residual=r, split_out=True)
        assert tuple(one.shape) == (M, 3 * N) and torch.equal(one[:, :2 * N], two[:, :2 * N]), (M, K, N, res)
    # the fused q | k | v projection writing k | v as the attention kernel's planes (vidseg_linear_a16_qkv_planes): the bits of
    # split_planes(linear_x(..)[..., Ci:]), q and both taps unchanged; the split tile, a split-K shape and the small tile
    for (B, N, Ci) in ((7, 4096, 320), (4, 256, 1280), (28, 64, 1280), (1, 300, 64)):
        a, w = rnd((B, N, Ci), 75), rnd((3 * Ci, Ci), 76, 0.03)
        a3, w3 = X.split3(a.to(dev)), X.pack_linear_x(w, dev)
        t1, t2 = torch.empty((B, N, Ci), dtype=torch.float16, device=dev), torch.empty((B, N, Ci), dtype=torch.float16, device=dev)
        u1, u2 = torch.empty_like(t1), torch.empty_like(t2)
        qkv = X.linear_x(a3, w3, tap=t1, tap2=t2, tap_cols=Ci)
        hi0, lo0 = X.split_planes(qkv[..., Ci:])
        q, (hi, lo) = X.linear_qkv_x(a3, w3, Ci, tap=u1, tap2=u2)
        assert torch.equal(q, qkv[..., :Ci]) and torch.equal(hi, hi0) and torch.equal(lo, lo0) and torch.equal(t1, u1) and torch.equal(t2, u2), (B, N, Ci)
        if N >= 128:
            heads = Ci // 64
            assert torch.equal(X.attention_x(q, None, heads, B, N, N, split_out=True, planes=(hi, lo))[..., :2 * Ci],
                               X.attention_x(qkv[..., :Ci], qkv[..., Ci:], heads, B, N, N, split_out=True)[..., :2 * Ci])
    x, w, b = rnd((2, 64, 12, 20), 45), rnd((128, 64, 3, 3), 46, 0.05), rnd((128,), 47)
    r = rnd((2, 12, 20, 128), 48, 3.0)
    out = X.conv3x3_x(X.split3(x.permute(0, 2, 3, 1).contiguous().to(dev)), X.pack_conv3x3_x(w, dev), ops.f32(b, dev), residual=r.to(dev)).cpu()
    ref = TF.conv2d(x.double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1) + r.double()
    assert rel(out, ref) <= 5e-6, rel(out, ref)
    B, H, Nq, Nk = 2, 5, 256, 192
    C = H * 64
    q, kv = rnd((B, Nq, C), 49), rnd((B, Nk, 2 * C), 50)
    o32 = X.attention_mfma(q.to(dev), kv.to(dev), H, B, Nq, Nk).cpu()
    o3 = X.attention_mfma(q.to(dev), kv.to(dev), H, B, Nq, Nk, split_out=True).cpu()
    assert tuple(o3.shape) == (B, Nq, 3 * C) and torch.equal(o3[..., :C], o32.half())          # (third plane: never written for C % 64 == 0)
    assert float((o3[..., :C].double() + o3[..., C:2 * C].double() - o32.double()).abs().max()) <= 2.0 ** -21 * float(o32.abs().max())

    def join(s, C):
        s = s.cpu().double()
        return s[..., :C] + s[..., C:2 * C]

    for (B, Hh, Ww, C0, C1) in ((2, 64, 64, 320, 0), (2, 16, 16, 1280, 1280), (3, 8, 8, 192, 0), (1, 32, 32, 640, 320)):
        C = C0 + C1
        x0 = rnd((B, Hh, Ww, C0), 51) * 1.5 + 0.4
        x1 = rnd((B, Hh, Ww, C1), 52, 0.7) if C1 else None
        g, bt = rnd((C,), 53) * 0.3 + 1.0, rnd((C,), 54) * 0.2
        out = join(X.groupnorm_split3(x0.to(dev), ops.f32(g, dev), ops.f32(bt, dev), x1=x1.to(dev) if C1 else None, eps=1e-5, silu=True), C)
        xs = torch.cat([x0, x1], -1) if C1 else x0
        ref = TF.silu(TF.group_norm(xs.double().permute(0, 3, 1, 2), 32, g.double(), bt.double(), 1e-5).permute(0, 2, 3, 1))
        e = rel(out, ref)
        print(f"groupnorm_split3 B={B} {Hh}x{Ww} C={C0}+{C1}: max err {e:.2e}")
        assert e <= 2e-6, (B, Hh, Ww, C0, C1, e)


def test_geglu_projection_fused_into_the_gemm(X):
    """vidseg_linear_a16_geglu_x3: value * gelu_erf(gate) formed in fp32 inside the GEMM epilogue and written as the [hi | lo | hi]
    image, against float64 and against the two-launch form (projection to fp32 + k_x_geglu_split3) on the same operands."""
    from vidseg_diffusion_amd import ops
    dev = torch.device("cuda:0")
    tile0 = X._GEGLU_TILE
    try:
        for tile in ("p7x", "ph"):                                  # the 224 x 256 split tile (16-row groups) and the 256 x 256 phased tile (32-row groups)
            X._GEGLU_TILE = tile
            # (.., 64, 128) / (.., 128, 128): one and two macro-tiles on the split tile (prologue and tail of its staging schedule)
            for (M, K, inner) in ((300, 320, 1280), (7168, 1280, 5120), (1000, 640, 2560), (230, 64, 96), (300, 64, 128), (5000, 128, 128)):
                a, w, b = rnd((M, K), 61, 1.5), rnd((2 * inner, K), 62, 0.03), rnd((2 * inner,), 63, 0.5)
                a3 = X.split3(a.to(dev))
                w3g, bg, grp = X.pack_geglu_x(w, b, dev)
                assert grp == (16 if tile == "p7x" and (2 * inner) % 256 == 0 else 32)
                fused = X.geglu_linear_x(a3, w3g, bg, grp).cpu()
                two = X.geglu_split3(X.linear_x(a3, X.pack_linear_x(w, dev), ops.f32(b, dev))).cpu()
                assert tuple(fused.shape) == (M, 3 * inner) and (inner % 64 == 0 or torch.equal(fused[:, :inner], fused[:, 2 * inner:]))
                y = a.double() @ w.double().t() + b.double()
                ref = y[:, :inner] * TF.gelu(y[:, inner:])
                got = fused[:, :inner].double() + fused[:, inner:2 * inner].double()
                e, e2 = rel(got, ref), rel(two[:, :inner].double() + two[:, inner:2 * inner].double(), ref)
                same = float((fused[:, :2 * inner] == two[:, :2 * inner]).double().mean())
                print(f"fused GEGLU [{tile}, groups of {grp}] {M}x{2 * inner}x{K}: max err {e:.2e} (two launches {e2:.2e}); fp16 words identical to the "
                      f"two-launch form {same:.4f}")
                assert e <= 5e-6, (tile, M, K, inner, e)
    finally:
        X._GEGLU_TILE = tile0


def test_split_survives_rounding_ties(X):
    """hi + lo must reconstruct x when x sits EXACTLY on an fp16 rounding tie and is the product of two fp32 values (the case in
    which hipcc fused the conversion with the multiply differently per use: hi of the operand image from the rounded fp32 product,
    lo against the singly-rounded exact product -- one fp16 ulp apart on ties, csrc/exact_ops.hip:split_hl).  The attention
    prologue multiplies q by dim_head^-0.5 log2 e before the split: feed q values whose scaled fp32 product is a tie."""
    dev = torch.device("cuda:0")
    sc = np.float32(0.125 * 1.44269504088896340736)
    rs = np.random.Generator(np.random.PCG64(5))
    # candidates: fp32 q with fp32(q * sc) = (2 m + 1) 2^-13 for odd numerators in [0.25, 0.5), found by search around the quotient
    ties = []
    while len(ties) < 64 * 128:
        m = int(rs.integers(1024, 2048))
        target = np.float32((2 * m + 1) * 2.0 ** -13)                     # midpoint between two fp16 values in [0.25, 0.5)
        q0 = np.float32(target / sc)
        for cand in (q0, np.nextafter(q0, np.float32(1)), np.nextafter(q0, np.float32(-1))):
            if np.float32(cand * sc) == target:
                ties.append(float(cand) * (1 if rs.integers(2) else -1))
                break
    B, H, Nq, Nk = 1, 1, 128, 256
    q = torch.tensor(ties[:Nq * 64], dtype=torch.float32).view(B, Nq, 64)
    k, vv = rnd((B, Nk, 64), 31), rnd((B, Nk, 64), 32)
    out = X.attention_mfma(q.to(dev), torch.cat([k, vv], -1).to(dev), H, B, Nq, Nk).cpu()
    ref = TF.scaled_dot_product_attention(q.double()[:, None], k.double()[:, None], vv.double()[:, None])[:, 0]
    e = rel(out, ref)
    print(f"attention_mfma on {Nq * 64} tie-valued q entries: max err {e:.2e}")
    assert e <= 2e-6, e


def test_temporal_attention_in_place(X):
    """k_x_temporal_attention: the time stack's self-attention read and written in the spatial row order (b t) s against float64
    attention through the reference's own rearranges (video_attention.py:171-199), for T = 14 (the drivers' window), short and
    full-length (16) sequences, location counts that are no multiple of anything, one and several videos; fp32 and split-image
    outputs agree bit for bit, the taps are fp16(q), fp16(k) in the reference's [(b s), t, c] layout."""
    dev = torch.device("cuda:0")
    for (nv, T, S, H, sc) in ((2, 14, 37, 5, 1.0), (1, 14, 144, 10, 1.0), (3, 3, 50, 1, 1.0), (1, 16, 9, 2, 3.0), (2, 1, 5, 1, 1.0), (1, 14, 2304, 5, 0.05)):
        C = H * 64
        qkv = rnd((nv * T, S, 3 * C), 30 + T, sc)
        tq = torch.empty((nv * S, T, C), dtype=torch.float16, device=dev)
        tk = torch.empty_like(tq)
        out = X.temporal_attention_x(qkv.to(dev), nv, T, S, H, split_out=False).cpu()
        img = X.temporal_attention_x(qkv.to(dev), nv, T, S, H, split_out=True, tap_q=tq, tap_k=tk).cpu()
        perm = lambda t: t.view(nv, T, S, -1).permute(0, 2, 1, 3).reshape(nv * S, T, -1)             # noqa: E731  (b t) s c -> (b s) t c
        q, k, v = (perm(qkv[..., i * C:(i + 1) * C].contiguous()) for i in range(3))
        hd = lambda t: t.double().view(nv * S, T, H, 64).transpose(1, 2)                            # noqa: E731
        ref = TF.scaled_dot_product_attention(hd(q), hd(k), hd(v)).transpose(1, 2).reshape(nv, S, T, C).permute(0, 2, 1, 3).reshape(nv * T, S, C)
        e = rel(out, ref)
        print(f"temporal attention videos={nv} T={T} S={S} H={H} scale={sc}: max err {e:.2e}")
        assert e <= 2e-6 * max(1.0, sc * sc), (nv, T, S, H, e)
        assert torch.equal(img[..., :C], out.half()) and torch.equal(img[..., :2 * C], X.split3(out.to(dev)).cpu()[..., :2 * C])
        assert torch.equal(tq.cpu(), q.half()) and torch.equal(tk.cpu(), k.half())


def test_alpha_blend_epilogues(X):
    """AlphaBlender inside the GEMM epilogue (vidseg_linear_a16_rf32_blend, vidseg_conv_temporal3_a16_f32_blend; incl. a split-K
    shape) against float64 of alpha * spatial + (1 - alpha) * (result + residual)."""
    from vidseg_diffusion_amd import ops
    dev = torch.device("cuda:0")
    alpha = 0.3775
    for (M, K, N) in ((448, 1280, 320), (7168, 5120, 1280), (300, 320, 640)):
        a, w, b, r, sp = rnd((M, K), 41), rnd((N, K), 42, 0.03), rnd((N,), 43), rnd((M, N), 44), rnd((M, N), 45, 2.0)
        out = X.linear_blend_x(X.split3(a.to(dev)), X.pack_linear_x(w, dev), ops.f32(b, dev), r.to(dev), sp.to(dev), alpha).cpu()
        ref = alpha * sp.double() + (1 - alpha) * (a.double() @ w.double().t() + b.double() + r.double())
        assert rel(out, ref) <= 5e-6, (M, K, N, rel(out, ref))
    T, HW, C, Co = 7, 20, 64, 128
    x, w, b = rnd((2 * T, 4, 5, C), 46), rnd((Co, C, 3, 1, 1), 47, 0.05), rnd((Co,), 48)
    r, sp = rnd((2 * T, 4, 5, Co), 49), rnd((2 * T, 4, 5, Co), 50)
    out = X.conv_temporal3_blend_x(X.split3(x.to(dev)), X.pack_conv_temporal3_x(w, dev), ops.f32(b, dev), T, r.to(dev), sp.to(dev), alpha).cpu()
    x5 = x.double().view(2, T, 4, 5, C).permute(0, 4, 1, 2, 3)                                       # b c t h w
    conv = TF.conv3d(x5, w.double(), b.double(), padding=(1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(2 * T, 4, 5, Co)
    ref = alpha * sp.double() + (1 - alpha) * (conv + r.double())
    assert rel(out, ref) <= 5e-6, rel(out, ref)


def test_video_unet_fused_paths_equal_the_plain_ones(X):
    """The round-5 fusions of the exact VideoUNet change no value beyond fp32 rounding order: (i) attention over a one-token context
    taken as the identity on v (softmax of one score is exactly 1; norm2 / to_q / the attention are not evaluated, to_out(v) rides on
    the preceding projection as a per-sample row vector), (ii) the temporal attention in place, (iii) AlphaBlender in the epilogues --
    each against the full evaluation of the same network (VIDSEG_X_NK1 / _TEMPORAL / _BLEND = 0), and the dumped taps unchanged."""
    from vidseg_diffusion_amd.video_unet import VideoUNet
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(os.path.dirname(G), "unet_svd_narrow.npz"))
    net = VideoUNet(**synthetic.SVD_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=4321).items()})
    T = int(z["T"])
    x, t, ctx, y = (torch.from_numpy(z[k]).to(dev) for k in ("fw_x", "fw_t", "fw_ctx", "fw_y"))
    kw = dict(timesteps=t, context=ctx, y=y, num_video_frames=T, image_only_indicator=torch.zeros(2, T))
    net.set_precision("exact")

    def run():
        out = net(x, **kw).cpu().double()
        blk = net.output_blocks[7][1]
        return out, [blk.transformer_blocks[0].attn1.q.cpu(), blk.time_stack[0].attn1.q.cpu(), blk.time_stack[0].attn1.k.cpu(),
                     blk.time_stack[0].attn2.k.cpu()]
    saved = (X._NK1_IDENTITY, X._TEMPORAL_FUSED, X._BLEND_FUSED)
    try:
        fused, ftaps = run()
        for flag in ("_NK1_IDENTITY", "_TEMPORAL_FUSED", "_BLEND_FUSED"):
            setattr(X, flag, False)
            plain, ptaps = run()
            setattr(X, flag, True)
            e = float((fused - plain).norm() / plain.norm())
            print(f"exact VideoUNet with {flag} off vs on: nrms {e:.2e}")
            assert e <= 2e-6, (flag, e)
            for a, b in zip(ftaps, ptaps):
                assert a.shape == b.shape and float((a != b).float().mean()) <= 0.01, flag
    finally:
        X._NK1_IDENTITY, X._TEMPORAL_FUSED, X._BLEND_FUSED = saved


def _step4_passes(X, kind, precision):
    """Feature pass + the reference-style modulated / injected / latent-blended passes of tests/golden/{sd,svd}_modulated_narrow.npz
    in the given precision; returns per pass (normalised rms of x after every step, of the final latent's DIFFERENCE from the
    unmodulated one) against the reference's own run."""
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import build_sd_engine, build_svd_engine, save_feature_maps
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(os.path.dirname(G), f"{kind}_modulated_narrow.npz"))
    g = {k: z[k] for k in z.files}
    nrms = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(np.asarray(b, np.float64)))   # noqa: E731
    Fn = g["latent"].shape[0]
    if kind == "svd":
        from vidseg_diffusion_amd.video_unet import VideoUNet
        net = VideoUNet(**synthetic.SVD_NARROW)
        seed, T0 = 4321, 22
    else:
        from vidseg_diffusion_amd.unet import UNetModel
        net = UNetModel(**synthetic.SD21_NARROW)
        seed, T0 = 1234, 22
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=seed).items()})
    net.pack(dev)
    net.set_precision(precision)
    if kind == "svd":
        eng = build_svd_engine(net, num_frames=Fn)
        c = {k[2:]: torch.from_numpy(v).to(dev) for k, v in g.items() if k.startswith("c_")}
        uc = {k[3:]: torch.from_numpy(v).to(dev) for k, v in g.items() if k.startswith("uc_")}
        extra = {"image_only_indicator": torch.zeros(2, Fn), "num_video_frames": Fn}
        tags = [(k[4:-6], float(g[f"lam_{k[4:-6]}"])) for k in g if k.startswith("mod_") and k.endswith("_final")]
        mkw = dict(modulate_block_idx=[8], modulate_layer_type=["spatial", "temporal"], modulate_attn_type=["self_attn"],
                   injected_feature_types=["temporal_cross_attn_k", "temporal_cross_attn_q", "temporal_self_attn_k", "temporal_self_attn_q"],
                   input_block_indices=[3, 4, 5, 6, 7, 8, 10, 11], latent_mask_end=25)
    else:
        eng = build_sd_engine(net)
        c = {"crossattn": torch.from_numpy(g["c"]).to(dev)}
        uc = {"crossattn": torch.zeros_like(c["crossattn"])}
        extra = {}
        tags = [("pos", 50.0), ("neg", -50.0)]
        mkw = dict(modulate_block_idx=[7], modulate_layer_type=["spatial"], modulate_attn_type=["cross_attn"],
                   injected_feature_types=["spatial_cross_attn_k", "spatial_cross_attn_q", "spatial_self_attn_k", "spatial_self_attn_q"],
                   input_block_indices=[3, 4, 5, 6, 7, 8, 9, 10, 11], latent_mask_end=23)
    noised = eng.sampler.add_noise(torch.from_numpy(g["latent"]).to(dev), cond=c, uc=uc, num_steps=25, noise_level=T0,
                                   noise=torch.from_numpy(g["noise"]).to(dev))
    FE.FeatureStore.clear()
    base, exp = f"/nonexistent/x_step4_{kind}", "exp"

    def denoiser(inp, sigma, cc, is_modulate_step=False, is_injected_step=False, modulate_params=None):
        return eng.denoiser(eng.model, inp, sigma, cc, is_modulate_step=is_modulate_step, is_injected_step=is_injected_step,
                            modulate_params=modulate_params, **extra)

    feat = eng.sampler(denoiser, noised.clone(), cond=c, uc=uc, t_start=T0, img_callback=lambda xt, i: save_feature_maps(eng, base, exp, i, xt=xt))
    out = {"feat": nrms(feat.cpu().numpy(), g["feat_final"])}
    for tag, lam in tags:
        mp = {"feature_masks": [torch.from_numpy(m).to(dev) for m in g["masks"]], "modulate_timestep": [T0], "modulate_schedule": "constant",
              "modulate_lambda_start": lam, "modulate_lambda_end": lam, "num_frames": Fn, "modulate_uc": True, "is_injected_features": True,
              "injected_block_types": ["output"], "output_block_indices": list(range(1, 12)), "feature_folder": base, "exp_name": exp,
              "injected_features_group": {}, "modulate_layer_frames": {}, "modulate_block_frames": {}, "modulate_timestep_frames": {},
              "modulate_lambda_layers": {}, "latent_mask_start": T0, **mkw}
        xs = []
        final = eng.sampler(denoiser, noised.clone(), cond=c, uc=uc, img_callback=lambda xt, i: xs.append(xt.cpu().numpy()), is_modulate=True,
                            modulate_params=mp, t_start=T0, is_latent_blending=True, feature_height=8, feature_width=8, model=None)
        ref = g[f"mod_{tag}_x_steps"]
        assert len(xs) == ref.shape[0]
        d_ref = g[f"mod_{tag}_final"] - g["feat_final"]
        out[tag] = (max(nrms(xs[i], ref[i]) for i in range(len(xs))), nrms(final.cpu().numpy() - feat.cpu().numpy(), d_ref))
    FE.FeatureStore.clear()
    net.release_exact()
    return out


@pytest.mark.parametrize("kind", ["sd", "svd"])
def test_exact_mode_step4_vs_reference(X, kind):
    """Step 4 (a17) in the exact mode: the modulated + injected + latent-blended sampler passes of the narrow SD UNet (lambda * mask on
    block 7's cross-attention output, injected spatial q / k) and of the narrow VideoUNet (block 8's spatial AND temporal self-attention
    outputs, injected temporal q / k) against the REFERENCE's own run (tests/golden/{sd,svd}_modulated_narrow.npz).  The injected q / k are
    the fp16 dumps of the feature pass here and fp32 tensors in the reference's CPU run, which bounds the agreement at the fp16
    rounding of q / k (the CPU oracle with fp16 dumps reads the same); bars: x after every step <= 1e-3 normalised rms and the modulation's
    EFFECT (modulated minus plain final latent) within 1 % of the reference's, both well inside the 16-bit mode's figures, which are
    measured beside them."""
    ex = _step4_passes(X, kind, "exact")
    lo = _step4_passes(X, kind, "fp16")
    for tag in [k for k in ex if k != "feat"]:
        print(f"{kind} Step 4 pass {tag}: exact mode x-step nrms {ex[tag][0]:.2e}, effect nrms {ex[tag][1]:.2e}   (16-bit mode {lo[tag][0]:.2e}, {lo[tag][1]:.2e})")
        assert ex[tag][0] <= 1e-3 and ex[tag][1] <= 1e-2, (kind, tag, ex[tag])
        assert ex[tag][0] <= lo[tag][0] and ex[tag][1] <= lo[tag][1] * 1.05 + 1e-4, (kind, tag, ex[tag], lo[tag])
    print(f"{kind} feature pass final latent: exact {ex['feat']:.2e}, 16-bit {lo['feat']:.2e}")
    assert ex["feat"] <= 5e-5


def test_exact_unet_forward_vs_reference(X):
    """Narrow SD UNet in the exact mode against the REFERENCE's fp32 forward (tests/golden/unet_sd_narrow.npz): output and every
    dumped Q/K tap -- and the 16-bit mode of the same object on the same inputs, for the ratio."""
    from vidseg_diffusion_amd.unet import UNetModel
    dev = torch.device("cuda:0")
    z = np.load(G)
    g = {k: z[k] for k in z.files}
    net = UNetModel(**synthetic.SD21_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, 1234).items()})
    x, t, ctx = (torch.from_numpy(g[k]).to(dev) for k in ("fw_x", "fw_t", "fw_ctx"))

    def nrms(a, b):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        return float(np.linalg.norm(a - b) / np.linalg.norm(b))

    out16 = net(x, timesteps=t, context=ctx).cpu().numpy()
    net.set_precision("exact")
    out = net(x, timesteps=t, context=ctx).cpu().numpy()
    e, e16 = nrms(out, g["fw_out"]), nrms(out16, g["fw_out"])
    print(f"narrow UNet output vs reference: exact mode nrms {e:.2e}, 16-bit mode {e16:.2e}")
    assert e <= 5e-5, e
    worst = 0.0
    for b in range(3, 12):
        tb = net.output_blocks[b][1].transformer_blocks[0]
        for nm, a in (("self", tb.attn1), ("cross", tb.attn2)):
            for w in ("q", "k"):
                ref = g[f"fw_output_block_{b}_spatial_{nm}_attn_{w}"]
                got = getattr(a, w).cpu().numpy()
                assert got.dtype == np.float16 and got.shape == ref.shape
                # both are fp16 roundings of fp32 values 1e-5 apart: identical except where a value sits on a rounding boundary
                d = np.abs(got.astype(np.float32) - ref.astype(np.float32))
                assert float(d.max()) <= 2.0 ** -10 * float(np.abs(ref.astype(np.float32)).max()), (b, nm, w)
                worst = max(worst, nrms(got.astype(np.float32), ref.astype(np.float32)))
                assert np.mean(got != ref) <= 0.05, (b, nm, w, float(np.mean(got != ref)))
    print(f"exact mode: worst fp16 tap nrms vs reference {worst:.2e}")
    assert worst <= 1e-4
    net.set_precision("fp16")
    assert np.array_equal(net(x, timesteps=t, context=ctx).cpu().numpy(), out16)


def test_exact_video_unet_forward_vs_reference(X):
    """The SVD VideoUNet (narrow, T = 3) in the exact mode against the REFERENCE's fp32 forward (tests/golden/unet_svd_narrow.npz):
    output, spatial taps and the temporal taps in the reference's [(b s), t, c] layout -- the time stack (3-D ResBlock on the
    [3,1,1] temporal conv, AlphaBlender, frame-index embedding, ff_in, temporal self-attention, cross-attention to the first frame's
    context) on split operands too."""
    from vidseg_diffusion_amd.video_unet import VideoUNet
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(os.path.dirname(G), "unet_svd_narrow.npz"))
    g = {k: z[k] for k in z.files}
    net = VideoUNet(**synthetic.SVD_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=4321).items()})
    T = int(g["T"])
    x, t, ctx, y = (torch.from_numpy(g[k]).to(dev) for k in ("fw_x", "fw_t", "fw_ctx", "fw_y"))

    def nrms(a, b):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        return float(np.linalg.norm(a - b) / np.linalg.norm(b))

    kw = dict(timesteps=t, context=ctx, y=y, num_video_frames=T, image_only_indicator=torch.zeros(2, T))
    out16 = net(x, **kw).cpu().numpy()
    net.set_precision("exact")
    out = net(x, **kw).cpu().numpy()
    e, e16 = nrms(out, g["fw_out"]), nrms(out16, g["fw_out"])
    print(f"narrow VideoUNet output vs reference: exact mode nrms {e:.2e}, 16-bit mode {e16:.2e}")
    assert e <= 5e-5, e
    worst = 0.0
    for i in (3, 7, 8, 11):
        blk = net.output_blocks[i]
        pairs = [("spatial_self_attn_q", blk[1].transformer_blocks[0].attn1.q), ("temporal_self_attn_q", blk[1].time_stack[0].attn1.q),
                 ("temporal_self_attn_k", blk[1].time_stack[0].attn1.k), ("temporal_cross_attn_k", blk[1].time_stack[0].attn2.k)]
        for name, got in pairs:
            ref = g[f"fw_output_block_{i}_{name}"]
            got = got.cpu().numpy()
            assert got.dtype == np.float16 and got.shape == ref.shape, (i, name, got.shape, ref.shape)
            worst = max(worst, nrms(got.astype(np.float32), ref.astype(np.float32)))
            assert np.mean(got != ref) <= 0.05, (i, name, float(np.mean(got != ref)))
    print(f"exact VideoUNet: worst fp16 tap nrms vs reference {worst:.2e}")
    assert worst <= 1e-4
    net.set_precision("fp16")
    assert np.array_equal(net(x, **kw).cpu().numpy(), out16)


def _narrow_sd_exact():
    from vidseg_diffusion_amd.unet import UNetModel
    net = UNetModel(**synthetic.SD21_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, 1234).items()})
    net.pack(torch.device("cuda:0"))
    net.set_precision("exact")
    return net


def _nrms(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def test_exact_mode_inversion_vs_reference(X):
    """a3b in the parity precision (VERDICT r5 #3a): `EDMSampler.inversion` (sampling.py:264-296: sigmas ascending, sigma[0] += 1e-8, the
    first pair skips the network, 24 CFG evaluations up to sigma = 14.6, result / sqrt(1 + sigma_last^2)) of the narrow SD UNet in the exact
    mode against the REFERENCE's own fp32 run (tests/golden/unet_sd_narrow.npz: inv_step5, inv_final).  ABSOLUTE bars -- the 16-bit mode's
    test (tests/test_gpu_unet.py::test_inversion_vs_reference) can only be stated relative to its storage format's 2e-2."""
    from vidseg_diffusion_amd.pipeline import build_sd_engine
    dev = torch.device("cuda:0")
    z = np.load(G)
    g = {k: z[k] for k in ("sm_c", "sm_latent", "inv_step5", "inv_final")}
    net = _narrow_sd_exact()
    eng = build_sd_engine(net)
    c = {"crossattn": torch.from_numpy(g["sm_c"]).to(dev)}
    uc = {"crossattn": torch.zeros_like(c["crossattn"])}
    x, lats = eng.sampler.inversion(lambda inp, s, cc, **k: eng.denoiser(eng.model, inp, s, cc), torch.from_numpy(g["sm_latent"]).to(dev),
                                    cond=c, uc=uc, num_steps=25)
    assert len(lats) == 26 and lats[-1] is x
    e5, ef = _nrms(lats[5].cpu().numpy(), g["inv_step5"]), _nrms(x.cpu().numpy(), g["inv_final"])
    print(f"exact-mode inversion vs reference: x after pair 5 nrms {e5:.2e}, final (24 evaluations) nrms {ef:.2e}")
    net.release_exact()
    assert e5 <= 1e-5, e5                                  # measured 6.0e-7
    assert ef <= 1e-4, ef                                  # measured 2.4e-5 after 24 evaluations (16-bit mode: 2.3e-2)


def test_exact_mode_inversion_window_vs_reference(X):
    """`--inversion_type inversion` through the harness in the parity precision (sd_pipeline_vspw.py:233-236, 340-345): sampler.inversion,
    the feature pass from t_start = 0 (dumps at all 25 steps), Steps 3 / 3b -- against the window the REFERENCE itself ran that way
    (tests/golden/sd_inversion_window_narrow.npz, tools/gen_golden_inversion_window.py): 49 network evaluations.  Absolute bars on the
    trajectory and the step-24 Q taps, and the masks must be the reference's."""
    from tools_metrics import matched_iou
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import build_sd_engine, segment_window
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(os.path.dirname(G), "sd_inversion_window_narrow.npz"))
    g = {k: z[k] for k in z.files}
    Fn, K = int(g["F"]), int(g["K"])
    net = _narrow_sd_exact()
    eng = build_sd_engine(net, num_steps=25, scale=5.0)
    c = {"crossattn": torch.from_numpy(g["c"]).to(dev)}
    uc = {"crossattn": torch.zeros_like(c["crossattn"])}
    FE.FeatureStore.clear()
    FE.MaskStore.clear()
    labels, _ = segment_window(eng, torch.from_numpy(g["latent"]).to(dev), c, uc, num_masks=K, num_steps=25, t_start=22, seed=17,
                               is_refine_mask=True, feature_folder="/nonexistent/xinv", exp_name="w", keep_all_steps=True,
                               inversion_type="inversion")
    store = FE.FeatureStore.folder("/nonexistent/xinv", "w")
    errs = {i: _nrms(store[f"xt_time_{i}"].cpu().numpy(), g[f"x_step{i}"]) for i in (0, 12, 24)}
    taps = {b: _nrms(store[f"output_block_{b}_spatial_self_attn_q_time_24"].float().cpu().numpy(), g[f"q{b}"].astype(np.float32)) for b in (6, 7, 8)}
    iou, same = matched_iou(np.asarray(labels).reshape(-1), g["corrected_labels"].astype(np.int64).reshape(-1), K)
    print("exact-mode inversion window vs reference: x nrms", {i: f"{e:.2e}" for i, e in errs.items()}, "step-24 Q taps nrms",
          {b: f"{e:.2e}" for b, e in taps.items()}, f"masks IoU {iou:.4f} identical {same:.4f}")
    FE.FeatureStore.clear()
    FE.MaskStore.clear()
    net.release_exact()
    assert max(errs.values()) <= 1e-4, errs                # measured 1.1e-5 / 3.3e-5 / 3.5e-5 (16-bit mode: 1e-2 ... 2.9e-2)
    assert max(taps.values()) <= 2e-4, taps                # measured 1.2e-4 = the fp16 rounding of the taps themselves (16-bit mode: 3.0e-2)
    assert iou >= 0.99 and same >= 0.995, (iou, same)      # measured: identical to the reference's masks on every token


@pytest.mark.parametrize("precision", ["exact", "fp16"])
@pytest.mark.parametrize("kind", ["sd", "svd"])
def test_step4_sweep_shared_prefix_is_bit_identical(X, kind, precision):
    """pipeline.modulation_sweep(share_prefix=True): the first evaluation of the 2K modulated passes shares the encoder / middle / decoder
    blocks before the first modulated one (+ that block's ResBlock).  Same launches on the same data, so every final latent must equal the
    unshared sweep's bit for bit -- narrow SD (block 7 cross-attention, injected spatial q / k, blending at steps 22-23: the SD driver's
    Step 4) and narrow SVD (block 8 spatial + temporal self-attention, injected temporal q / k, blending at every step)."""
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import build_sd_engine, build_svd_engine, modulation_sweep, segment_window
    dev = torch.device("cuda:0")
    Fn = 3
    lat = torch.from_numpy(synthetic.latent_clip(Fn, 16, 16, seed=5)).to(dev)
    noise = torch.randn(lat.shape, generator=torch.Generator().manual_seed(3)).to(dev)
    if kind == "svd":
        from vidseg_diffusion_amd.video_unet import VideoUNet
        net = VideoUNet(**synthetic.SVD_NARROW)
        g = np.random.Generator(np.random.PCG64(7))
        ctx = torch.from_numpy(g.standard_normal((1, 1, 64)).astype(np.float32)).repeat(Fn, 1, 1).to(dev)
        cat = lat[:1].repeat(Fn, 1, 1, 1) / 0.18215 * 0.2
        vec = torch.from_numpy(g.standard_normal((1, 64)).astype(np.float32)).repeat(Fn, 1).to(dev)
        c = {"crossattn": ctx, "concat": cat, "vector": vec}
        uc = {"crossattn": torch.zeros_like(ctx), "concat": torch.zeros_like(cat), "vector": vec.clone()}
        t0 = 20
        kw = dict(modulate_block_idx=(8,), modulate_layer_type=("spatial", "temporal"), modulate_attn_type=("self_attn",))
    else:
        from vidseg_diffusion_amd.unet import UNetModel
        net = UNetModel(**synthetic.SD21_NARROW)
        cc, ucc = synthetic.sd_conditioning(Fn, context_dim=64, seq=7, seed=2)
        c, uc = {"crossattn": torch.from_numpy(cc).to(dev)}, {"crossattn": torch.from_numpy(ucc).to(dev)}
        t0 = 22
        kw = dict(modulate_block_idx=(7,), modulate_layer_type=("spatial",), modulate_attn_type=("cross_attn",))
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, 1234).items()})
    net.pack(dev)
    net.set_precision(precision)                                         # the exact runner and the 16-bit forward both carry the fork
    eng = build_svd_engine(net, num_frames=Fn) if kind == "svd" else build_sd_engine(net)
    FE.FeatureStore.clear()
    FE.MaskStore.clear()
    base, exp = f"/nonexistent/x_share_{kind}_{precision}", "exp"
    labels, _ = segment_window(eng, lat, c, uc, num_masks=3, t_start=t0, seed=17, noise=noise, feature_folder=base, exp_name=exp, keep_all_steps=True)
    folder = os.path.join(base, exp, "match_gt_mask", "output_block_8_output_block_7_output_block_6_spatial_self_attn_q_masks_3")
    uniq = np.unique(labels)
    for inject in (True, False):
        skw = dict(t_start=t0, feature_folder=base, exp_name=exp, noise=noise, seed=17, is_injected_features=inject, **kw)
        plain = modulation_sweep(eng, lat, c, uc, uniq, folder, share_prefix=False, lanes=1, **skw)     # every pass in full, one after the other
        shared = modulation_sweep(eng, lat, c, uc, uniq, folder, share_prefix=True, lanes=2, **skw)     # the default: shared prefix, two passes in flight
        assert set(plain) == set(shared) == {(s, int(l)) for s in (1, -1) for l in uniq}
        for k in plain:
            assert torch.isfinite(shared[k]).all() and torch.equal(plain[k], shared[k]), (kind, inject, k)
        assert (shared[(1, int(uniq[0]))] - shared[(-1, int(uniq[0]))]).abs().max() > 0
    FE.FeatureStore.clear()
    FE.MaskStore.clear()
    if precision == "exact":
        net.release_exact()
