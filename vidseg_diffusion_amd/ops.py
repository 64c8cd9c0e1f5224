"""Thin Python wrappers over the UNet operator entry points of libvidseg_hip.so.

Activations are NHWC bf16 device tensors (tokens [B, H*W, C] and images [B, H, W, C] share the
same memory); weights are packed once by `pack_*`.  Every function launches asynchronously on
torch's current HIP stream and returns the output tensor it allocated.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib
from ._lib import call, ptr, stream

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

_lib.register({
    "vidseg_linear_a16": [_P, _P, _I, _I, _L, _P, _I, _P, _P, _I, _I, _P, _I, _P, _P, _I, _P, _P, _I, _I, _P, _I, _P],
    "vidseg_conv3x3_a16": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _I, _P, _P, _I, _P, _P],
    "vidseg_conv3x3_a16_tap": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _I, _P, _P, _I, _P, _I, _P],
    "vidseg_softmax_rows_a16": [_P, _L, _I, _F, _P, _P],
    "vidseg_gaussian_sample": [_P, _P, _I, _I, _I, _F, _P, _P],
    "vidseg_conv_in": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P],
    "vidseg_conv_out4": [_P, _P, _P, _I, _I, _I, _I, _P, _P],
    "vidseg_groupnorm_nhwc_a16": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _F, _I, _I, _P, _I, _P, _I, _P, _P],
    "vidseg_layernorm_a16": [_P, _L, _I, _P, _P, _F, _P, _P],
    "vidseg_attention_a16": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "vidseg_attention_fp8": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "vidseg_quant_fp8": [_P, _L, _P, _P],
    "vidseg_time_mix3_f32": [_P, _I, _I, _I, _L, _I, _P, _P, _P, _P],
    "vidseg_timestep_embedding": [_P, _I, _I, _F, _P, _P],
    "vidseg_silu_a16": [_P, _L, _P, _P],
    "vidseg_f32_to_a16": [_P, _L, _P, _P],
    "vidseg_f16_to_a16": [_P, _L, _P, _P],
    "vidseg_prepare_net_input": [_P, _P, _P, _I, _I, _I, _I, _F, _P, _P],
    "vidseg_cfg_euler_step": [_P, _P, _I, _I, _I, _F, _F, _P, _F, _F, _F, _P, _P],
    "vidseg_add_noise": [_P, _P, _L, _F, _F, _P],
    "vidseg_scale_f32": [_P, _L, _F, _P],
    "vidseg_latent_blend": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vidseg_rows_axpby": [_P, _P, _P, _P, _L, _L, _P, _P],
    "vidseg_cfg_combine": [_P, _L, _L, _P, _I, _F, _P, _P],
    "vidseg_euler_update": [_P, _P, _P, _P, _L, _L, _P, _P],
    "vidseg_axpy_f32": [_P, _P, _L, _F, _F, _P, _P],
    "vidseg_blend_f32": [_P, _P, _P, _L, _P, _P],
    "vidseg_bind_workspace": [_P, _P, _L],
    "vidseg_linear_a16_ttap": [_P, _L, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P],
    "vidseg_conv_temporal3_a16": [_P, _I, _I, _I, _I, _P, _I, _P, _P, _I, _P, _P, _P],
    "vidseg_temporal_attention_a16": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "vidseg_alpha_blend_a16": [_P, _P, _P, _L, _P, _P],
    "vidseg_add_rowvec_a16": [_P, _P, _L, _I, _I, _I, _P, _P],
    "vidseg_gemm_profiler_create": [_P],
    "vidseg_gemm_profiler_destroy": [_P],
    "vidseg_gemm_profile_begin": [_P],
    "vidseg_gemm_profile_end": [_P, _P],
    "vidseg_gemm_profile_kinds": [_P, _P],
    "vidseg_gemm_profile_bytes": [_P, _P],
})



def act_dtype():
    """Storage dtype of activations and packed weights (float16 unless the library was built with -DVIDSEG_ACT_BF16)."""
    return _lib.act_dtype()


def __getattr__(name):                                      # ops.ACT / ops.BF16 (historical name) resolve lazily: the .so decides
    if name in ("ACT", "BF16"):
        return _lib.act_dtype()
    raise AttributeError(name)


F32 = torch.float32
ACT_NONE, ACT_SILU, ACT_GEGLU = 0, 1, 2


# ----------------------------------------------------------------------------- weight packing
def pack_linear(weight: torch.Tensor, device) -> torch.Tensor:
    """nn.Linear weight [N, K] -> bf16 [N, K] (already the K-contiguous 'B^T' layout MFMA wants)."""
    return weight.detach().to(device=device, dtype=act_dtype()).contiguous()


def pack_conv3x3(weight: torch.Tensor, device) -> torch.Tensor:
    """Conv2d weight [Cout, Cin, 3, 3] -> bf16 [Cout, c//64, kh*3+kw, c%64] flattened: the NHWC implicit GEMM walks K chunk-major
    (the 9 taps of one 64-channel chunk are consecutive K-tiles, so the shifted re-reads of an input line hit L1/L2)."""
    co, ci, kh, kw = weight.shape
    assert ci % 64 == 0, "conv3x3: Cin must be a multiple of 64"
    w = weight.detach().reshape(co, ci // 64, 64, kh * kw).permute(0, 1, 3, 2)
    return w.reshape(co, kh * kw * ci).to(device=device, dtype=act_dtype()).contiguous()


def pack_conv_in(weight: torch.Tensor, device) -> torch.Tensor:
    """Conv2d weight [Cout, Cin, 3, 3] (Cin = 4/8) -> fp32 [3, 3, Cin, Cout] (cout fastest)."""
    return weight.detach().permute(2, 3, 1, 0).to(device=device, dtype=F32).contiguous()


def pack_conv_out(weight: torch.Tensor, device) -> torch.Tensor:
    """Conv2d weight [4, Cin, 3, 3] -> bf16 [4, 3, 3, Cin]."""
    return weight.detach().permute(0, 2, 3, 1).to(device=device, dtype=act_dtype()).contiguous()


def pack_geglu(weight: torch.Tensor, bias: torch.Tensor, device):
    """GEGLU proj [2*inner, K]: rows [0,inner) = value, [inner, 2*inner) = gate (attention.py:92-96).
    Interleave in 32-row groups (value group, gate group, ...) so that one MFMA wave tile holds the
    value and the gate of the same output column in the same lane."""
    two_inner, K = weight.shape
    inner = two_inner // 2
    assert inner % 32 == 0
    w = weight.detach().view(2, inner // 32, 32, K).permute(1, 0, 2, 3).reshape(two_inner, K)
    b = bias.detach().view(2, inner // 32, 32).permute(1, 0, 2).reshape(two_inner)
    return w.to(device=device, dtype=act_dtype()).contiguous(), b.to(device=device, dtype=F32).contiguous()


def f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=F32).contiguous()


# ----------------------------------------------------------------------------- operators
def linear(a, w, bias=None, *, a1=None, rowvec=None, rows_per_sample=0, residual=None, act=ACT_NONE, out_f32=False,
           tap=None, tap2=None, tap_cols=0, rowadd=None):
    """out = act(cat(a, a1) @ w.T + bias + rowvec[sample]) + rowadd[row] + residual.  a: bf16 [..., K0]."""
    workspace(a.device)
    C0 = a.shape[-1]
    C1 = a1.shape[-1] if a1 is not None else 0
    M = a.numel() // C0
    N = w.shape[0]
    n_out = N // 2 if act == ACT_GEGLU else N
    out = torch.empty(a.shape[:-1] + (n_out,), dtype=F32 if out_f32 else act_dtype(), device=a.device)
    call("vidseg_linear_a16", ptr(a), ptr(a1), C0, C1, M, ptr(w), N, ptr(bias), ptr(rowvec),
         rowvec.stride(0) if rowvec is not None else 0, rows_per_sample, ptr(residual),
         residual.shape[-1] if residual is not None else 0,
         None if out_f32 else ptr(out), ptr(out) if out_f32 else None, n_out,
         ptr(tap), ptr(tap2), tap_cols, tap.shape[-1] if tap is not None else 0, ptr(rowadd), act, stream())
    return out


def conv3x3(x0, w, bias, *, x1=None, stride=1, up=1, rowvec=None, residual=None, pad=1, want_f32=False, tap=None):
    """3x3 conv, padding 1, on NHWC bf16 [B, H, W, C0] (+ channel-concat x1), optional fused nearest-2x
    upsample of the input (openaimodel.py:149-167) or stride 2 (openaimodel.py:202-217).  pad=0: the first stage's
    (0,1,0,1)-padded Downsample (model.py:84-91).  want_f32: also return the result in fp32 (same shape).
    tap = "early" | "late": also return an fp16 NHWC copy taken inside the epilogue after conv + bias -- before the per-sample
    vector ("early": ResBlock.in_layers_features, openaimodel.py:349-350) or after it and before the residual ("late":
    ResBlock.out_layers_features, openaimodel.py:367-368)."""
    workspace(x0.device)
    if tap is not None:
        assert tap in ("early", "late") and not want_f32
        B, H, W, C0 = x0.shape
        Cout = w.shape[0]
        Ho, Wo = (H * up + 2 - 3) // stride + 1, (W * up + 2 - 3) // stride + 1
        out = torch.empty((B, Ho, Wo, Cout), dtype=act_dtype(), device=x0.device)
        t16 = torch.empty((B, Ho, Wo, Cout), dtype=torch.float16, device=x0.device)
        call("vidseg_conv3x3_a16_tap", ptr(x0), ptr(x1), C0, x1.shape[-1] if x1 is not None else 0, B, H, W, stride, up, ptr(w), Cout,
             ptr(bias), ptr(rowvec), rowvec.stride(0) if rowvec is not None else 0, ptr(residual), ptr(out), pad, ptr(t16),
             1 if tap == "early" else 0, stream())
        return out, t16
    B, H, W, C0 = x0.shape
    C1 = x1.shape[-1] if x1 is not None else 0
    Cout = w.shape[0]
    Ho = (H * up + 2 - 3) // stride + 1
    Wo = (W * up + 2 - 3) // stride + 1
    out = torch.empty((B, Ho, Wo, Cout), dtype=act_dtype(), device=x0.device)
    out32 = torch.empty((B, Ho, Wo, Cout), dtype=F32, device=x0.device) if want_f32 else None
    call("vidseg_conv3x3_a16", ptr(x0), ptr(x1), C0, C1, B, H, W, stride, up, ptr(w), Cout, ptr(bias), ptr(rowvec),
         rowvec.stride(0) if rowvec is not None else 0, ptr(residual), ptr(out), pad, ptr(out32), stream())
    return (out, out32) if want_f32 else out


def softmax_rows(x_f32, scale):
    """softmax(scale * x) over the last dim of an fp32 matrix -> bf16 (first-stage mid attention)."""
    cols = x_f32.shape[-1]
    out = torch.empty(x_f32.shape, dtype=act_dtype(), device=x_f32.device)
    call("vidseg_softmax_rows_a16", ptr(x_f32), x_f32.numel() // cols, cols, float(scale), ptr(out), stream())
    return out


def gaussian_sample(moments_nhwc_f32, noise_nchw, scale):
    """(mean + exp(0.5 * clamp(logvar, -30, 20)) * noise) * scale; moments [B, h, w, 2z] fp32 -> fp32 NCHW [B, z, h, w]."""
    B, H, W, Z2 = moments_nhwc_f32.shape
    out = torch.empty((B, Z2 // 2, H, W), dtype=F32, device=moments_nhwc_f32.device)
    call("vidseg_gaussian_sample", ptr(moments_nhwc_f32), ptr(noise_nchw), B, H * W, Z2 // 2, float(scale), ptr(out), stream())
    return out


def conv_in(x_nhwc_f32, w, bias):
    """Input conv (openaimodel.py:638-644): fp32 NHWC [B,H,W,4|8] -> bf16 NHWC [B,H,W,Cout]."""
    B, H, W, Cin = x_nhwc_f32.shape
    Cout = w.shape[-1]
    out = torch.empty((B, H, W, Cout), dtype=act_dtype(), device=x_nhwc_f32.device)
    call("vidseg_conv_in", ptr(x_nhwc_f32), ptr(w), ptr(bias), B, H, W, Cin, Cout, ptr(out), stream())
    return out


def conv_out4(x, w, bias):
    """Output conv (openaimodel.py:825-829): bf16 NHWC [B,H,W,Cin] -> fp32 NCHW [B,4,H,W]."""
    B, H, W, Cin = x.shape
    assert w.shape[0] == 4
    out = torch.empty((B, 4, H, W), dtype=F32, device=x.device)
    call("vidseg_conv_out4", ptr(x), ptr(w), ptr(bias), B, H, W, Cin, ptr(out), stream())
    return out


class Workspace:
    """Scratch of ONE (device, stream): GroupNorm partial sums and per-(sample, channel) scale/shift, fp32 split-K partials.
    Kernels that use it are ordered on that stream, so two streams (the window lanes of pipeline.WindowPipeline, the analysis
    side stream) never share a buffer; the split-K scratch is bound to the stream inside the library (vidseg_bind_workspace)."""

    def __init__(self, device, cuda_stream, floats=1 << 24, splitk_floats=40 << 20):
        self.part = torch.empty(floats, dtype=F32, device=device)
        self.stats = torch.empty(1 << 20, dtype=F32, device=device)          # GroupNorm per-(sample, channel) scale/shift
        self.splitk = torch.empty(splitk_floats, dtype=F32, device=device)   # fp32 split-K partials (160 MB)
        with torch.cuda.device(device):
            call("vidseg_bind_workspace", cuda_stream, ptr(self.splitk), self.splitk.numel())


_ws = {}


def workspace(device) -> Workspace:
    st = torch.cuda.current_stream(device).cuda_stream
    key = (device.index if device.index is not None else torch.cuda.current_device(), st)
    if key not in _ws:
        _ws[key] = Workspace(device, st)
    return _ws[key]


_side = {}


def side_stream(device):
    """One helper HIP stream per (device, caller stream): independent branches of the network (the ResBlock's 1x1 skip conv) are
    queued there and joined with wait_stream; keyed on the caller's stream so that two window lanes never share one."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    if key not in _side:
        _side[key] = torch.cuda.Stream(device=device)
    return _side[key]


# ----------------------------------------------------------------------------- per-window caches
# Values that are constant over the sampler steps of ONE window (the CFG-stacked conditioning, its 16-bit copy, the cross-attention
# K/V projections of that context: attention.py:317-322 recomputes to_k(context) / to_v(context) in every block of every step) are
# computed once per window.  `new_window()` (called by pipeline.feature_pass / parallel.sharded_feature_pass) starts a new epoch,
# so nothing is ever carried from one window -- or one stream -- to the next; inside an epoch an entry is valid for exactly the
# tensors it was made from (object identity + torch's in-place version counter; the entry keeps them alive).
_EPOCH = 0
_CACHE_ON = os.environ.get("VIDSEG_WINDOW_CACHE", "1") != "0"


def new_window():
    global _EPOCH
    _EPOCH += 1


def window_cached(owner, slot, deps, make):
    """make() once per (epoch, deps) for `owner.<slot>`; deps: tuple of tensors the value is a pure function of."""
    if not _CACHE_ON:
        return make()
    key = (_EPOCH,) + tuple((id(t), t._version) for t in deps)
    ent = getattr(owner, slot, None)
    if ent is not None and ent[0] == key:
        return ent[2]
    val = make()
    setattr(owner, slot, (key, deps, val))
    return val


_GN_RPC_FORCE = int(os.environ.get("VIDSEG_GN_RPC", "0"))       # experiments: a fixed chunk (another fp32 summation order of the statistics)


def gn_rows_per_chunk(B, HW):
    """Rows of one sample per GroupNorm block: HW / 64 clamped to [4, 64] and rounded down to a power of two -- at least 64 (16 for an
    8 x 8 map) blocks per sample.  A function of HW alone: the chunking fixes the fp32 summation order of the statistics, and a
    sample's result must not depend on how many other samples share the launch (chunked first-stage decodes, the batch-chunked
    reference runs)."""
    if _GN_RPC_FORCE:
        return _GN_RPC_FORCE
    rpc = 4
    while rpc < 64 and rpc * 2 * 64 <= HW:
        rpc *= 2
    return rpc


def groupnorm(x0, gamma, beta, *, x1=None, groups=32, eps=1e-5, silu=True):
    """GroupNorm32 (+SiLU) over NHWC bf16 [B, H, W, C0] (+concat x1) -> bf16 [B, H, W, C0+C1]."""
    B = x0.shape[0]
    C0 = x0.shape[-1]
    C1 = x1.shape[-1] if x1 is not None else 0
    HW = x0.numel() // (B * C0)
    ws = workspace(x0.device)
    rpc = gn_rows_per_chunk(B, HW)
    need = B * ((HW + rpc - 1) // rpc) * 2 * (C0 + C1)                # per-chunk partial sums
    if need > ws.part.numel():                                        # first-stage activations at 576x1024 need 33 M floats
        ws.part = torch.empty(need, dtype=F32, device=x0.device)
    if B * 2 * (C0 + C1) > ws.stats.numel():
        ws.stats = torch.empty(B * 2 * (C0 + C1), dtype=F32, device=x0.device)
    out = torch.empty(x0.shape[:-1] + (C0 + C1,), dtype=act_dtype(), device=x0.device)
    call("vidseg_groupnorm_nhwc_a16", ptr(x0), ptr(x1), C0, C1, B, HW, groups, ptr(gamma), ptr(beta), eps, int(silu), rpc,
         ptr(ws.part), ws.part.numel(), ptr(ws.stats), ws.stats.numel(), ptr(out), stream())
    return out


def layernorm(x, gamma, beta, eps=1e-5):
    C = x.shape[-1]
    out = torch.empty_like(x)
    call("vidseg_layernorm_a16", ptr(x), x.numel() // C, C, ptr(gamma), ptr(beta), eps, ptr(out), stream())
    return out


_ATTN_FP8 = os.environ.get("VIDSEG_ATTN_FP8", "0") == "1"
_ATTN_FP8_MIN_KEYS = 1024       # the one-off quantisation pays from here on (measured: 1024 keys break even, 4096 keys -18 %)


def set_attention_fp8(on, min_keys=1024):
    """Route the 64-wide-head attention (spatial self/cross attention of both UNets) with at least `min_keys` keys through the
    OCP-e4m3 kernel (BASELINE configs[4]); also selectable with VIDSEG_ATTN_FP8=1.  Returns the previous on/off setting."""
    global _ATTN_FP8, _ATTN_FP8_MIN_KEYS
    prev, _ATTN_FP8, _ATTN_FP8_MIN_KEYS = _ATTN_FP8, bool(on), int(min_keys)
    return prev


def quant_fp8(x):
    """fp16/bf16 -> OCP e4m3 bytes (saturating, round to nearest even); same shape, contiguous."""
    assert x.is_contiguous() and x.dtype == act_dtype()
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    call("vidseg_quant_fp8", ptr(x), x.numel(), ptr(out), stream())
    return out


def attention(q, k, v, heads, *, q_ld=None, k_ld=None, v_ld=None, Nq=None, Nk=None, B=None, fp8=None):
    """softmax(q k^T / 8) v per 64-wide head.  q/k/v may be column slices of wider row-major buffers
    (pass the data pointers' leading dimensions).  fp8: quantise q, k, v (whole underlying buffers, once each) to e4m3
    and run the fp8 MFMA kernel; the output stays in the activation dtype."""
    B = q.shape[0] if B is None else B
    Nq = q.shape[1] if Nq is None else Nq
    Nk = k.shape[1] if Nk is None else Nk
    out = torch.empty((B, Nq, heads * 64), dtype=act_dtype(), device=q.device)
    if (_ATTN_FP8 and Nk >= _ATTN_FP8_MIN_KEYS) if fp8 is None else fp8:
        done = {}

        def q8(t):                                   # pointer of t's first element inside the quantised copy of its base buffer
            base = t._base if t._base is not None else t
            if base.data_ptr() not in done:
                done[base.data_ptr()] = quant_fp8(base if base.is_contiguous() else base.contiguous())
            return done[base.data_ptr()].data_ptr() + (t.data_ptr() - base.data_ptr()) // 2, done[base.data_ptr()]

        (pq, _kq), (pk, _kk), (pv, _kv) = q8(q), q8(k), q8(v)
        call("vidseg_attention_fp8", pq, q_ld or q.stride(1), pk, k_ld or k.stride(1), pv, v_ld or v.stride(1), ptr(out), heads * 64,
             B, heads, Nq, Nk, 64, stream())
        return out
    call("vidseg_attention_a16", q.data_ptr(), q_ld or q.stride(1), k.data_ptr(), k_ld or k.stride(1), v.data_ptr(),
         v_ld or v.stride(1), ptr(out), heads * 64, B, heads, Nq, Nk, 64, stream())
    return out


def timestep_embedding(t, dim, max_period=10000.0):
    out = torch.empty((t.shape[0], dim), dtype=act_dtype(), device=t.device)
    call("vidseg_timestep_embedding", ptr(t), t.shape[0], dim, float(max_period), ptr(out), stream())
    return out


def silu(x):
    out = torch.empty_like(x)
    call("vidseg_silu_a16", ptr(x), x.numel(), ptr(out), stream())
    return out


def f16_to_bf16(x):
    out = torch.empty(x.shape, dtype=act_dtype(), device=x.device)
    xc = x.contiguous()
    call("vidseg_f16_to_a16", ptr(xc), xc.numel(), ptr(out), stream())
    return out


def to_bf16(x):
    out = torch.empty(x.shape, dtype=act_dtype(), device=x.device)
    call("vidseg_f32_to_a16", ptr(x), x.numel(), ptr(out), stream())
    return out


# ----------------------------------------------------------------------------- sampler arithmetic (fp32 latents)
def _rowvec(v, B, device):
    """per-row scalars as a small fp32 device vector [B] (accepts python floats or CPU/GPU tensors)."""
    if not torch.is_tensor(v):
        v = torch.full((B,), float(v), dtype=F32)
    v = v.reshape(-1).to(dtype=F32)
    if v.numel() == 1 and B > 1:
        v = v.expand(B)
    return v.contiguous().to(device, non_blocking=True)


def rows_axpby(a, sa, b=None, sb=None):
    """a * sa[row] (+ b * sb[row]) with per-batch-row scalars."""
    B = a.shape[0]
    a = a.contiguous()
    out = torch.empty_like(a)
    dsa = _rowvec(sa, B, a.device)
    dsb = _rowvec(sb, B, a.device) if b is not None else None
    bc = b.contiguous() if b is not None else None
    call("vidseg_rows_axpby", ptr(a), ptr(dsa), ptr(bc), ptr(dsb), a.numel(), a.numel() // B, ptr(out), stream())
    return out


def cfg_combine(x, scale, num_frames=0):
    """x_u + s (x_c - x_u) on a [2F, ...] stack (guiders.py:28-31); `scale` float or per-frame device vector."""
    x = x.contiguous()
    half = x.numel() // 2
    out = torch.empty((x.shape[0] // 2,) + tuple(x.shape[1:]), dtype=F32, device=x.device)
    fs = scale if torch.is_tensor(scale) else None
    call("vidseg_cfg_combine", ptr(x), half, x.numel() // x.shape[0], ptr(fs), num_frames,
         0.0 if fs is not None else float(scale), ptr(out), stream())
    return out


def euler_update(x, denoised, sigma, sigma_next):
    B = x.shape[0]
    x = x.contiguous()
    out = torch.empty_like(x)
    den = denoised.contiguous()
    s0, s1 = _rowvec(sigma, B, x.device), _rowvec(sigma_next, B, x.device)   # keep alive until the launch is enqueued
    call("vidseg_euler_update", ptr(x), ptr(den), ptr(s0), ptr(s1), x.numel(), x.numel() // B, ptr(out), stream())
    return out


def axpy(x, e, s, post=1.0):
    """(x + e * s) * post."""
    x = x.contiguous()
    out = torch.empty_like(x)
    ec = e.contiguous()
    call("vidseg_axpy_f32", ptr(x), ptr(ec), x.numel(), float(s), float(post), ptr(out), stream())
    return out


def scale(x, s):
    out = x.clone()
    call("vidseg_scale_f32", ptr(out), out.numel(), float(s), stream())
    return out


def blend(x, y, m):
    x = x.contiguous()
    out = torch.empty_like(x)
    yc, mc = y.contiguous(), m.contiguous()
    call("vidseg_blend_f32", ptr(x), ptr(yc), ptr(mc), x.numel(), ptr(out), stream())
    return out


class GemmProfiler:
    """Caller-owned profiler handle of the C ABI (vidseg_gemm_profiler_create): between begin() and end() the conv / linear MFMA
    launches of this thread are timed into it with HIP events riding on the dispatch packets (bench.py roofline)."""

    def __init__(self):
        h = ctypes.c_void_p()
        call("vidseg_gemm_profiler_create", ctypes.byref(h))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.lib().vidseg_gemm_profiler_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def begin(self):
        call("vidseg_gemm_profile_begin", self.h)

    def end(self):
        """-> (kernel_ms_total, algorithmic_flops, launches)."""
        out = (ctypes.c_double * 3)()
        call("vidseg_gemm_profile_end", self.h, out)
        return float(out[0]), float(out[1]), int(out[2])

    def kinds(self):
        """Per-kernel split of the region closed by end(): list of (name, ms, flops, launches, algorithmic_bytes)."""
        nk = len(GEMM_KIND_NAMES)
        out = (ctypes.c_double * (3 * nk))()
        call("vidseg_gemm_profile_kinds", self.h, out)
        ab = (ctypes.c_double * nk)()
        call("vidseg_gemm_profile_bytes", self.h, ab)
        return [(GEMM_KIND_NAMES[k], float(out[3 * k]), float(out[3 * k + 1]), int(out[3 * k + 2]), float(ab[k])) for k in range(nk)]


_PROFILER = None


def _profiler():
    global _PROFILER
    if _PROFILER is None:
        _PROFILER = GemmProfiler()
    return _PROFILER


def gemm_profile_begin():
    """Start HIP-event timing of every conv/linear MFMA launch of this thread into the package's default GemmProfiler."""
    _profiler().begin()


def gemm_profile_end():
    """-> (kernel_ms_total, algorithmic_flops, launches)."""
    return _profiler().end()


GEMM_KIND_NAMES = ("k_gemm_dma (128x128, LDS-DMA)", "k_gemm_ph<NJ> (256x320 / 256x256, LDS-DMA, phased)",
                   "k_gemm_tile<NJ,4,32,1> (128x320, LDS-DMA)", "k_gemm_conv<256,64>", "k_gemm_p7 (224x320, LDS-DMA, phased, 16x16x32 MFMA)",
                   "k_gemm_ws (weight-stationary streaming, K = 320 / 640)",
                   "k_gemm_p7x<5, false> (224x320 on split (hi, lo) operands: each plane staged once, 3 MFMA products per 64 channels)",
                   "k_gemm_p7x<4, true> (224x256 on split operands, GEGLU product in the epilogue, split-image output)")


def gemm_profile_kinds():
    """Per-kernel split of the region closed by gemm_profile_end: list of (name, ms, flops, launches, algorithmic_bytes)."""
    return _profiler().kinds()


# ----------------------------------------------------------------------------- video (SVD) operators
def pack_conv_temporal3(weight: torch.Tensor, device) -> torch.Tensor:
    """Conv3d weight [Cout, Cin, 3, 1, 1] -> bf16 [Cout, c//64, dt, c%64] flattened (same chunk-major K order as pack_conv3x3)."""
    co, ci = weight.shape[:2]
    assert ci % 64 == 0, "conv_temporal3: Cin must be a multiple of 64"
    w = weight.detach().reshape(co, ci // 64, 64, 3).permute(0, 1, 3, 2)
    return w.reshape(co, 3 * ci).to(device=device, dtype=act_dtype()).contiguous()


def conv_temporal3(x, w, bias, T, *, rowvec=None, residual=None):
    """Conv3d kernel [3,1,1] over the frame axis of NHWC bf16 [(b t), H, W, C] (video_model.py:45-58)."""
    workspace(x.device)
    BT, H, W, C = x.shape
    Cout = w.shape[0]
    out = torch.empty((BT, H, W, Cout), dtype=act_dtype(), device=x.device)
    call("vidseg_conv_temporal3_a16", ptr(x), C, BT, H * W, T, ptr(w), Cout, ptr(bias), ptr(rowvec),
         rowvec.stride(0) if rowvec is not None else 0, ptr(residual), ptr(out), stream())
    return out


def linear_temporal_tap(a, w, T, S, tap, tap2, tap_cols):
    """Bias-free projection whose fp16 q/k taps are written in the reference's [(b s), t, c] layout."""
    workspace(a.device)
    C0 = a.shape[-1]
    M = a.numel() // C0
    N = w.shape[0]
    out = torch.empty(a.shape[:-1] + (N,), dtype=act_dtype(), device=a.device)
    call("vidseg_linear_a16_ttap", ptr(a), M, C0, ptr(w), N, ptr(out), N, ptr(tap), ptr(tap2), tap_cols,
         tap.shape[-1] if tap is not None else 0, T if tap is not None else 0, S, stream())
    return out


def temporal_attention(q, k, v, heads, Bv, T, S):
    """Attention across the T frames of every (sample, location); q/k/v rows in spatial order (b t) s."""
    out = torch.empty((Bv * T, S, heads * 64), dtype=act_dtype(), device=q.device)
    call("vidseg_temporal_attention_a16", q.data_ptr(), q.stride(-2), k.data_ptr(), k.stride(-2), v.data_ptr(), v.stride(-2),
         ptr(out), heads * 64, Bv, T, S, heads, 64, stream())
    return out


def alpha_blend(x_spatial, x_temporal, mix_factor):
    """sigmoid(mix) * spatial + (1 - sigmoid(mix)) * temporal (AlphaBlender, image_only_indicator == 0)."""
    out = torch.empty_like(x_spatial)
    call("vidseg_alpha_blend_a16", ptr(x_spatial), ptr(x_temporal), ptr(mix_factor), x_spatial.numel(), ptr(out), stream())
    return out


def time_mix3(x_nchw_f32, w, bias, T, C):
    """AE3DConv.time_mix_conv (temporal_ae.py:84-107) on fp32 NCHW [(b t), xC, H, W] -> [(b t), C, H, W]."""
    BT, xC, H, W = x_nchw_f32.shape
    out = torch.empty((BT, C, H, W), dtype=F32, device=x_nchw_f32.device)
    call("vidseg_time_mix3_f32", ptr(x_nchw_f32), BT, xC, C, H * W, T, ptr(w), ptr(bias), ptr(out), stream())
    return out


def add_rowvec(x, vec, rows_per_sample):
    """x[(sample, row), :] + vec[sample % len(vec), :]  (bf16)."""
    C = x.shape[-1]
    out = torch.empty_like(x)
    call("vidseg_add_rowvec_a16", ptr(x), ptr(vec), x.numel() // C, C, rows_per_sample, vec.shape[0], ptr(out), stream())
    return out
