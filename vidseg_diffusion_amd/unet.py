"""MI355X-native UNet behind the reference's network plug-in seam.

`UNetModel` takes the constructor kwargs of configs/inference/sd_2_1.yaml:19-30 (the class the
YAML's `network_config.target` names, sgm/modules/diffusionmodules/openaimodel.py:487-829), keeps
the upstream state-dict key names (load_state_dict works on SD 2.1 checkpoints) and the attribute
protocol the drivers poke (scripts/sampling/sd_pipeline_vspw.py:107-120):

    len(block) > 1 and "SpatialTransformer" in str(type(block[1]))
    block[1].transformer_blocks[0].attn{1,2}.{q,k}      # tensors valid after each forward

The forward pass is a sequence of hand-written HIP kernels on NHWC bf16 activations (ops.py); the
torch.nn modules here only hold parameters and names.  Nothing falls back to torch operators.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import VidsegError

F16 = torch.float16
# VIDSEG_SIDE_SKIP=1: queue the ResBlock's 1x1 skip conv on a second HIP stream beside the in_layers chain.  Measured (round 3, same-box
# A/B, two repeats): 147.0 vs 148.6 frames/s WITHOUT it -- the two event hand-overs per block cost more than the idle CUs it fills; off.
_SIDE_SKIP = __import__("os").environ.get("VIDSEG_SIDE_SKIP", "0") == "1"


def _meta(factory, *a, **k):
    return factory(*a, device="meta", **k)


class GroupNorm32(nn.GroupNorm):
    """Parameter holder for normalization(channels) (diffusionmodules/util.py:261-278)."""


class Upsample(nn.Module):
    """openaimodel.py:117-167 (dims=2, use_conv=True): nearest x2 fused into the conv's addressing."""

    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.conv = _meta(nn.Conv2d, channels, self.out_channels, 3, padding=1)

    def pack(self, dev):
        self.w = ops.pack_conv3x3(self.conv.weight, dev)
        self.b = ops.f32(self.conv.bias, dev)

    def run(self, x):
        return ops.conv3x3(x, self.w, self.b, up=2)


class Downsample(nn.Module):
    """openaimodel.py:170-217 (dims=2, use_conv=True): stride-2 3x3 conv."""

    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.op = _meta(nn.Conv2d, channels, self.out_channels, 3, stride=2, padding=1)

    def pack(self, dev):
        self.w = ops.pack_conv3x3(self.op.weight, dev)
        self.b = ops.f32(self.op.bias, dev)

    def run(self, x):
        return ops.conv3x3(x, self.w, self.b, stride=2)


class ResBlock(nn.Module):
    """openaimodel.py:220-369, the configuration SD 2.1 / SVD use (no scale-shift norm, no up/down)."""

    def __init__(self, channels, emb_channels, out_channels=None):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.in_layers = nn.Sequential(_meta(GroupNorm32, 32, channels), nn.SiLU(),
                                       _meta(nn.Conv2d, channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), _meta(nn.Linear, emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(_meta(GroupNorm32, 32, self.out_channels), nn.SiLU(), nn.Dropout(p=0.0),
                                        _meta(nn.Conv2d, self.out_channels, self.out_channels, 3, padding=1))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = _meta(nn.Conv2d, channels, self.out_channels, 1)
        self.emb_offset = None                                     # column offset into the batched emb GEMM
        # openaimodel.py:326-327, 349-350, 367-368: the "mid-block spatial features" the reference leaves on every ResBlock after a
        # forward -- in_layers(x) BEFORE the emb add and out_layers(h) BEFORE the skip add.  Nothing downstream reads them, so they
        # are produced only when `stash_features` is set (UNetModel.stash_resblock_features(True)): fp16 copies taken inside the two
        # conv epilogues, exposed NCHW-shaped like the reference's tensors (a permuted view of the NHWC buffer).
        self.stash_features = False
        self.in_layers_features = None
        self.out_layers_features = None

    def pack(self, dev):
        self.g1, self.b1 = ops.f32(self.in_layers[0].weight, dev), ops.f32(self.in_layers[0].bias, dev)
        self.w1, self.cb1 = ops.pack_conv3x3(self.in_layers[2].weight, dev), ops.f32(self.in_layers[2].bias, dev)
        self.g2, self.b2 = ops.f32(self.out_layers[0].weight, dev), ops.f32(self.out_layers[0].bias, dev)
        self.w2, self.cb2 = ops.pack_conv3x3(self.out_layers[3].weight, dev), ops.f32(self.out_layers[3].bias, dev)
        if not isinstance(self.skip_connection, nn.Identity):
            self.ws = ops.pack_linear(self.skip_connection.weight.reshape(self.out_channels, self.channels), dev)
            self.bs = ops.f32(self.skip_connection.bias, dev)

    def run(self, x0, x1, emb_all):
        """x0 (+ x1: skip tensor to concatenate on channels, openaimodel.py:912) NHWC bf16;
        emb_all: fp32 [B, sum(Cout)] = every block's emb_layers output from one batched GEMM."""
        side = skip = None
        if _SIDE_SKIP and not isinstance(self.skip_connection, nn.Identity):
            # the 1x1 skip conv (OAI:369 `self.skip_connection(x)`) depends on the block's input only: it is queued on a second HIP
            # stream and fills the CUs the GroupNorm passes and the conv tails of the main chain leave idle.  Same kernel, same split-K
            # choice, same bits; its scratch is that stream's own (ops.workspace is per stream).
            main = torch.cuda.current_stream()
            side = ops.side_stream(x0.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                skip = ops.linear(x0, self.ws, self.bs, a1=x1)
            skip.record_stream(main)
            x0.record_stream(side)
            if x1 is not None:
                x1.record_stream(side)
        h = ops.groupnorm(x0, self.g1, self.b1, x1=x1, eps=1e-5, silu=True)
        rv = emb_all[:, self.emb_offset:self.emb_offset + self.out_channels]
        if self.stash_features:
            h, t = ops.conv3x3(h, self.w1, self.cb1, rowvec=rv, tap="early")           # tap = in_layers(x), OAI:349-350
            self.in_layers_features = t.permute(0, 3, 1, 2)
        else:
            h = ops.conv3x3(h, self.w1, self.cb1, rowvec=rv)                           # conv + bias + emb_out (OAI:353-365)
        h = ops.groupnorm(h, self.g2, self.b2, eps=1e-5, silu=True)
        if isinstance(self.skip_connection, nn.Identity):
            if x1 is not None:
                raise VidsegError("ResBlock: identity skip with a concatenated input")
            res = x0
        elif side is not None:
            torch.cuda.current_stream().wait_stream(side)                              # the 1x1 skip conv ran beside in_layers / GroupNorm
            res = skip
        else:
            res = ops.linear(x0, self.ws, self.bs, a1=x1)
        if self.stash_features:
            out, t = ops.conv3x3(h, self.w2, self.cb2, residual=res, tap="late")       # tap = out_layers(h), OAI:367-368
            self.out_layers_features = t.permute(0, 3, 1, 2)
            return out
        return ops.conv3x3(h, self.w2, self.cb2, residual=res)                         # OAI:369


class CrossAttention(nn.Module):
    """attention.py:257-364.  q/k hold the fp16 projections before the head split (ATT:330-331)."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64):
        super().__init__()
        if dim_head != 64:
            raise NotImplementedError("only 64-channel heads are on the path (num_head_channels: 64)")
        inner = heads * dim_head
        self.heads = heads
        self.inner = inner
        self.is_self = context_dim is None
        context_dim = context_dim or query_dim
        self.scale = dim_head ** -0.5
        self.to_q = _meta(nn.Linear, query_dim, inner, bias=False)
        self.to_k = _meta(nn.Linear, context_dim, inner, bias=False)
        self.to_v = _meta(nn.Linear, context_dim, inner, bias=False)
        self.to_out = nn.Sequential(_meta(nn.Linear, inner, query_dim), nn.Dropout(0.0))
        self.q = None
        self.k = None

    def pack(self, dev):
        if self.is_self:
            self.w_qkv = ops.pack_linear(torch.cat([self.to_q.weight, self.to_k.weight, self.to_v.weight], 0), dev)
        else:
            self.w_q = ops.pack_linear(self.to_q.weight, dev)
            self.w_kv = ops.pack_linear(torch.cat([self.to_k.weight, self.to_v.weight], 0), dev)
        self.w_o = ops.pack_linear(self.to_out[0].weight, dev)
        self.b_o = ops.f32(self.to_out[0].bias, dev)

    def run(self, x, context, residual, tap, inj_q=None, inj_k=None, rowadd=None):
        """x: normed tokens bf16 [B, N, C]; returns to_out(attn) + rowadd[row] + residual.
        inj_q / inj_k: fp16 dumps that replace the computed projections (attention.py:305-315)."""
        B, N, _ = x.shape
        C = self.inner
        dev = x.device
        if self.is_self:
            tq = torch.empty((B, N, C), dtype=F16, device=dev) if tap else None
            tk = torch.empty((B, N, C), dtype=F16, device=dev) if tap else None
            qkv = ops.linear(x, self.w_qkv, tap=tq, tap2=tk, tap_cols=C)
            q = ops.f16_to_bf16(inj_q) if inj_q is not None else qkv[..., :C]
            k = ops.f16_to_bf16(inj_k) if inj_k is not None else qkv[..., C:2 * C]
            a = ops.attention(q, k, qkv[..., 2 * C:], self.heads)
        else:
            L = context.shape[1]
            tq = torch.empty((B, N, C), dtype=F16, device=dev) if tap else None
            tk = torch.empty((B, L, C), dtype=F16, device=dev) if tap else None
            q = ops.linear(x, self.w_q, tap=tq, tap_cols=C)
            if tap:                                                              # the context is constant over a window's steps

                def make_kv():
                    t = torch.empty((B, L, C), dtype=F16, device=dev)
                    return ops.linear(context, self.w_kv, tap=t, tap_cols=C), t

                kv, tk = ops.window_cached(self, "_kv_tap", (context, self.w_kv), make_kv)
            else:
                kv = ops.window_cached(self, "_kv", (context, self.w_kv), lambda: ops.linear(context, self.w_kv))
            if inj_q is not None:
                q = ops.f16_to_bf16(inj_q)
            k = ops.f16_to_bf16(inj_k) if inj_k is not None else kv[..., :C]
            a = ops.attention(q, k, kv[..., C:], self.heads)
        if tap:
            self.q, self.k = tq, tk
        if inj_q is not None:
            self.q = inj_q                                                       # ATT:330-331 stores what was used
        if inj_k is not None:
            self.k = inj_k
        return ops.linear(a, self.w_o, self.b_o, residual=residual, rowadd=rowadd)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = _meta(nn.Linear, dim_in, dim_out * 2)


class FeedForward(nn.Module):
    """attention.py:98-115 with glu=True."""

    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(GEGLU(dim, dim * mult), nn.Dropout(0.0), _meta(nn.Linear, dim * mult, dim))

    def pack(self, dev):
        self.w1, self.b1 = ops.pack_geglu(self.net[0].proj.weight, self.net[0].proj.bias, dev)
        self.w2, self.b2 = ops.pack_linear(self.net[2].weight, dev), ops.f32(self.net[2].bias, dev)

    def run(self, x, residual, rowadd=None):
        g = ops.linear(x, self.w1, self.b1, act=ops.ACT_GEGLU)
        return ops.linear(g, self.w2, self.b2, residual=residual, rowadd=rowadd)


class BasicTransformerBlock(nn.Module):
    """attention.py:504-759 (feature-dump path: no modulation, no injection)."""

    def __init__(self, dim, n_heads, d_head, context_dim=None):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, n_heads, d_head)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, context_dim, n_heads, d_head)
        self.norm1 = _meta(nn.LayerNorm, dim)
        self.norm2 = _meta(nn.LayerNorm, dim)
        self.norm3 = _meta(nn.LayerNorm, dim)

    def pack(self, dev):
        for m in (self.attn1, self.attn2, self.ff):
            m.pack(dev)
        self.ln = [(ops.f32(n.weight, dev), ops.f32(n.bias, dev)) for n in (self.norm1, self.norm2, self.norm3)]

    def run(self, x, context, tap, mod=None):
        """mod: None, or (inject: dict|None, rowadd: dict attn_type -> fp32 [B*N]) for the modulated pass
        (attention.py:616-634, 646-663, 674-687, 697-719, 733-755)."""
        inj, ra = mod if mod is not None else (None, None)
        inj, ra = inj or {}, ra or {}

        def pick(sub):
            for k, v in inj.items():
                if sub in k:
                    return v
            return None

        x = self.attn1.run(ops.layernorm(x, *self.ln[0]), None, x, tap, pick("spatial_self_attn_q"), pick("spatial_self_attn_k"),
                           ra.get("self_attn"))                                         # ATT:636-672
        x = self.attn2.run(ops.layernorm(x, *self.ln[1]), context, x, tap, pick("spatial_cross_attn_q"),
                           pick("spatial_cross_attn_k"), ra.get("cross_attn"))          # ATT:689-726
        return self.ff.run(ops.layernorm(x, *self.ln[2]), x, ra.get("ff_out"))          # ATT:728-757


class SpatialTransformer(nn.Module):
    """attention.py:806-927 with use_linear=True."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, context_dim=None):
        super().__init__()
        inner = n_heads * d_head
        self.in_channels = in_channels
        self.norm = _meta(nn.GroupNorm, 32, in_channels, eps=1e-6)
        self.proj_in = _meta(nn.Linear, in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, n_heads, d_head, context_dim) for _ in range(depth)])
        self.proj_out = _meta(nn.Linear, inner, in_channels)
        self.tap = True

    def pack(self, dev):
        self.g, self.b = ops.f32(self.norm.weight, dev), ops.f32(self.norm.bias, dev)
        self.w_in, self.b_in = ops.pack_linear(self.proj_in.weight, dev), ops.f32(self.proj_in.bias, dev)
        self.w_out, self.b_out = ops.pack_linear(self.proj_out.weight, dev), ops.f32(self.proj_out.bias, dev)
        for blk in self.transformer_blocks:
            blk.pack(dev)

    def run(self, x, context, mod=None):
        B, H, W, C = x.shape
        t = ops.groupnorm(x, self.g, self.b, eps=1e-6, silu=False).view(B, H * W, C)  # ATT:897-903
        t = ops.linear(t, self.w_in, self.b_in)
        bmod = block_modulation(mod, "spatial", H * W, x.device)                      # ATT:906-915
        for i, blk in enumerate(self.transformer_blocks):
            t = blk.run(t, context, self.tap and i == 0, bmod)
        out = ops.linear(t, self.w_out, self.b_out, residual=x.view(B, H * W, C))     # ATT:921-927
        return out.view(B, H, W, C)


def block_modulation(mod, layer_type, N, device):
    """Translate the reference's mutable `modulate_params` protocol into (inject, rowadd) for one transformer:
    mod = (is_modulate_step, is_injected_step, modulate_params) as decided per block by UNetModel.forward
    (openaimodel.py:911-937).  rowadd[attn_type] is the fp32 [2F*N] vector lambda_i * mask_i on rows of frame i
    of the conditional half (and of the unconditional half when modulate_uc), zero elsewhere."""
    if mod is None:
        return None
    is_mod, is_inj, mp = mod
    inject = mp["injected_features_group"] if is_inj else None
    rowadd = None
    if is_mod and layer_type in mp["modulate_layer_type"]:
        frames = list(range(mp["num_frames"]))
        mp["modulate_layer_frames_group"] = mp["modulate_layer_frames"].get(layer_type, frames)
        masks = mp["feature_masks"]
        Fm = len(masks)
        ra = torch.zeros((2 * Fm, N), dtype=torch.float32, device=device)
        for i, mask in enumerate(masks):
            if i in mp["modulate_block_frames_group"] and i in mp["modulate_layer_frames_group"] and \
                    i in mp["modulate_timestep_frames_group"]:
                lam = get_modulate_lambda(mp["modulate_lambda_start"], mp["modulate_lambda_end"], mp["modulate_schedule"],
                                          total_steps=mp["num_frames"], current_step=i)
                row = (lam * mask.to(device=device, dtype=torch.float64)).to(torch.float32)
                ra[i + Fm] = row
                if mp["modulate_uc"]:
                    ra[i] = row
        ra = ra.reshape(-1).contiguous()
        rowadd = {t: ra for t in ("self_attn", "cross_attn", "ff_out") if t in mp["modulate_attn_type"]}
    if inject is None and rowadd is None:
        return None
    return inject, rowadd


def get_modulate_lambda(modulate_lambda_start, modulate_lambda_end, modulate_schedule, total_steps, current_step):
    """sgm/modules/diffusionmodules/util.py:383-392."""
    assert modulate_schedule in ["constant", "linear"]
    if modulate_schedule == "constant":
        return modulate_lambda_start
    return modulate_lambda_start + (modulate_lambda_end - modulate_lambda_start) * current_step / total_steps


class TimestepEmbedSequential(nn.Sequential):
    """openaimodel.py:67-114: children dispatched by type."""

    def run(self, x, x_skip, emb_all, context, mod=None, skip_resample=False, res_cache=None):
        """res_cache: the fork point of a Step-4 sweep (pipeline.modulation_sweep, `shared_prefix`): receives the ResBlock's output on the
        pass that computes it and hands it back on the later ones, which skip the ResBlock."""
        for layer in self:
            if isinstance(layer, ResBlock):
                if res_cache is not None and "x" in res_cache:
                    x = res_cache["x"]
                else:
                    x = layer.run(x, x_skip, emb_all)
                    if res_cache is not None:
                        res_cache["x"] = x
                x_skip = None
            elif isinstance(layer, SpatialTransformer):
                x = layer.run(x, context, mod)
            elif isinstance(layer, (Upsample, Downsample)):
                if skip_resample:                               # taps-only evaluation: nothing downstream reads this
                    continue
                x = layer.run(x)
            else:
                raise VidsegError(f"unexpected layer {type(layer)}")
        return x


class _ConvIn(nn.Conv2d):
    pass


class UNetModel(nn.Module):
    """sgm/modules/diffusionmodules/openaimodel.py:487-954 for the SD 2.1 configuration."""
    HOST_MASTERS = True          # engine.DiffusionEngine._apply leaves these modules alone (.to / .cuda / .half are no-ops)

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0.0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False, num_heads=-1,
                 num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 transformer_depth=1, context_dim=None, disable_self_attentions=None, num_attention_blocks=None,
                 disable_middle_self_attn=False, disable_middle_transformer=False, use_linear_in_transformer=False,
                 spatial_transformer_attn_type="softmax", adm_in_channels=None):
        super().__init__()
        unsupported = dict(dims=(dims, 2), use_scale_shift_norm=(use_scale_shift_norm, False), resblock_updown=(resblock_updown, False),
                           conv_resample=(conv_resample, True), disable_self_attentions=(disable_self_attentions, None),
                           num_attention_blocks=(num_attention_blocks, None), disable_middle_self_attn=(disable_middle_self_attn, False),
                           disable_middle_transformer=(disable_middle_transformer, False),
                           use_linear_in_transformer=(use_linear_in_transformer, True))
        for k, (v, want) in unsupported.items():
            if v != want:
                raise NotImplementedError(f"UNetModel({k}={v!r}) is not on the SD 2.1 / SVD path (expects {want!r})")
        if num_head_channels != 64:
            raise NotImplementedError("num_head_channels must be 64")
        if context_dim is None:
            raise NotImplementedError("context_dim is required (cross-attention UNet)")
        if model_channels % 64 != 0:
            raise NotImplementedError("model_channels must be a multiple of 64 (MFMA K tiling)")
        if isinstance(num_res_blocks, int):
            num_res_blocks = len(channel_mult) * [num_res_blocks]
        if isinstance(transformer_depth, int):
            transformer_depth = len(channel_mult) * [transformer_depth]
        transformer_depth_middle = transformer_depth[-1]
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions = num_res_blocks, list(attention_resolutions)
        self.channel_mult, self.num_classes, self.context_dim = list(channel_mult), num_classes, context_dim
        ted = model_channels * 4
        self.time_embed = nn.Sequential(_meta(nn.Linear, model_channels, ted), nn.SiLU(), _meta(nn.Linear, ted, ted))
        if num_classes is not None:
            if num_classes != "sequential":
                raise NotImplementedError("only num_classes='sequential' (SVD) is supported")
            self.label_emb = nn.Sequential(nn.Sequential(_meta(nn.Linear, adm_in_channels, ted), nn.SiLU(), _meta(nn.Linear, ted, ted)))

        def attn(ch, depth):
            return SpatialTransformer(ch, ch // num_head_channels, num_head_channels, depth=depth, context_dim=context_dim)

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(_meta(_ConvIn, in_channels, model_channels, 3, padding=1))])
        chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks[level]):
                layers = [ResBlock(ch, ted, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(attn(ch, transformer_depth[level]))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(ResBlock(ch, ted, ch), attn(ch, transformer_depth_middle), ResBlock(ch, ted, ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks[level] + 1):
                ich = chans.pop()
                layers = [ResBlock(ch + ich, ted, model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(attn(ch, transformer_depth[level]))
                if level and i == num_res_blocks[level]:
                    layers.append(Upsample(ch, ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(_meta(GroupNorm32, 32, ch), nn.SiLU(), _meta(nn.Conv2d, model_channels, out_channels, 3, padding=1))
        self._packed_on = None
        self.tap_mode = "output"                                    # "output" (reference dumps), "all", "none"
        self.precision = "fp16"                                     # "fp16" (16-bit activations, this file) or "exact" (exact.py)
        self._exact = None

    # ------------------------------------------------------------------ parameters
    def load_state_dict(self, state_dict, strict=True, assign=True):
        r = super().load_state_dict(state_dict, strict=strict, assign=True)
        self._packed_on = None
        self._exact = None
        return r

    def _resblocks(self):
        return [m for m in self.modules() if isinstance(m, ResBlock)]

    def set_precision(self, mode):
        """"fp16": the 16-bit path of this file (activations stored in the library's 16-bit format; taps ~1.2e-3 from an fp32
        evaluation).  "exact": exact.ExactRunner -- fp32 activations, every conv / linear on the same MFMA kernels over split
        (hi, lo) operands, taps ~1e-5 from fp32: what best-of-10 K-means++ needs to return the reference's masks (3x the MFMA
        work).

        MEMORY: the exact runner packs its own weight images ([w_hi | w_hi | w_lo]: 3x the 16-bit weights, 5.2 GB for SD 2.1, 9.1 GB
        for SVD) on first use and KEEPS them when the mode is switched back to "fp16" (bench.py and the tests flip modes per window);
        a caller that is done with the exact mode calls `release_exact()` to free them."""
        if mode not in ("fp16", "exact"):
            raise ValueError(f"unknown precision {mode!r}")
        self.precision = mode                                       # the exact runner (its split weight images) stays cached: release_exact()

    def release_exact(self):
        """Drop the exact mode's weight images (3x the 16-bit ones) and its per-window caches; the next exact forward packs them again."""
        self._exact = None

    def stash_resblock_features(self, on=True):
        """Make every ResBlock leave `in_layers_features` / `out_layers_features` (openaimodel.py:349-350, 367-368) after a forward.
        Off by default: nothing on the path reads them and they cost two fp16 stores per ResBlock."""
        for rb in self._resblocks():
            rb.stash_features = bool(on)
            if not on:
                rb.in_layers_features = rb.out_layers_features = None

    def pack(self, device):
        """One-time weight packing into the layouts the kernels read (bf16 K-contiguous matrices)."""
        device = torch.device(device)
        for p in self.parameters():
            if p.is_meta:
                raise VidsegError("UNetModel has no weights: call load_state_dict() first")
        for m in self.modules():
            if m is not self and hasattr(m, "pack") and not isinstance(m, (BasicTransformerBlock, CrossAttention, FeedForward)):
                m.pack(device)
        te = self.time_embed
        self.te_w1, self.te_b1 = ops.pack_linear(te[0].weight, device), ops.f32(te[0].bias, device)
        self.te_w2, self.te_b2 = ops.pack_linear(te[2].weight, device), ops.f32(te[2].bias, device)
        if self.num_classes is not None:
            le = self.label_emb[0]
            self.le_w1, self.le_b1 = ops.pack_linear(le[0].weight, device), ops.f32(le[0].bias, device)
            self.le_w2, self.le_b2 = ops.pack_linear(le[2].weight, device), ops.f32(le[2].bias, device)
        rbs = self._resblocks()
        off = 0
        for rb in rbs:
            rb.emb_offset = off
            off += rb.out_channels
        self.emb_w = ops.pack_linear(torch.cat([rb.emb_layers[1].weight for rb in rbs], 0), device)
        self.emb_b = ops.f32(torch.cat([rb.emb_layers[1].bias for rb in rbs], 0), device)
        cin = self.input_blocks[0][0]
        self.cin_w, self.cin_b = ops.pack_conv_in(cin.weight, device), ops.f32(cin.bias, device)
        self.out_g, self.out_beta = ops.f32(self.out[0].weight, device), ops.f32(self.out[0].bias, device)
        self.out_w, self.out_b = ops.pack_conv_out(self.out[2].weight, device), ops.f32(self.out[2].bias, device)
        self._set_taps()
        self._packed_on = device

    def _set_taps(self):
        for name, blocks in (("input", self.input_blocks), ("middle", [self.middle_block]), ("output", self.output_blocks)):
            for blk in blocks:
                for layer in blk:
                    if isinstance(layer, SpatialTransformer):
                        layer.tap = self.tap_mode == "all" or (self.tap_mode == "output" and name == "output")

    # ------------------------------------------------------------------ forward
    def embed(self, timesteps, y=None):
        t_emb = ops.timestep_embedding(timesteps.float(), self.model_channels)                         # DU:209-233
        emb = ops.linear(ops.linear(t_emb, self.te_w1, self.te_b1, act=ops.ACT_SILU), self.te_w2, self.te_b2)
        if self.num_classes is not None:
            yb = y if y.dtype == ops.act_dtype() else ops.to_bf16(y.float().contiguous())
            emb = ops.linear(ops.linear(yb, self.le_w1, self.le_b1, act=ops.ACT_SILU), self.le_w2, self.le_b2, residual=emb)
        return emb

    def _block_mod(self, kind, i, blk, is_modulate_step, is_injected_step, mp, device):
        """Per-block flags of openaimodel.py:884-937 (input blocks: injection only; output blocks: both)."""
        if not (is_modulate_step or is_injected_step):
            return None
        has_st = len(blk) > 1 and "SpatialTransformer" in str(type(blk[1]))
        is_mod = False
        if kind == "output" and is_modulate_step and i in mp["modulate_block_idx"] and has_st:
            is_mod = True
            mp["modulate_block_frames_group"] = mp["modulate_block_frames"].get(i, list(range(mp["num_frames"])))
        is_inj = False
        idx_key = "input_block_indices" if kind == "input" else "output_block_indices"
        if is_injected_step and has_st and kind in mp["injected_block_types"] and i in mp[idx_key]:
            from .util import load_target_features
            mp["injected_features_group"] = load_target_features(mp["feature_folder"], mp["exp_name"], mp["timestep"], kind,
                                                                 mp["injected_feature_types"], i, device)
            is_inj = len(mp["injected_features_group"]) > 0
        return (is_mod, is_inj, mp) if (is_mod or is_inj) else None

    def forward_nhwc(self, x_nhwc_f32, timesteps, context_bf16, y=None, is_modulate_step=False, is_injected_step=False,
                     modulate_params=None, stop_after_block=None):
        """x: fp32 NHWC [B, h, w, Cin]; context: bf16 [B, L, ctx]; returns fp32 NCHW [B, Cout, h, w].
        stop_after_block=b: taps-only evaluation -- output blocks 0..b run (b without its Upsample), their Q/K taps are left on
        the attention modules, and None is returned (pipeline.feature_pass(masks_only=True))."""
        if self._packed_on is None:
            self.pack(x_nhwc_f32.device)
        dev = x_nhwc_f32.device
        emb = self.embed(timesteps, y)
        emb_all = ops.linear(ops.silu(emb), self.emb_w, self.emb_b, out_f32=True)                       # every ResBlock's emb_layers
        from .util import shared_prefix
        pre = shared_prefix(modulate_params, is_modulate_step)      # Step-4 sweep: the first evaluation's shared prefix (see exact.ExactRunner.forward)
        resume = pre is not None and pre.get("state") is not None
        if resume:
            hs, h = list(pre["state"][0]), None
        else:
            h = ops.conv_in(x_nhwc_f32, self.cin_w, self.cin_b)
            hs = [h]
            for i, blk in list(enumerate(self.input_blocks))[1:]:
                h = blk.run(h, None, emb_all, context_bf16, self._block_mod("input", i, blk, False, is_injected_step, modulate_params, dev))
                hs.append(h)
            h = self.middle_block.run(h, None, emb_all, context_bf16)
        for i, blk in enumerate(self.output_blocks):
            if resume and i < pre["fork"]:
                continue
            mod = self._block_mod("output", i, blk, is_modulate_step, is_injected_step, modulate_params, dev)
            if stop_after_block is not None and i == stop_after_block:
                blk.run(h, hs.pop(), emb_all, context_bf16, mod, skip_resample=True)
                return None
            if pre is not None and i == pre["fork"]:
                if resume:
                    h = blk.run(None, None, emb_all, context_bf16, mod, res_cache={"x": pre["state"][1]})
                else:
                    rc = {}
                    h = blk.run(h, hs.pop(), emb_all, context_bf16, mod, res_cache=rc)
                    pre["state"] = (tuple(hs), rc["x"])
                continue
            h = blk.run(h, hs.pop(), emb_all, context_bf16, mod)                                         # OAI:911-948
        h = ops.groupnorm(h, self.out_g, self.out_beta, eps=1e-5, silu=True)
        return ops.conv_out4(h, self.out_w, self.out_b)

    def forward(self, x, timesteps=None, context=None, y=None, is_modulate_step=False, is_injected_step=False,
                modulate_params=None, stop_after_block=None, **kwargs):
        """Reference signature (openaimodel.py:831-841): x NCHW."""
        if (is_modulate_step or is_injected_step) and modulate_params is None:
            raise AssertionError("modulate_params is required for a modulated / injected step")
        if (y is not None) != (self.num_classes is not None):
            raise AssertionError("must specify y if and only if the model is class-conditional")
        if not x.is_cuda:
            raise VidsegError("UNetModel runs on a HIP device only (no CPU fallback)")
        if self.precision == "exact":
            if any(rb.stash_features for rb in self._resblocks()):
                raise NotImplementedError("exact precision does not produce ResBlock.in_layers_features / out_layers_features: "
                                          "stash_resblock_features(False) or set_precision('fp16')")
            if self._exact is None:
                from .exact import ExactRunner
                for p in self.parameters():
                    if p.is_meta:
                        raise VidsegError("UNetModel has no weights: call load_state_dict() first")
                self._set_taps()
                self._exact = ExactRunner(self, x.device)
            return self._exact.forward(x, timesteps, context, stop_after_block=stop_after_block, is_modulate_step=is_modulate_step,
                                       is_injected_step=is_injected_step, modulate_params=modulate_params)
        xn = x.float().permute(0, 2, 3, 1).contiguous()
        ctx = context if context.dtype == ops.act_dtype() else \
            ops.window_cached(self, "_ctx16", (context,), lambda: ops.to_bf16(context.float().contiguous()))
        return self.forward_nhwc(xn, timesteps, ctx, y, is_modulate_step, is_injected_step, modulate_params, stop_after_block)
