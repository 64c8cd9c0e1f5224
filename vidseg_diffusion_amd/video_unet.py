"""MI355X-native SVD `VideoUNet` behind the reference's plug-in seam (configs/inference/svd.yaml:14-34).

Mirrors sgm/modules/diffusionmodules/video_model.py (VideoResBlock :15-89, VideoUNet :92-566) and
sgm/modules/video_attention.py (VideoTransformerBlock :18-285, SpatialVideoTransformer :291-489): same
constructor kwargs, same 1428 state-dict keys, and the attribute protocol of the SVD driver
(scripts/sampling/svd_pipeline_vspw.py:111-119):

    "SpatialVideoTransformer" in str(type(block[1]))
    block[1].transformer_blocks[0].attn{1,2}.{q,k}   block[1].time_stack[0].attn{1,2}.{q,k}

Layout decision: tokens never leave the spatial order ``(b t) s c``.  Every per-token op of the temporal
branch (LayerNorm, GEGLU feed-forwards, q/k/v/out projections) is order-independent, so only the T x T
attention itself needs to know about time: a dedicated kernel strides over frames in place, the cross-attention
to the first frame's context is an ordinary cross-attention with batch = sample and T*S queries, and the
3-tap temporal convolution is the implicit-GEMM kernel with its taps running over frames.  No
``(b t) s c <-> (b s) t c`` transposes are materialised; only the fp16 dumps are written in the reference's
``[(b s), t, c]`` layout (by a row permutation in the GEMM epilogue).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from ._lib import VidsegError
from .unet import (BasicTransformerBlock, CrossAttention, Downsample, FeedForward, GroupNorm32, ResBlock, SpatialTransformer,
                   TimestepEmbedSequential, UNetModel, Upsample, _ConvIn, _meta, block_modulation)

F16 = torch.float16


class AlphaBlender(nn.Module):
    """diffusionmodules/util.py:314-380, strategy 'learned_with_images' (image_only_indicator is all zeros on
    the path, svd_pipeline_vspw.py:307-311) -> alpha = sigmoid(mix_factor)."""

    def __init__(self, alpha: float = 0.5, merge_strategy: str = "learned_with_images"):
        super().__init__()
        if merge_strategy not in ("learned", "learned_with_images"):
            raise NotImplementedError(f"merge_strategy {merge_strategy!r}")
        self.merge_strategy = merge_strategy
        self.mix_factor = nn.Parameter(torch.empty(1, device="meta"))

    def pack(self, dev):
        self.mix = ops.f32(self.mix_factor, dev)

    def run(self, x_spatial, x_temporal):
        return ops.alpha_blend(x_spatial, x_temporal, self.mix)


class _TimeStack(nn.Module):
    """The dims=3 ResBlock of VideoResBlock.time_stack (kernel [3,1,1], exchange_temb_dims)."""

    def __init__(self, channels, emb_channels):
        super().__init__()
        self.out_channels = channels
        self.in_layers = nn.Sequential(_meta(GroupNorm32, 32, channels), nn.SiLU(),
                                       _meta(nn.Conv3d, channels, channels, (3, 1, 1), padding=(1, 0, 0)))
        self.emb_layers = nn.Sequential(nn.SiLU(), _meta(nn.Linear, emb_channels, channels))
        self.out_layers = nn.Sequential(_meta(GroupNorm32, 32, channels), nn.SiLU(), nn.Dropout(p=0.0),
                                        _meta(nn.Conv3d, channels, channels, (3, 1, 1), padding=(1, 0, 0)))
        self.skip_connection = nn.Identity()
        self.emb_offset = None

    def pack(self, dev):
        self.g1, self.b1 = ops.f32(self.in_layers[0].weight, dev), ops.f32(self.in_layers[0].bias, dev)
        self.w1, self.cb1 = ops.pack_conv_temporal3(self.in_layers[2].weight, dev), ops.f32(self.in_layers[2].bias, dev)
        self.g2, self.b2 = ops.f32(self.out_layers[0].weight, dev), ops.f32(self.out_layers[0].bias, dev)
        self.w2, self.cb2 = ops.pack_conv_temporal3(self.out_layers[3].weight, dev), ops.f32(self.out_layers[3].bias, dev)

    def run(self, x, emb_all, T):
        BT, H, W, C = x.shape
        xv = x.view(BT // T, T * H, W, C)                         # GroupNorm over (c/32, t, h, w) per sample b
        h = ops.groupnorm(xv, self.g1, self.b1, eps=1e-5, silu=True).view(BT, H, W, C)
        rv = emb_all[:, self.emb_offset:self.emb_offset + C]       # per (b t) vector == emb "(b t) c -> b c t"
        h = ops.conv_temporal3(h, self.w1, self.cb1, T, rowvec=rv)
        h = ops.groupnorm(h.view(BT // T, T * H, W, C), self.g2, self.b2, eps=1e-5, silu=True).view(BT, H, W, C)
        return ops.conv_temporal3(h, self.w2, self.cb2, T, residual=x)


class VideoResBlock(ResBlock):
    """video_model.py:15-89."""

    def __init__(self, channels, emb_channels, out_channels=None, merge_strategy="learned_with_images", merge_factor=0.5):
        super().__init__(channels, emb_channels, out_channels)
        self.time_stack = _TimeStack(self.out_channels, emb_channels)
        self.time_mixer = AlphaBlender(merge_factor, merge_strategy)
        self.video_features = None

    def pack(self, dev):
        super().pack(dev)
        self.time_stack.pack(dev)
        self.time_mixer.pack(dev)

    def run(self, x0, x1, emb_all, T=None):
        x = super().run(x0, x1, emb_all)                           # VM:74
        xt = self.time_stack.run(x, emb_all, T)                    # VM:79-81
        out = self.time_mixer.run(x, xt)                           # VM:82-86
        self.video_features = out
        return out


class VideoTransformerBlock(nn.Module):
    """video_attention.py:18-285 with ff_in=True, inner_dim == dim (is_res), cross-attention to the first-frame context."""

    def __init__(self, dim, n_heads, d_head, context_dim=None):
        super().__init__()
        self.norm_in = _meta(nn.LayerNorm, dim)
        self.ff_in = FeedForward(dim)
        self.attn1 = CrossAttention(dim, None, n_heads, d_head)
        self.ff = FeedForward(dim)
        self.norm2 = _meta(nn.LayerNorm, dim)
        self.attn2 = CrossAttention(dim, context_dim, n_heads, d_head)
        self.norm1 = _meta(nn.LayerNorm, dim)
        self.norm3 = _meta(nn.LayerNorm, dim)
        self.heads = n_heads

    def pack(self, dev):
        for m in (self.ff_in, self.attn1, self.ff, self.attn2):
            m.pack(dev)
        self.ln = {n: (ops.f32(getattr(self, n).weight, dev), ops.f32(getattr(self, n).bias, dev))
                   for n in ("norm_in", "norm1", "norm2", "norm3")}

    def run(self, x, time_context, T, tap, mod=None):
        """x: bf16 [(b t), S, C] in spatial order; time_context: bf16 [b, L, ctx] (first frame of each sample).
        mod: None or (inject, rowadd) from block_modulation(..., "temporal"): injected temporal_self_attn_{q,k,v} dumps
        replace the projections of attn1 (VA:166-195) and rowadd[attn_type] = lambda_i * mask_i on the rows of frame i is
        added to attn1_out / attn2_out / ff_out (VA:197-216, 231-250, 258-277).  In the reference's [(b s), t, c] layout
        that is out[half_hw:, i] += lambda*mask[:, None]; here rows are (b t) s, the same vector the spatial blocks use."""
        BT, S, C = x.shape
        Bv = BT // T
        dev = x.device
        inj, ra = mod if mod is not None else (None, None)
        inj, ra = inj or {}, ra or {}

        def pick(sub):
            for k, v in inj.items():
                if sub in k:
                    return v
            return None

        def to_spatial(t16):                                        # fp16 [(b s), t, c] dump -> bf16 [(b t), s, c]
            return ops.f16_to_bf16(t16.view(Bv, S, T, C).permute(0, 2, 1, 3).contiguous().view(BT, S, C))

        x = self.ff_in.run(ops.layernorm(x, *self.ln["norm_in"]), x)                       # VA:155-159
        n1 = ops.layernorm(x, *self.ln["norm1"])
        a1 = self.attn1
        tq = torch.empty((Bv * S, T, C), dtype=F16, device=dev) if tap else None
        tk = torch.empty((Bv * S, T, C), dtype=F16, device=dev) if tap else None
        qkv = ops.linear_temporal_tap(n1, a1.w_qkv, T, S, tq, tk, C)
        iq, ik, iv = pick("temporal_self_attn_q"), pick("temporal_self_attn_k"), pick("temporal_self_attn_v")
        q = to_spatial(iq) if iq is not None else qkv[..., :C]
        k = to_spatial(ik) if ik is not None else qkv[..., C:2 * C]
        v = to_spatial(iv) if iv is not None else qkv[..., 2 * C:]
        att = ops.temporal_attention(q, k, v, self.heads, Bv, T, S)                          # VA:166-195
        x = ops.linear(att, a1.w_o, a1.b_o, residual=x, rowadd=ra.get("self_attn"))         # VA:197-218
        a2 = self.attn2
        L = time_context.shape[1]
        n2 = ops.layernorm(x, *self.ln["norm2"])
        tq2 = torch.empty((Bv * S, T, C), dtype=F16, device=dev) if tap else None
        q2 = ops.linear_temporal_tap(n2, a2.w_q, T, S, tq2, None, C)
        tk2 = torch.empty((Bv, L, C), dtype=F16, device=dev) if tap else None
        kv = ops.linear(time_context, a2.w_kv, tap=tk2, tap_cols=C)
        att2 = ops.attention(q2.view(Bv, T * S, C), kv[..., :C], kv[..., C:], self.heads)   # VA:224-250
        x = ops.linear(att2.view(BT, S, C), a2.w_o, a2.b_o, residual=x, rowadd=ra.get("cross_attn"))
        if tap:
            a1.q, a1.k = tq, tk
            a2.q = tq2
            a2.k = tk2[:, None].expand(Bv, S, L, C).reshape(Bv * S, L, C)                   # reference shape [(b s), L, C]
        if iq is not None:
            a1.q = iq                                                                       # ATT:330-331 stores what was used
        if ik is not None:
            a1.k = ik
        return self.ff.run(ops.layernorm(x, *self.ln["norm3"]), x, ra.get("ff_out"))        # VA:252-281


class SpatialVideoTransformer(SpatialTransformer):
    """video_attention.py:291-489 with use_linear, use_spatial_context, ff_in, depth-matched time_stack."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, context_dim=None, merge_strategy="learned_with_images",
                 merge_factor=0.5, max_time_embed_period=10000):
        super().__init__(in_channels, n_heads, d_head, depth=depth, context_dim=context_dim)
        inner = n_heads * d_head
        self.time_stack = nn.ModuleList([VideoTransformerBlock(inner, n_heads, d_head, context_dim) for _ in range(depth)])
        ted = in_channels * 4
        self.time_pos_embed = nn.Sequential(_meta(nn.Linear, in_channels, ted), nn.SiLU(), _meta(nn.Linear, ted, in_channels))
        self.time_mixer = AlphaBlender(merge_factor, merge_strategy)
        self.max_time_embed_period = max_time_embed_period
        self.features_after_temporal = None

    def pack(self, dev):
        super().pack(dev)
        for blk in self.time_stack:
            blk.pack(dev)
        self.tp_w1, self.tp_b1 = ops.pack_linear(self.time_pos_embed[0].weight, dev), ops.f32(self.time_pos_embed[0].bias, dev)
        self.tp_w2, self.tp_b2 = ops.pack_linear(self.time_pos_embed[2].weight, dev), ops.f32(self.time_pos_embed[2].bias, dev)
        self.time_mixer.pack(dev)
        self._temb = {}

    def _frame_emb(self, T, dev):
        """time_pos_embed(timestep_embedding(arange(T))) -- depends on weights and T only (VA:417-427)."""
        if T not in self._temb:
            fr = torch.arange(T, dtype=torch.float32, device=dev)
            te = ops.timestep_embedding(fr, self.in_channels, self.max_time_embed_period)
            self._temb[T] = ops.linear(ops.linear(te, self.tp_w1, self.tp_b1, act=ops.ACT_SILU), self.tp_w2, self.tp_b2)
        return self._temb[T]

    def run(self, x, context, T=None, mod=None):
        B, H, W, C = x.shape
        S = H * W
        t = ops.groupnorm(x, self.g, self.b, eps=1e-6, silu=False).view(B, S, C)
        t = ops.linear(t, self.w_in, self.b_in)
        time_context = context[::T].contiguous()                                            # VA:400-404
        emb = self._frame_emb(T, x.device)
        for i, (blk, mix) in enumerate(zip(self.transformer_blocks, self.time_stack)):
            # VA:432-451 / 456-472: the spatial and the temporal block each get their own layer-frames group
            t = blk.run(t, context, self.tap and i == 0, block_modulation(mod, "spatial", S, x.device))
            tm = mix.run(ops.add_rowvec(t, emb, S), time_context, T, self.tap and i == 0,
                         block_modulation(mod, "temporal", S, x.device))                    # VA:429-431, 453-476
            t = self.time_mixer.run(t, tm)                                                  # VA:463-467
        out = ops.linear(t, self.w_out, self.b_out, residual=x.view(B, S, C))
        self.features_after_temporal = out
        return out.view(B, H, W, C)


class VideoTimestepEmbedSequential(TimestepEmbedSequential):
    def run(self, x, x_skip, emb_all, context, T=None, mod=None, skip_resample=False, res_cache=None):
        for layer in self:
            if isinstance(layer, VideoResBlock):
                if res_cache is not None and "x" in res_cache:                  # Step-4 sweep fork point (unet.TimestepEmbedSequential.run)
                    x = res_cache["x"]
                else:
                    x = layer.run(x, x_skip, emb_all, T)
                    if res_cache is not None:
                        res_cache["x"] = x
                x_skip = None
            elif isinstance(layer, SpatialVideoTransformer):
                x = layer.run(x, context, T, mod)
            elif isinstance(layer, (Upsample, Downsample)):
                if skip_resample:
                    continue
                x = layer.run(x)
            else:
                raise VidsegError(f"unexpected layer {type(layer)}")
        return x


class VideoUNet(UNetModel):
    """sgm/modules/diffusionmodules/video_model.py:92-566 for the SVD configuration."""

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0.0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False, num_heads=-1,
                 num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 transformer_depth=1, transformer_depth_middle=None, context_dim=None, time_downup=False, time_context_dim=None,
                 extra_ff_mix_layer=False, use_spatial_context=False, merge_strategy="fixed", merge_factor=0.5,
                 spatial_transformer_attn_type="softmax", video_kernel_size=3, use_linear_in_transformer=False,
                 adm_in_channels=None, disable_temporal_crossattention=False, max_ddpm_temb_period=10000):
        nn.Module.__init__(self)
        unsupported = dict(dims=(dims, 2), use_scale_shift_norm=(use_scale_shift_norm, False), resblock_updown=(resblock_updown, False),
                           conv_resample=(conv_resample, True), time_downup=(time_downup, False),
                           extra_ff_mix_layer=(extra_ff_mix_layer, True), use_spatial_context=(use_spatial_context, True),
                           use_linear_in_transformer=(use_linear_in_transformer, True),
                           disable_temporal_crossattention=(disable_temporal_crossattention, False),
                           video_kernel_size=(list(video_kernel_size) if not isinstance(video_kernel_size, int) else video_kernel_size, [3, 1, 1]))
        for k, (v, want) in unsupported.items():
            if v != want:
                raise NotImplementedError(f"VideoUNet({k}={v!r}) is not on the SVD path (expects {want!r})")
        if num_head_channels != 64 or context_dim is None or model_channels % 64 != 0:
            raise NotImplementedError("VideoUNet needs num_head_channels=64, a context_dim and model_channels % 64 == 0")
        if num_classes != "sequential":
            raise NotImplementedError("VideoUNet: only num_classes='sequential' (svd.yaml)")
        if isinstance(num_res_blocks, int):
            num_res_blocks = len(channel_mult) * [num_res_blocks]
        if isinstance(transformer_depth, int):
            transformer_depth = len(channel_mult) * [transformer_depth]
        tdm = transformer_depth[-1] if transformer_depth_middle is None else transformer_depth_middle
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions = num_res_blocks, list(attention_resolutions)
        self.channel_mult, self.num_classes, self.context_dim = list(channel_mult), num_classes, context_dim
        ted = model_channels * 4
        self.time_embed = nn.Sequential(_meta(nn.Linear, model_channels, ted), nn.SiLU(), _meta(nn.Linear, ted, ted))
        self.label_emb = nn.Sequential(nn.Sequential(_meta(nn.Linear, adm_in_channels, ted), nn.SiLU(), _meta(nn.Linear, ted, ted)))

        def attn(ch, depth):
            return SpatialVideoTransformer(ch, ch // 64, 64, depth=depth, context_dim=context_dim, merge_strategy=merge_strategy,
                                           merge_factor=merge_factor, max_time_embed_period=max_ddpm_temb_period)

        def res(ch, out_ch):
            return VideoResBlock(ch, ted, out_ch, merge_strategy=merge_strategy, merge_factor=merge_factor)

        Seq = VideoTimestepEmbedSequential
        self.input_blocks = nn.ModuleList([Seq(_meta(_ConvIn, in_channels, model_channels, 3, padding=1))])
        chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks[level]):
                layers = [res(ch, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(attn(ch, transformer_depth[level]))
                self.input_blocks.append(Seq(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                ds *= 2
                self.input_blocks.append(Seq(Downsample(ch, ch)))
                chans.append(ch)
        self.middle_block = Seq(res(ch, ch), attn(ch, tdm), res(ch, ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks[level] + 1):
                ich = chans.pop()
                layers = [res(ch + ich, model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(attn(ch, transformer_depth[level]))
                if level and i == num_res_blocks[level]:
                    ds //= 2
                    layers.append(Upsample(ch, ch))
                self.output_blocks.append(Seq(*layers))
        self.out = nn.Sequential(_meta(GroupNorm32, 32, ch), nn.SiLU(), _meta(nn.Conv2d, model_channels, out_channels, 3, padding=1))
        self._packed_on = None
        self.tap_mode = "output"
        self.precision = "fp16"
        self._exact = None

    def _resblocks(self):
        # every emb_layers user, spatial and temporal, shares one batched GEMM
        return [m for m in self.modules() if isinstance(m, (ResBlock, _TimeStack))]

    def _block_mod(self, kind, i, blk, is_modulate_step, is_injected_step, mp, device):
        """video_model.py:480-545: as UNetModel._block_mod, keyed on SpatialVideoTransformer."""
        if not (is_modulate_step or is_injected_step):
            return None
        has_st = len(blk) > 1 and "SpatialVideoTransformer" in str(type(blk[1]))
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

    def forward_nhwc(self, x_nhwc_f32, timesteps, context_bf16, y=None, num_video_frames=None, is_modulate_step=False,
                     is_injected_step=False, modulate_params=None, stop_after_block=None):
        if num_video_frames is None:
            raise VidsegError("VideoUNet needs num_video_frames")
        T = int(num_video_frames)
        if self._packed_on is None:
            self.pack(x_nhwc_f32.device)
        emb = self.embed(timesteps, y)
        emb_all = ops.linear(ops.silu(emb), self.emb_w, self.emb_b, out_f32=True)
        dev = x_nhwc_f32.device
        from .util import shared_prefix
        pre = shared_prefix(modulate_params, is_modulate_step)      # Step-4 sweep: the first evaluation's shared prefix (see exact.ExactRunner.forward)
        resume = pre is not None and pre.get("state") is not None
        if resume:
            hs, h = list(pre["state"][0]), None
        else:
            h = ops.conv_in(x_nhwc_f32, self.cin_w, self.cin_b)
            hs = [h]
            for i, blk in list(enumerate(self.input_blocks))[1:]:
                h = blk.run(h, None, emb_all, context_bf16, T,
                            self._block_mod("input", i, blk, False, is_injected_step, modulate_params, dev))   # VM:480-510
                hs.append(h)
            h = self.middle_block.run(h, None, emb_all, context_bf16, T)
        for i, blk in enumerate(self.output_blocks):
            if resume and i < pre["fork"]:
                continue
            mod = self._block_mod("output", i, blk, is_modulate_step, is_injected_step, modulate_params, dev)
            if stop_after_block is not None and i == stop_after_block:                                    # taps-only evaluation
                blk.run(h, hs.pop(), emb_all, context_bf16, T, mod, skip_resample=True)
                return None
            if pre is not None and i == pre["fork"]:
                if resume:
                    h = blk.run(None, None, emb_all, context_bf16, T, mod, res_cache={"x": pre["state"][1]})
                else:
                    rc = {}
                    h = blk.run(h, hs.pop(), emb_all, context_bf16, T, mod, res_cache=rc)
                    pre["state"] = (tuple(hs), rc["x"])
                continue
            h = blk.run(h, hs.pop(), emb_all, context_bf16, T, mod)                                       # VM:521-562
        h = ops.groupnorm(h, self.out_g, self.out_beta, eps=1e-5, silu=True)
        return ops.conv_out4(h, self.out_w, self.out_b)

    def forward(self, x, timesteps=None, context=None, y=None, time_context=None, num_video_frames=None, image_only_indicator=None,
                is_modulate_step=False, is_injected_step=False, modulate_params=None, stop_after_block=None, **kwargs):
        """Reference signature (video_model.py:451-463)."""
        if (is_modulate_step or is_injected_step) and modulate_params is None:
            raise AssertionError("modulate_params is required for a modulated / injected step")
        if y is None:
            raise AssertionError("must specify y if and only if the model is class-conditional")
        if image_only_indicator is not None and bool(torch.as_tensor(image_only_indicator).any()):
            raise NotImplementedError("image_only_indicator != 0 (image batches) is not on the path")
        if not x.is_cuda:
            raise VidsegError("VideoUNet runs on a HIP device only (no CPU fallback)")
        if self.precision == "exact":                                 # UNetModel.set_precision: exact.ExactRunner (fp32-accurate, 3x MFMA work)
            if self._exact is None:
                from .exact import ExactRunner
                for p in self.parameters():
                    if p.is_meta:
                        raise VidsegError("VideoUNet has no weights: call load_state_dict() first")
                self._set_taps()
                self._exact = ExactRunner(self, x.device)
            return self._exact.forward(x, timesteps, context, y=y, num_video_frames=num_video_frames, stop_after_block=stop_after_block,
                                       is_modulate_step=is_modulate_step, is_injected_step=is_injected_step, modulate_params=modulate_params)
        xn = x.float().permute(0, 2, 3, 1).contiguous()
        ctx = context if context.dtype == ops.act_dtype() else ops.to_bf16(context.float().contiguous())
        return self.forward_nhwc(xn, timesteps, ctx, y, num_video_frames, is_modulate_step, is_injected_step, modulate_params,
                                 stop_after_block)
