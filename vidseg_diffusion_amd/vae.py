"""First stage: `AutoencoderKL.encode` and `.decode` of the reference on MI355X.

    sgm/models/autoencoder.py:437-522   AutoencodingEngineLegacy / AutoencoderKL: encoder -> quant_conv -> regularizer
    sgm/modules/diffusionmodules/model.py:84-91 (Downsample), 94-151 (ResnetBlock), 161-202 (AttnBlock),
                                          487-600 (Encoder)
    sgm/modules/distributions/distributions.py:24-41  DiagonalGaussianDistribution.sample
    sgm/models/diffusion.py:138-151     encode_first_stage (* scale_factor)

Same constructor kwargs and state-dict keys as the reference (`encoder.*`, `quant_conv.*`; `decoder.*` /
`post_quant_conv.*` keys of a checkpoint are ignored by `load_state_dict(strict=False)`), NHWC bf16 activations on the
same kernels as the UNet: k_conv_in (Cin = 3), implicit-GEMM 3x3 convs (the (0,1,0,1)-padded stride-2 Downsample is the
`pad=0` form), GroupNorm(eps 1e-6)+swish, 1x1 convs as linears.  The one single-head attention of dim `block_in` (512)
over H/8*W/8 tokens runs per frame as GEMM (fp32 logits) -> row softmax -> GEMM.  `quant_conv` (1x1, 8 -> 8) is folded
into `conv_out`'s weights at pack time (exact algebra: W' = Wq Wc, b' = Wq bc + bq).

Decode side (model.py:604-748 Decoder, :58-71 Upsample; autoencoder.py:490-506 decode; diffusion.py:117-136
decode_first_stage): `post_quant_conv` (1x1, embed -> z) is folded into the decoder's `conv_in` (z = 4 input channels: the
UNet's small-Cin kernel), the nearest-2x Upsample is folded into its conv's addressing (`up=2`), `conv_out` (128 -> 3) runs on
the UNet's 4-output-channel kernel with a zero fourth filter.  SVD's first stage (`AutoencodingEngine`, svd.yaml:98-133) uses
`VideoDecoder` (temporal_ae.py:293-349, time_mode conv-only): every ResnetBlock gains a 3-D time stack (GroupNorm over
(c/32, t, h, w), [3,1,1] convs: the video UNet's temporal-conv kernel) merged by a learned alpha, and conv_out a [3,1,1] frame mix.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from ._lib import VidsegError
from .unet import GroupNorm32, _meta


def Normalize(in_channels, num_groups=32):
    return _meta(nn.GroupNorm, num_groups, in_channels, eps=1e-6, affine=True)


class Downsample(nn.Module):
    """model.py:72-91 with_conv=True: pad (0,1,0,1) then 3x3 stride 2 without padding."""

    def __init__(self, in_channels, with_conv=True):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("Downsample(with_conv=False) (avg_pool) is not on the path")
        self.conv = _meta(nn.Conv2d, in_channels, in_channels, 3, stride=2, padding=0)

    def pack(self, dev):
        self.w, self.b = ops.pack_conv3x3(self.conv.weight, dev), ops.f32(self.conv.bias, dev)

    def run(self, x):
        return ops.conv3x3(x, self.w, self.b, stride=2, pad=0)


class Upsample(nn.Module):
    """model.py:58-71 with_conv=True: nearest x2, then 3x3 conv (one kernel: the upsample is an address shift)."""

    def __init__(self, in_channels, with_conv=True):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("Upsample(with_conv=False) is not on the path")
        self.conv = _meta(nn.Conv2d, in_channels, in_channels, 3, stride=1, padding=1)

    def pack(self, dev):
        self.w, self.b = ops.pack_conv3x3(self.conv.weight, dev), ops.f32(self.conv.bias, dev)

    def run(self, x):
        return ops.conv3x3(x, self.w, self.b, up=2)


class ResnetBlock(nn.Module):
    """model.py:94-151 with temb_channels = 0 (the autoencoder has no timestep embedding)."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=0):
        super().__init__()
        if conv_shortcut or temb_channels:
            raise NotImplementedError("ResnetBlock(conv_shortcut / temb) is not used by the first stage")
        self.in_channels = in_channels
        self.out_channels = in_channels if out_channels is None else out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = _meta(nn.Conv2d, in_channels, self.out_channels, 3, padding=1)
        self.norm2 = Normalize(self.out_channels)
        self.conv2 = _meta(nn.Conv2d, self.out_channels, self.out_channels, 3, padding=1)
        if self.in_channels != self.out_channels:
            self.nin_shortcut = _meta(nn.Conv2d, in_channels, self.out_channels, 1)

    def pack(self, dev):
        self.g1, self.b1 = ops.f32(self.norm1.weight, dev), ops.f32(self.norm1.bias, dev)
        self.g2, self.b2 = ops.f32(self.norm2.weight, dev), ops.f32(self.norm2.bias, dev)
        self.w1, self.cb1 = ops.pack_conv3x3(self.conv1.weight, dev), ops.f32(self.conv1.bias, dev)
        self.w2, self.cb2 = ops.pack_conv3x3(self.conv2.weight, dev), ops.f32(self.conv2.bias, dev)
        if self.in_channels != self.out_channels:
            self.ws = ops.pack_linear(self.nin_shortcut.weight.reshape(self.out_channels, self.in_channels), dev)
            self.bs = ops.f32(self.nin_shortcut.bias, dev)

    def run(self, x):
        h = ops.groupnorm(x, self.g1, self.b1, eps=1e-6, silu=True)
        h = ops.conv3x3(h, self.w1, self.cb1)
        h = ops.groupnorm(h, self.g2, self.b2, eps=1e-6, silu=True)
        res = x if self.in_channels == self.out_channels else ops.linear(x, self.ws, self.bs)
        return ops.conv3x3(h, self.w2, self.cb2, residual=res)


class AttnBlock(nn.Module):
    """model.py:161-202: GroupNorm -> q/k/v 1x1 convs -> softmax(q k^T / sqrt(C)) v (one head of dim C) -> proj_out -> + x."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = _meta(nn.Conv2d, in_channels, in_channels, 1)
        self.k = _meta(nn.Conv2d, in_channels, in_channels, 1)
        self.v = _meta(nn.Conv2d, in_channels, in_channels, 1)
        self.proj_out = _meta(nn.Conv2d, in_channels, in_channels, 1)

    def pack(self, dev):
        C = self.in_channels
        self.g, self.b = ops.f32(self.norm.weight, dev), ops.f32(self.norm.bias, dev)
        self.w_qkv = ops.pack_linear(torch.cat([m.weight.reshape(C, C) for m in (self.q, self.k, self.v)], 0), dev)
        self.b_qkv = ops.f32(torch.cat([self.q.bias, self.k.bias, self.v.bias], 0), dev)
        self.w_o, self.b_o = ops.pack_linear(self.proj_out.weight.reshape(C, C), dev), ops.f32(self.proj_out.bias, dev)

    def run(self, x):
        B, H, W, C = x.shape
        N = H * W
        t = ops.groupnorm(x, self.g, self.b, eps=1e-6, silu=False).view(B, N, C)
        qkv = ops.linear(t, self.w_qkv, self.b_qkv)                                   # [B, N, 3C]
        att = torch.empty((B, N, C), dtype=ops.act_dtype(), device=x.device)
        Np = -(-N // 64) * 64                                                          # P @ V contracts over the tokens: a GEMM K, % 64
        for b in range(B):                                                             # one frame at a time: N x N fp32 logits
            q = qkv[b, :, :C].contiguous()
            k = qkv[b, :, C:2 * C].contiguous()
            vt = qkv[b, :, 2 * C:].t().contiguous()                                    # [C, N]: the K-contiguous "weight" of P @ V
            logits = ops.linear(q, k, out_f32=True)                                    # q k^T, fp32 [N, N]
            p = ops.softmax_rows(logits, float(C) ** -0.5)                             # SDPA's default scale (model.py:189-191)
            if Np != N:                                                                # e.g. a 64 x 96 frame: 96 tokens -> zero columns up to 128 (exact)
                p, vt = torch.nn.functional.pad(p, (0, Np - N)), torch.nn.functional.pad(vt, (0, Np - N))
            att[b] = ops.linear(p, vt)
        out = ops.linear(att, self.w_o, self.b_o, residual=x.view(B, N, C))
        return out.view(B, H, W, C)


class _Level(nn.Module):
    pass


class Encoder(nn.Module):
    """model.py:487-600 (attn_resolutions = [], attn_type "vanilla")."""

    def __init__(self, *, ch, out_ch=None, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions=(), dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        if list(attn_resolutions) or use_linear_attn or attn_type not in ("vanilla", "vanilla-xformers"):
            raise NotImplementedError("Encoder: only the AutoencoderKL configuration (no per-level attention) is on the path")
        if ch % 64 != 0:
            raise NotImplementedError("Encoder: ch must be a multiple of 64 (implicit-GEMM K chunking)")
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.in_channels, self.z_channels, self.double_z = in_channels, z_channels, double_z
        self.conv_in = _meta(nn.Conv2d, in_channels, ch, 3, padding=1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            level = _Level()
            level.block = nn.ModuleList()
            level.attn = nn.ModuleList()
            for _ in range(num_res_blocks):
                level.block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, dropout=dropout))
                block_in = block_out
            if i_level != self.num_resolutions - 1:
                level.downsample = Downsample(block_in, resamp_with_conv)
            self.down.append(level)
        self.mid = _Level()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, dropout=dropout)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = _meta(nn.Conv2d, block_in, 2 * z_channels if double_z else z_channels, 3, padding=1)


class _TimeStack3D(nn.Module):
    """The dims=3 ResBlock of temporal_ae.VideoResBlock.time_stack (openaimodel.py ResBlock with emb_channels=0, skip_t_emb=True,
    kernel [3,1,1]): GroupNorm32(eps 1e-5) over (c/32, t, h, w) -> SiLU -> Conv3d, twice, + identity skip."""

    def __init__(self, channels):
        super().__init__()
        self.in_layers = nn.Sequential(_meta(GroupNorm32, 32, channels), nn.SiLU(),
                                       _meta(nn.Conv3d, channels, channels, (3, 1, 1), padding=(1, 0, 0)))
        self.out_layers = nn.Sequential(_meta(GroupNorm32, 32, channels), nn.SiLU(), nn.Dropout(p=0.0),
                                        _meta(nn.Conv3d, channels, channels, (3, 1, 1), padding=(1, 0, 0)))

    def pack(self, dev):
        self.g1, self.b1 = ops.f32(self.in_layers[0].weight, dev), ops.f32(self.in_layers[0].bias, dev)
        self.w1, self.cb1 = ops.pack_conv_temporal3(self.in_layers[2].weight, dev), ops.f32(self.in_layers[2].bias, dev)
        self.g2, self.b2 = ops.f32(self.out_layers[0].weight, dev), ops.f32(self.out_layers[0].bias, dev)
        self.w2, self.cb2 = ops.pack_conv_temporal3(self.out_layers[3].weight, dev), ops.f32(self.out_layers[3].bias, dev)

    def run(self, x, T):
        BT, H, W, C = x.shape
        h = ops.groupnorm(x.view(BT // T, T * H, W, C), self.g1, self.b1, eps=1e-5, silu=True).view(BT, H, W, C)
        h = ops.conv_temporal3(h, self.w1, self.cb1, T)
        h = ops.groupnorm(h.view(BT // T, T * H, W, C), self.g2, self.b2, eps=1e-5, silu=True).view(BT, H, W, C)
        return ops.conv_temporal3(h, self.w2, self.cb2, T, residual=x)


class VideoResBlock(ResnetBlock):
    """temporal_ae.py:18-81 (merge_strategy "learned"): spatial ResnetBlock, the 3-D time stack on top, and
    sigmoid(mix_factor) * temporal + (1 - sigmoid(mix_factor)) * spatial."""

    def __init__(self, *, in_channels, out_channels=None, dropout=0.0, **kw):
        super().__init__(in_channels=in_channels, out_channels=out_channels, dropout=dropout)
        self.time_stack = _TimeStack3D(self.out_channels)
        self.mix_factor = nn.Parameter(torch.empty(1, device="meta"))

    def pack(self, dev):
        super().pack(dev)
        self.time_stack.pack(dev)
        self.mix = ops.f32(self.mix_factor, dev)

    def run(self, x, T=None):
        x = super().run(x)
        return ops.alpha_blend(self.time_stack.run(x, T), x, self.mix)       # alpha weighs the TEMPORAL branch here (temporal_ae.py:77-78)


class AE3DConv(nn.Conv2d):
    """temporal_ae.py:84-107: the 2-D conv plus a Conv3d [3,1,1] over frames on its output channels."""

    def __init__(self, in_channels, out_channels, kernel_size=3, padding=1, device="meta"):
        super().__init__(in_channels, out_channels, kernel_size, padding=padding, device=device)
        self.time_mix_conv = nn.Conv3d(out_channels, out_channels, (3, 1, 1), padding=(1, 0, 0), device=device)


class Decoder(nn.Module):
    """model.py:604-748 (attn_resolutions = [], attn_type "vanilla", no tanh_out / give_pre_end)."""

    video = False

    def _resblock(self, **kw):
        return ResnetBlock(**kw)

    def _conv_out(self, cin, cout):
        return _meta(nn.Conv2d, cin, cout, 3, padding=1)

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions=(), dropout=0.0, resamp_with_conv=True,
                 in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        if list(attn_resolutions) or use_linear_attn or give_pre_end or tanh_out or attn_type not in ("vanilla", "vanilla-xformers"):
            raise NotImplementedError("Decoder: only the AutoencoderKL configuration is on the path")
        if ch % 64 != 0 or out_ch > 4:
            raise NotImplementedError("Decoder: ch must be a multiple of 64 and out_ch <= 4")
        self.ch, self.num_resolutions, self.num_res_blocks, self.out_ch = ch, len(ch_mult), num_res_blocks, out_ch
        block_in = ch * ch_mult[self.num_resolutions - 1]
        self.conv_in = _meta(nn.Conv2d, z_channels, block_in, 3, padding=1)
        self.mid = _Level()
        self.mid.block_1 = self._resblock(in_channels=block_in, out_channels=block_in, dropout=dropout)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = self._resblock(in_channels=block_in, out_channels=block_in, dropout=dropout)
        ups = []
        for i_level in reversed(range(self.num_resolutions)):
            block_out = ch * ch_mult[i_level]
            level = _Level()
            level.block = nn.ModuleList()
            level.attn = nn.ModuleList()
            for _ in range(num_res_blocks + 1):
                level.block.append(self._resblock(in_channels=block_in, out_channels=block_out, dropout=dropout))
                block_in = block_out
            if i_level != 0:
                level.upsample = Upsample(block_in, resamp_with_conv)
            ups.insert(0, level)                                     # `up.0` is the full-resolution level, like the reference
        self.up = nn.ModuleList(ups)
        self.norm_out = Normalize(block_in)
        self.conv_out = self._conv_out(block_in, out_ch)


class VideoDecoder(Decoder):
    """temporal_ae.py:293-349 with time_mode "conv-only" (svd.yaml:119-133): every ResnetBlock is a VideoResBlock, conv_out an
    AE3DConv; the mid attention stays the plain single-head AttnBlock."""
    video = True

    def __init__(self, *args, video_kernel_size=3, alpha=0.0, merge_strategy="learned", time_mode="conv-only", **kwargs):
        ks = list(video_kernel_size) if isinstance(video_kernel_size, (list, tuple)) else [video_kernel_size] * 3
        if time_mode != "conv-only" or merge_strategy != "learned" or ks != [3, 1, 1]:
            raise NotImplementedError("VideoDecoder: only time_mode 'conv-only', merge_strategy 'learned', kernel [3,1,1] is on the path")
        super().__init__(*args, **kwargs)

    def _resblock(self, **kw):
        return VideoResBlock(**kw)

    def _conv_out(self, cin, cout):
        return AE3DConv(cin, cout)


class AutoencoderKL(nn.Module):
    """sgm/models/autoencoder.py:508-522 (+ :437-506) with DiagonalGaussianRegularizer(sample=True)."""
    HOST_MASTERS = True          # engine.DiffusionEngine._apply leaves these modules alone (.to / .cuda / .half are no-ops)

    def __init__(self, embed_dim=4, ddconfig=None, lossconfig=None, loss_config=None, monitor=None, ckpt_path=None,
                 ckpt_engine=None, max_batch_size=None, **ignored):
        super().__init__()
        if ddconfig is None:
            raise ValueError("AutoencoderKL needs ddconfig")
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.post_quant_conv = _meta(nn.Conv2d, embed_dim, ddconfig["z_channels"], 1)
        zc = (1 + bool(ddconfig.get("double_z", True))) * ddconfig["z_channels"]
        self.quant_conv = _meta(nn.Conv2d, zc, (1 + bool(ddconfig.get("double_z", True))) * embed_dim, 1)
        self.embed_dim = embed_dim
        self.max_batch_size = max_batch_size
        self._packed_on = None

    # ------------------------------------------------------------------ packing
    def load_state_dict(self, state_dict, strict=False, assign=True):
        own = {k: v for k, v in state_dict.items() if k.split(".")[0] in ("encoder", "quant_conv", "decoder", "post_quant_conv")}
        self._packed_on = None
        return super().load_state_dict(own, strict=strict, assign=True)

    def pack(self, dev):
        self._pack_decoder(dev)
        self._packed_on = dev
        e = self.encoder
        self._enc_ready = not any(t.is_meta for t in list(e.parameters()) + (list(self.quant_conv.parameters()) if self.quant_conv is not None else []))
        if not self._enc_ready:                                       # decoder-only state dict
            return
        self.cin_w, self.cin_b = ops.pack_conv_in(e.conv_in.weight, dev), ops.f32(e.conv_in.bias, dev)
        for level in e.down:
            for blk in level.block:
                blk.pack(dev)
            if hasattr(level, "downsample"):
                level.downsample.pack(dev)
        for m in (e.mid.block_1, e.mid.attn_1, e.mid.block_2):
            m.pack(dev)
        self.no_g, self.no_b = ops.f32(e.norm_out.weight, dev), ops.f32(e.norm_out.bias, dev)
        # conv_out (3x3, C -> 2z) followed by quant_conv (1x1, 2z -> 2*embed): one conv with W' = Wq Wc, b' = Wq bc + bq
        wc = e.conv_out.weight.detach().double()
        if self.quant_conv is not None:
            wq = self.quant_conv.weight.detach().double().reshape(self.quant_conv.weight.shape[0], -1)
            wf = torch.einsum("oz,zikl->oikl", wq, wc)
            bf = wq @ e.conv_out.bias.detach().double() + self.quant_conv.bias.detach().double()
        else:                                                          # AutoencodingEngine: the encoder's moments are used as they are
            wf, bf = wc, e.conv_out.bias.detach().double()
        n_out = wf.shape[0]
        self.n_mom = n_out
        pad = (-n_out) % 8                                           # the GEMM epilogue writes 8 columns per lane
        if pad:
            wf = torch.cat([wf, wf.new_zeros((pad,) + tuple(wf.shape[1:]))], 0)
            bf = torch.cat([bf, bf.new_zeros(pad)], 0)
        self.out_w, self.out_b = ops.pack_conv3x3(wf.float(), dev), ops.f32(bf.float(), dev)

    def _pack_decoder(self, dev):
        d = self.decoder
        pq = self.post_quant_conv
        self._dec_ready = not any(t.is_meta for t in list(d.parameters()) + (list(pq.parameters()) if pq is not None else []))
        if not self._dec_ready:                                       # encoder-only checkpoint
            return
        # post_quant_conv (1x1, embed -> z) then conv_in (3x3, z -> C): W'[o,e,kh,kw] = sum_z Wc[o,z,kh,kw] Wp[z,e]; the bias of
        # the 1x1 passes through the 3x3's zero padding only where the tap is inside the image, so it stays a separate input
        # channel: a constant-one plane appended to z carries it exactly (Cin = embed + 1 <= 8)
        wc = d.conv_in.weight.detach().double()                                                                     # [C, z, 3, 3]
        if pq is not None:
            wp = pq.weight.detach().double().reshape(pq.weight.shape[0], -1)                                        # [z, e]
            w_e = torch.einsum("ozkl,ze->oekl", wc, wp)
            w_one = torch.einsum("ozkl,z->okl", wc, pq.bias.detach().double()).unsqueeze(1)
        else:                                                          # AutoencodingEngine: z goes straight into conv_in
            w_e, w_one = wc, wc.new_zeros((wc.shape[0], 1, 3, 3))
        wf = torch.cat([w_e, w_one], 1)                                                                             # [C, e+1, 3, 3]
        self.dec_cin = wf.shape[1]
        pad = (4 if self.dec_cin <= 4 else 8) - self.dec_cin
        if pad < 0:
            raise NotImplementedError("Decoder: embed_dim + 1 must be <= 8")
        wf = torch.cat([wf, wf.new_zeros((wf.shape[0], pad, 3, 3))], 1)
        self.dec_cin_pad = wf.shape[1]
        self.din_w, self.din_b = ops.pack_conv_in(wf.float(), dev), ops.f32(d.conv_in.bias, dev)
        for m in (d.mid.block_1, d.mid.attn_1, d.mid.block_2):
            m.pack(dev)
        for level in d.up:
            for blk in level.block:
                blk.pack(dev)
            if hasattr(level, "upsample"):
                level.upsample.pack(dev)
        self.dno_g, self.dno_b = ops.f32(d.norm_out.weight, dev), ops.f32(d.norm_out.bias, dev)
        wo, bo = d.conv_out.weight.detach(), d.conv_out.bias.detach()
        if wo.shape[0] < 4:                                            # the 4-output-channel kernel: zero filters for the rest
            wo = torch.cat([wo, wo.new_zeros((4 - wo.shape[0],) + tuple(wo.shape[1:]))], 0)
            bo = torch.cat([bo, bo.new_zeros(4 - bo.shape[0])], 0)
        self.dout_w, self.dout_b = ops.pack_conv_out(wo, dev), ops.f32(bo, dev)
        if d.video:
            tm = d.conv_out.time_mix_conv
            self.tmix_w = ops.f32(tm.weight.reshape(tm.weight.shape[0], tm.weight.shape[1], 3), dev)
            self.tmix_b = ops.f32(tm.bias, dev)

    # ------------------------------------------------------------------ forward
    def moments(self, x):
        """x: fp32 NCHW [B, 3, H, W] in [-1, 1] on the device -> fp32 NHWC [B, H/8, W/8, 2*embed] (mean | logvar)."""
        if not x.is_cuda:
            raise VidsegError("AutoencoderKL runs on a HIP device only (no CPU fallback)")
        if self._packed_on is None:
            self.pack(x.device)
        if not self._enc_ready:
            raise VidsegError("AutoencoderKL.encode: the loaded state dict has no encoder.* / quant_conv.* weights")
        e = self.encoder
        h = ops.conv_in(x.float().permute(0, 2, 3, 1).contiguous(), self.cin_w, self.cin_b)
        for level in e.down:
            for blk in level.block:
                h = blk.run(h)
            if hasattr(level, "downsample"):
                h = level.downsample.run(h)
        h = e.mid.block_1.run(h)
        h = e.mid.attn_1.run(h)
        h = e.mid.block_2.run(h)
        h = ops.groupnorm(h, self.no_g, self.no_b, eps=1e-6, silu=True)
        _, mom = ops.conv3x3(h, self.out_w, self.out_b, want_f32=True)
        return mom[..., :self.n_mom].contiguous() if mom.shape[-1] != self.n_mom else mom

    def encode(self, x, return_reg_log=False, noise=None, scale=1.0):
        """z = mean + std * randn (autoencoder.py:469-489, regularizers/__init__.py:21-31).  `noise` defaults to
        torch.randn(mean.shape) drawn on the HOST like the reference's posterior.sample() (so torch.manual_seed governs it)."""
        outs = []
        bs = self.max_batch_size or x.shape[0]
        for i in range(0, x.shape[0], bs):
            mom = self.moments(x[i:i + bs])
            B, h, w, Z2 = mom.shape
            nz = torch.randn((B, Z2 // 2, h, w)) if noise is None else noise[i:i + bs]
            outs.append(ops.gaussian_sample(mom, nz.to(mom.device, torch.float32).contiguous(), scale))
        z = torch.cat(outs, 0) if len(outs) > 1 else outs[0]
        return (z, {}) if return_reg_log else z

    def _decode_batch(self, z, T=None):
        d = self.decoder
        kw = {"T": T} if d.video else {}
        B, E, h, w = z.shape
        zin = torch.zeros((B, h, w, self.dec_cin_pad), dtype=torch.float32, device=z.device)
        zin[..., :E] = z.float().permute(0, 2, 3, 1)
        zin[..., E] = 1.0                                              # carries post_quant_conv's bias through the 3x3's padding
        x = ops.conv_in(zin, self.din_w, self.din_b)
        x = d.mid.block_1.run(x, **kw)
        x = d.mid.attn_1.run(x)
        x = d.mid.block_2.run(x, **kw)
        for i_level in reversed(range(d.num_resolutions)):
            level = d.up[i_level]
            for blk in level.block:
                x = blk.run(x, **kw)
            if hasattr(level, "upsample"):
                x = level.upsample.run(x)
        x = ops.groupnorm(x, self.dno_g, self.dno_b, eps=1e-6, silu=True)
        y = ops.conv_out4(x, self.dout_w, self.dout_b)
        if d.video:                                                    # AE3DConv.time_mix_conv over the frames of each video
            return ops.time_mix3(y, self.tmix_w, self.tmix_b, T, d.out_ch)
        return y[:, :d.out_ch]

    def decode(self, z, **decoder_kwargs):
        """autoencoder.py:490-506: post_quant_conv -> decoder, `max_batch_size` frames at a time.  z: fp32 NCHW [B, embed, h, w]
        on the device -> fp32 NCHW [B, out_ch, 8h, 8w]."""
        if not z.is_cuda:
            raise VidsegError("AutoencoderKL runs on a HIP device only (no CPU fallback)")
        if self._packed_on is None:
            self.pack(z.device)
        if not self._dec_ready:
            raise VidsegError("AutoencoderKL.decode: the loaded state dict has no decoder.* / post_quant_conv.* weights")
        bs = self.max_batch_size or z.shape[0]
        T = decoder_kwargs.get("timesteps")
        if self.decoder.video:
            if T is None or bs % T or z.shape[0] % T:
                raise VidsegError("VideoDecoder.decode needs timesteps=T with whole videos per call (diffusion.py:126-129)")
        outs = [self._decode_batch(z[i:i + bs], T) for i in range(0, z.shape[0], bs)]
        return torch.cat(outs, 0) if len(outs) > 1 else outs[0].contiguous()


class AutoencoderKLModeOnly(AutoencoderKL):
    """sgm/models/autoencoder.py:580-595: DiagonalGaussianRegularizer(sample=False) -- `encode` returns the posterior mode."""

    def encode(self, x, return_reg_log=False, noise=None, scale=1.0):
        bs = self.max_batch_size or x.shape[0]
        outs = []
        for i in range(0, x.shape[0], bs):
            mom = self.moments(x[i:i + bs])
            outs.append(mom[..., :mom.shape[-1] // 2].permute(0, 3, 1, 2) * scale)
        z = (torch.cat(outs, 0) if len(outs) > 1 else outs[0]).contiguous()
        return (z, {}) if return_reg_log else z


class AutoencodingEngine(AutoencoderKL):
    """sgm/models/autoencoder.py:77-254 as svd.yaml:98-133 configures it: `encoder_config` (the plain Encoder), `decoder_config`
    (temporal_ae.VideoDecoder or the image Decoder), DiagonalGaussianRegularizer -- no quant_conv / post_quant_conv."""

    def __init__(self, *, encoder_config, decoder_config, loss_config=None, regularizer_config=None, ckpt_path=None, **ignored):
        nn.Module.__init__(self)
        self.encoder = Encoder(**encoder_config.get("params", {}))
        video = str(decoder_config.get("target", "")).endswith("VideoDecoder")
        self.decoder = (VideoDecoder if video else Decoder)(**decoder_config.get("params", {}))
        self.quant_conv = self.post_quant_conv = None
        self.embed_dim = decoder_config.get("params", {}).get("z_channels", 4)
        self.max_batch_size = None
        self._packed_on = None

    def load_state_dict(self, state_dict, strict=False, assign=True):
        own = {k: v for k, v in state_dict.items() if k.split(".")[0] in ("encoder", "decoder")}
        self._packed_on = None
        return nn.Module.load_state_dict(self, own, strict=strict, assign=True)


def encode_first_stage(first_stage_model: AutoencoderKL, x, scale_factor=0.18215, n_samples=None, noise=None):
    """sgm/models/diffusion.py:138-151: chunked encode, then * scale_factor (folded into the sampling kernel)."""
    n = n_samples or x.shape[0]
    outs = [first_stage_model.encode(x[i:i + n], noise=None if noise is None else noise[i:i + n], scale=scale_factor)
            for i in range(0, x.shape[0], n)]
    return torch.cat(outs, 0) if len(outs) > 1 else outs[0]


def decode_first_stage(first_stage_model: AutoencoderKL, z, scale_factor=0.18215, n_samples=None):
    """sgm/models/diffusion.py:117-136: z / scale_factor, decoded `n_samples` (en_and_decode_n_samples_a_time) at a time; a
    VideoDecoder is told the number of frames of each chunk (`timesteps`)."""
    z = z.float() * (1.0 / scale_factor)
    n = n_samples or z.shape[0]
    video = first_stage_model.decoder.video
    outs = [first_stage_model.decode(z[i:i + n], **({"timesteps": z[i:i + n].shape[0]} if video else {})) for i in range(0, z.shape[0], n)]
    return torch.cat(outs, 0) if len(outs) > 1 else outs[0]
