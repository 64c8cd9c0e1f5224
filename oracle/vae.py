"""Oracle (CPU restatement) of the first stage (encode and decode) -- test infrastructure only.

    sgm/modules/diffusionmodules/model.py:487-600  Encoder.forward (ResnetBlock :94-151, Downsample :72-91, AttnBlock :161-202)
    sgm/models/autoencoder.py:469-489              AutoencodingEngineLegacy.encode (quant_conv, regularizer)
    sgm/modules/distributions/distributions.py:24-41  DiagonalGaussianDistribution (clamp, std, sample)
    sgm/models/diffusion.py:138-151                encode_first_stage (* scale_factor)

    sgm/modules/diffusionmodules/model.py:604-748  Decoder.forward (Upsample :58-71)
    sgm/models/autoencoder.py:490-506              decode (post_quant_conv -> decoder)
    sgm/models/diffusion.py:117-136                decode_first_stage (z / scale_factor)
    sgm/modules/autoencoding/temporal_ae.py:18-107, 293-349  VideoResBlock, AE3DConv, VideoDecoder (time_mode conv-only)

Functional torch-CPU fp32, driven by the reference's state-dict keys.  Pinned by tests/golden/vae_encoder_narrow.npz and
tests/golden/vae_{decoder,video_decoder}_narrow.npz (tools/gen_golden_vae.py runs the reference's own Encoder / Decoder /
DiagonalGaussianDistribution).
`round_bf16=True` rounds activations where the HIP path stores bf16 (format-error yardstick for the GPU tests).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _bf(x, on):
    """Round where the HIP path stores 16-bit activations: on = "f16" (default build), True / "bf16", or falsy (pure fp32)."""
    if not on:
        return x
    return x.half().float() if on == "f16" else x.bfloat16().float()


class VAEEncoderOracle:
    def __init__(self, state_dict, round_bf16=False):
        self.sd = {k: v.float() for k, v in state_dict.items()}
        self.rb = round_bf16

    def has(self, p):
        return any(k.startswith(p) for k in self.sd)

    def gn(self, x, p):
        return F.group_norm(x, 32, self.sd[p + ".weight"], self.sd[p + ".bias"], 1e-6)

    def conv(self, x, p, stride=1, pad=1):
        return F.conv2d(x, self.sd[p + ".weight"], self.sd[p + ".bias"], stride=stride, padding=pad)

    def swish(self, x):
        return x * torch.sigmoid(x)

    def resnet(self, x, p):
        h = _bf(self.swish(self.gn(x, p + ".norm1")), self.rb)
        h = _bf(self.conv(h, p + ".conv1"), self.rb)
        h = _bf(self.swish(self.gn(h, p + ".norm2")), self.rb)
        h = self.conv(h, p + ".conv2")
        if self.has(p + ".nin_shortcut."):
            x = _bf(self.conv(x, p + ".nin_shortcut", pad=0), self.rb)
        return _bf(x + h, self.rb)

    def attn(self, x, p):
        h = _bf(self.gn(x, p + ".norm"), self.rb)
        q, k, v = (_bf(self.conv(h, f"{p}.{n}", pad=0), self.rb) for n in "qkv")
        b, c, hh, ww = q.shape
        q, k, v = (t.reshape(b, c, hh * ww).transpose(1, 2) for t in (q, k, v))
        w = torch.softmax((q @ k.transpose(1, 2)) * (c ** -0.5), dim=-1)
        o = _bf(_bf(w, self.rb) @ v, self.rb).transpose(1, 2).reshape(b, c, hh, ww)
        return _bf(x + self.conv(o, p + ".proj_out", pad=0), self.rb)

    def moments(self, x):
        """x [B, 3, H, W] fp32 -> [B, 2*embed, H/8, W/8] (quant_conv applied)."""
        h = _bf(self.conv(x, "encoder.conv_in"), self.rb)
        lvl = 0
        while self.has(f"encoder.down.{lvl}."):
            j = 0
            while self.has(f"encoder.down.{lvl}.block.{j}."):
                h = self.resnet(h, f"encoder.down.{lvl}.block.{j}")
                j += 1
            if self.has(f"encoder.down.{lvl}.downsample."):
                h = _bf(self.conv(F.pad(h, (0, 1, 0, 1)), f"encoder.down.{lvl}.downsample.conv", stride=2, pad=0), self.rb)
            lvl += 1
        h = self.resnet(h, "encoder.mid.block_1")
        h = self.attn(h, "encoder.mid.attn_1")
        h = self.resnet(h, "encoder.mid.block_2")
        h = _bf(self.swish(self.gn(h, "encoder.norm_out")), self.rb)
        h = self.conv(h, "encoder.conv_out")
        return self.conv(h, "quant_conv", pad=0)

    def encode(self, x, noise, scale_factor=1.0):
        mom = self.moments(x)
        mean, logvar = torch.chunk(mom, 2, dim=1)
        std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
        return scale_factor * (mean + std * noise)


class VAEDecoderOracle(VAEEncoderOracle):
    """decoder.* / post_quant_conv.* keys; same block restatements as the encoder."""

    def gn5(self, x, p):
        return F.group_norm(x, 32, self.sd[p + ".weight"], self.sd[p + ".bias"], 1e-5)      # GroupNorm32 of openaimodel.ResBlock

    def vresnet(self, x, p, T):
        """temporal_ae.py:18-81 VideoResBlock: ResnetBlock, then the dims=3 ResBlock time_stack and the learned alpha merge."""
        x = self.resnet(x, p)
        if T is None or not self.has(p + ".time_stack."):
            return x
        bt, c, hh, ww = x.shape
        v = x.reshape(bt // T, T, c, hh, ww).permute(0, 2, 1, 3, 4)                          # (b t) c h w -> b c t h w
        q = p + ".time_stack"
        h = _bf(self.swish(self.gn5(v, q + ".in_layers.0")), self.rb)
        h = _bf(F.conv3d(h, self.sd[q + ".in_layers.2.weight"], self.sd[q + ".in_layers.2.bias"], padding=(1, 0, 0)), self.rb)
        h = _bf(self.swish(self.gn5(h, q + ".out_layers.0")), self.rb)
        h = F.conv3d(h, self.sd[q + ".out_layers.3.weight"], self.sd[q + ".out_layers.3.bias"], padding=(1, 0, 0))
        t = _bf(v + h, self.rb)
        a = torch.sigmoid(self.sd[p + ".mix_factor"])
        out = a * t + (1.0 - a) * v
        return _bf(out.permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww), self.rb)

    def decode(self, z, scale_factor=1.0, timesteps=None):
        """z [B, embed, h, w] -> [B, 3, 8h, 8w] (decode_first_stage: z / scale_factor first).  `timesteps` = frames per video
        for a VideoDecoder state dict (time_stack / time_mix_conv keys); post_quant_conv only if the state dict has one."""
        T = timesteps
        h = z * (1.0 / scale_factor)
        if self.has("post_quant_conv."):
            h = self.conv(h, "post_quant_conv", pad=0)
        h = _bf(self.conv(h, "decoder.conv_in"), self.rb)
        h = self.vresnet(h, "decoder.mid.block_1", T)
        h = self.attn(h, "decoder.mid.attn_1")
        h = self.vresnet(h, "decoder.mid.block_2", T)
        lvl = 0
        while self.has(f"decoder.up.{lvl + 1}."):
            lvl += 1
        for i in range(lvl, -1, -1):
            j = 0
            while self.has(f"decoder.up.{i}.block.{j}."):
                h = self.vresnet(h, f"decoder.up.{i}.block.{j}", T)
                j += 1
            if self.has(f"decoder.up.{i}.upsample."):
                h = _bf(self.conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), f"decoder.up.{i}.upsample.conv"), self.rb)
        h = _bf(self.swish(self.gn(h, "decoder.norm_out")), self.rb)
        h = self.conv(h, "decoder.conv_out")
        if T is not None and self.has("decoder.conv_out.time_mix_conv."):                    # AE3DConv, temporal_ae.py:84-107
            bt, c, hh, ww = h.shape
            v = h.reshape(bt // T, T, c, hh, ww).permute(0, 2, 1, 3, 4)
            v = F.conv3d(v, self.sd["decoder.conv_out.time_mix_conv.weight"], self.sd["decoder.conv_out.time_mix_conv.bias"], padding=(1, 0, 0))
            h = v.permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)
        return h
