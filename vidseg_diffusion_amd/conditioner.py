"""The parts of the reference's conditioner that are arithmetic rather than pretrained networks
(sgm/modules/encoders/modules.py):

    GeneralConditioner                  :71-184   embedder list -> {"vector", "crossattn", "concat"} dicts, force-zero and the
                                                  (c, uc) pair of get_unconditional_conditioning
    ConcatTimestepEmbedderND            :913-929  sinusoidal embedding of every scalar (fps id, motion bucket, cond_aug), concatenated
    VideoPredictionEmbedderWithEncoder  :951-1031 SVD's `cond_frames`: (optionally noise-augmented) conditioning frame through
                                                  the first stage's encoder (posterior MODE, AutoencoderKLModeOnly), repeated over frames

The OpenCLIP text / image embedders (FrozenOpenCLIPEmbedder, FrozenOpenCLIPImageEmbedder, FrozenOpenCLIPImagePredictionEmbedder) live in
`openclip.py` (the ViT-H towers on the HIP path); an embedding computed elsewhere still passes through them unchanged, and
`PrecomputedEmbedder` remains as the explicit stand-in for any pretrained embedder.
The sinusoid is evaluated in fp32 with torch on the device the inputs live on (a few hundred numbers: plumbing, not a kernel).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .util import instantiate_from_config


class AbstractEmbModel(nn.Module):
    """modules.py:27-68: the three attributes GeneralConditioner sets on every embedder."""

    def __init__(self):
        super().__init__()
        self.is_trainable = False
        self.ucg_rate = 0.0
        self.input_key = None


def timestep_embedding(t, dim, max_period=10000):
    """diffusionmodules/util.py:209-233 (repeat_only=False): cat(cos, sin) of t * exp(-ln(max_period) * i / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class ConcatTimestepEmbedderND(AbstractEmbModel):
    """modules.py:913-929: embeds each dimension independently and concatenates them."""

    def __init__(self, outdim):
        super().__init__()
        self.outdim = outdim

    def forward(self, x):
        if x.ndim == 1:
            x = x[:, None]
        assert x.ndim == 2
        b, dims = x.shape
        emb = timestep_embedding(x.reshape(-1), self.outdim)            # "b d -> (b d)"
        return emb.reshape(b, dims * self.outdim)                       # "(b d) d2 -> b (d d2)"


class PrecomputedEmbedder(AbstractEmbModel):
    """Stand-in for a pretrained embedder (OpenCLIP text / image towers): forwards what the batch already holds."""

    def __init__(self, **ignored):
        super().__init__()

    def forward(self, x):
        return x


class VideoPredictionEmbedderWithEncoder(AbstractEmbModel):
    """modules.py:951-1031 with is_ae=True (svd.yaml:66-91): encoder = the first stage in posterior-mode form."""

    def __init__(self, n_cond_frames: int, n_copies: int, encoder_config: Optional[dict] = None, sigma_sampler_config=None,
                 sigma_cond_config=None, is_ae: bool = False, scale_factor: float = 1.0, disable_encoder_autocast: bool = False,
                 en_and_decode_n_samples_a_time: Optional[int] = None, encoder=None):
        super().__init__()
        if sigma_sampler_config is not None or sigma_cond_config is not None:
            raise NotImplementedError("sigma_sampler / sigma_cond (training-time noise augmentation) are not on the inference path")
        if not is_ae:
            raise NotImplementedError("VideoPredictionEmbedderWithEncoder: only is_ae=True (AutoencoderKLModeOnly) is on the path")
        self.n_cond_frames, self.n_copies, self.scale_factor = n_cond_frames, n_copies, scale_factor
        self.en_and_decode_n_samples_a_time = en_and_decode_n_samples_a_time
        if encoder is None:
            from .vae import AutoencoderKL
            params = dict(encoder_config.get("params", {}))
            encoder = AutoencoderKL(embed_dim=params.get("embed_dim", 4), ddconfig=params["ddconfig"])
        self.encoder = encoder

    def forward(self, vid):
        """vid: fp32 NCHW [(b t), 3, H, W] conditioning frames -> [(b n_copies), t*z, H/8, W/8] (the `concat` conditioning)."""
        n = self.en_and_decode_n_samples_a_time or vid.shape[0]
        outs = []
        for i in range(0, vid.shape[0], n):
            mom = self.encoder.moments(vid[i:i + n])                                  # [B, h, w, 2z] fp32 NHWC
            outs.append(mom[..., :mom.shape[-1] // 2].permute(0, 3, 1, 2))            # posterior mode = mean (AutoencoderKLModeOnly)
        z = torch.cat(outs, 0) * self.scale_factor
        bt, c, h, w = z.shape
        z = z.reshape(bt // self.n_cond_frames, 1, self.n_cond_frames * c, h, w)      # "(b t) c h w -> b () (t c) h w"
        return z.expand(-1, self.n_copies, -1, -1, -1).reshape(-1, self.n_cond_frames * c, h, w).contiguous()


_TARGETS = {
    "sgm.modules.encoders.modules.ConcatTimestepEmbedderND": ConcatTimestepEmbedderND,
    "sgm.modules.encoders.modules.VideoPredictionEmbedderWithEncoder": VideoPredictionEmbedderWithEncoder,
}
_CLIP_TARGETS = ("FrozenOpenCLIPEmbedder", "FrozenOpenCLIPImageEmbedder", "FrozenOpenCLIPImagePredictionEmbedder")


def __getattr__(name):                                      # sgm.modules.encoders.modules.FrozenOpenCLIP* -> openclip.py (imports this module)
    if name in _CLIP_TARGETS:
        from . import openclip
        return getattr(openclip, name)
    raise AttributeError(name)


class GeneralConditioner(nn.Module):
    """modules.py:71-184 (inference subset: ucg dropout is a training feature and stays at rate 0)."""
    OUTPUT_DIM2KEYS = {2: "vector", 3: "crossattn", 4: "concat", 5: "concat"}
    KEY2CATDIM = {"vector": 1, "crossattn": 2, "concat": 1}

    def __init__(self, emb_models: List[dict]):
        super().__init__()
        embedders = []
        for cfg in emb_models:
            tgt = cfg.get("target")
            if isinstance(cfg, nn.Module):
                emb = cfg
            elif tgt in _TARGETS:
                emb = _TARGETS[tgt](**cfg.get("params", {}))
            elif tgt and tgt.startswith("sgm.modules.encoders.modules.") and tgt.rsplit(".", 1)[1] in _CLIP_TARGETS:
                emb = __getattr__(tgt.rsplit(".", 1)[1])(**cfg.get("params", {}))
            else:
                emb = instantiate_from_config(cfg)
            if not isinstance(emb, nn.Module) or not hasattr(emb, "forward"):
                raise TypeError(f"embedder {type(emb).__name__} has to be an AbstractEmbModel")
            emb.is_trainable = cfg.get("is_trainable", False) if isinstance(cfg, dict) else False
            emb.ucg_rate = cfg.get("ucg_rate", 0.0) if isinstance(cfg, dict) else 0.0
            if isinstance(cfg, dict):
                if "input_key" in cfg:
                    emb.input_key = cfg["input_key"]
                elif "input_keys" in cfg:
                    emb.input_keys = cfg["input_keys"]
                else:
                    raise KeyError(f"need either 'input_key' or 'input_keys' for embedder {type(emb).__name__}")
            embedders.append(emb)
        self.embedders = nn.ModuleList(embedders)

    @torch.no_grad()
    def forward(self, batch: Dict, force_zero_embeddings: Optional[List] = None) -> Dict:
        output = {}
        force_zero_embeddings = force_zero_embeddings or []
        for emb in self.embedders:
            if getattr(emb, "input_key", None) is not None:
                out = emb(batch[emb.input_key])
            else:
                out = emb(*[batch[k] for k in emb.input_keys])
            for e in (out if isinstance(out, (list, tuple)) else [out]):
                key = self.OUTPUT_DIM2KEYS[e.dim()]
                if getattr(emb, "input_key", None) in force_zero_embeddings:
                    e = torch.zeros_like(e)
                output[key] = torch.cat((output[key], e), self.KEY2CATDIM[key]) if key in output else e
        return output

    def get_unconditional_conditioning(self, batch_c, batch_uc=None, force_uc_zero_embeddings=None, force_cond_zero_embeddings=None):
        c = self(batch_c, force_cond_zero_embeddings)
        uc = self(batch_c if batch_uc is None else batch_uc, force_uc_zero_embeddings or [])
        return c, uc
