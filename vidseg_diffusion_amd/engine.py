"""`DiffusionEngine` without Lightning (sgm/models/diffusion.py:19-151, inference subset): the object the reference's drivers
build from `configs/inference/{sd_2_1,svd}.yaml` and then poke -- `.model` (`OpenAIWrapper(network)`), `.model.diffusion_model`,
`.denoiser`, `.sampler`, `.conditioner`, `.first_stage_model`, `.scale_factor`, `.en_and_decode_n_samples_a_time`,
`.encode_first_stage`, `.decode_first_stage`, `load_state_dict(sd, strict=False)` with the checkpoint's key prefixes
(SURVEY.md §8(b)4).  Every `target:` string resolves through `util.instantiate_from_config`, i.e. to this package's mirrors.
The pipeline functions take it wherever they take `pipeline.Engine` (they read `.model`, `.denoiser`, `.sampler`, `.video`).
"""
from __future__ import annotations

from typing import Optional

import torch.nn as nn

from .sampling import OpenAIWrapper
from .util import default, instantiate_from_config

UNCONDITIONAL_CONFIG = {"target": "sgm.modules.GeneralConditioner", "params": {"emb_models": []}}


class DiffusionEngine(nn.Module):
    def __init__(self, network_config, denoiser_config, first_stage_config, conditioner_config=None, sampler_config=None,
                 optimizer_config=None, scheduler_config=None, loss_fn_config=None, network_wrapper=None, ckpt_path=None,
                 use_ema=False, ema_decay_rate=0.9999, scale_factor: float = 1.0, disable_first_stage_autocast=False,
                 input_key: str = "jpg", log_keys=None, no_cond_log=False, compile_model=False,
                 en_and_decode_n_samples_a_time: Optional[int] = None):
        super().__init__()
        if use_ema or loss_fn_config is not None or compile_model:
            raise NotImplementedError("DiffusionEngine: the training-side options (EMA, loss, torch.compile) are not on the path")
        self.input_key, self.log_keys = input_key, log_keys
        network = instantiate_from_config(network_config)
        self.model = OpenAIWrapper(network)                                              # diffusion.py:49-51
        self.denoiser = instantiate_from_config(denoiser_config)
        self.sampler = instantiate_from_config(sampler_config) if sampler_config is not None else None
        self.conditioner = instantiate_from_config(default(conditioner_config, UNCONDITIONAL_CONFIG))
        self.first_stage_model = instantiate_from_config(first_stage_config)             # diffusion.py:103-108
        self.scale_factor = scale_factor
        self.disable_first_stage_autocast = disable_first_stage_autocast
        self.en_and_decode_n_samples_a_time = en_and_decode_n_samples_a_time
        self.video = hasattr(network, "forward_nhwc") and type(network).__name__ == "VideoUNet"
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path)

    def _apply(self, fn, recurse=True):
        """`.to(device)` / `.cuda()` / `.half()` as the drivers call them on the engine (svd_pipeline_vspw.py:566-569, `load_model`).
        The modules that hold HOST MASTERS (class flag `HOST_MASTERS`: the UNet, the first stage, the OpenCLIP towers -- fp32 on the
        host, possibly still on the meta device before a checkpoint is loaded; the kernels' device copies are packed from them on first
        use, on the device the inputs live on) are left alone.  Everything else -- e.g. an ordinary torch embedder that
        GeneralConditioner built through `instantiate_from_config` and that owns real parameters -- follows the call as in torch."""
        def walk(m):
            for child in m.children():
                if getattr(child, "HOST_MASTERS", False):
                    continue
                walk(child)
                if any(p.is_meta for p in child.parameters(recurse=False)):
                    continue                                              # a named slot that no checkpoint has filled yet
                nn.Module._apply(child, fn, recurse=False)
        walk(self)
        return self

    # ------------------------------------------------------------------ checkpoints (diffusion.py:85-101)
    def init_from_ckpt(self, path: str):
        """diffusion.py:85-101: the RAW checkpoint dict (Lightning prefixes kept: `model.diffusion_model.*`, `first_stage_model.*`,
        `conditioner.embedders.N.*`) goes through `load_state_dict(strict=False)`; `.ckpt` -> torch.load(...)["state_dict"],
        `.safetensors` -> safetensors; anything else is NotImplementedError like the reference."""
        from .util import load_raw_checkpoint
        missing, unexpected = self.load_state_dict(load_raw_checkpoint(path), strict=False)
        print(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")
        if missing:
            print(f"Missing Keys: {missing}")
        if unexpected:
            print(f"Unexpected Keys: {unexpected}")
        return missing, unexpected

    def load_state_dict(self, state_dict, strict=False, assign=True):
        """Keys as in the released checkpoints, routed by prefix: `model.diffusion_model.*` -> the network,
        `first_stage_model.*` -> the first stage, `conditioner.embedders.N.*` -> embedder N when it owns parameters (SVD's
        `VideoPredictionEmbedderWithEncoder.encoder`, svd.yaml:66-91), `cond_stage_model.*` (the LDM layout of the SD 2.1 checkpoint) ->
        the first FrozenOpenCLIPEmbedder.  Keys of embedders that are stand-ins here (the OpenCLIP
        towers, `PrecomputedEmbedder`) and `denoiser.*` buffers are ignored; anything else is reported as unexpected.
        Returns (missing, unexpected) with the checkpoint's full key names."""
        parts = {"model.diffusion_model.": {}, "first_stage_model.": {}}
        cond, legacy = {}, {}
        unexpected = []
        for k, v in state_dict.items():
            for pre, d in parts.items():
                if k.startswith(pre):
                    d[k[len(pre):]] = v
                    break
            else:
                if k.startswith("conditioner.embedders."):
                    idx, _, rest = k[len("conditioner.embedders."):].partition(".")
                    if idx.isdigit() and rest:
                        cond.setdefault(int(idx), {})[rest] = v
                    else:
                        unexpected.append(k)
                elif k.startswith("cond_stage_model."):
                    legacy[k[len("cond_stage_model."):]] = v
                elif not k.startswith("conditioner.") and not k.startswith("denoiser."):
                    unexpected.append(k)
        missing = []
        embedders = list(getattr(self.conditioner, "embedders", []))
        if legacy:
            # The SD 2.1 checkpoint the SD driver loads (v2-1_512-ema-pruned.safetensors, sd_pipeline_vspw.py:663) is in the LDM layout:
            # the text encoder sits under `cond_stage_model.model.*` (open_clip names).  The reference reports those keys as unexpected
            # and takes the same weights from open_clip's pretrained download (modules.py:511-516), which this image cannot do -- so they
            # are routed to the first FrozenOpenCLIPEmbedder (whose `model.*` they are), unless the checkpoint also carries that
            # embedder's own `conditioner.embedders.N.*` keys.
            idx = next((i for i, e in enumerate(embedders) if type(e).__name__ == "FrozenOpenCLIPEmbedder"), None)
            if idx is None or idx in cond:
                unexpected.extend("cond_stage_model." + k for k in legacy)
            else:
                r = embedders[idx].load_state_dict(legacy, strict=False)
                missing.extend(f"conditioner.embedders.{idx}." + k for k in r[0])
                unexpected.extend("cond_stage_model." + k for k in r[1])

        def _load(module, sub, prefix):
            r = module.load_state_dict(sub, strict=False)
            miss = getattr(r, "missing_keys", r[0] if isinstance(r, tuple) else [])
            unex = getattr(r, "unexpected_keys", r[1] if isinstance(r, tuple) else [])
            missing.extend(prefix + k for k in miss)
            unexpected.extend(prefix + k for k in unex)

        if parts["model.diffusion_model."]:
            _load(self.model.diffusion_model, parts["model.diffusion_model."], "model.diffusion_model.")
        if parts["first_stage_model."]:
            _load(self.first_stage_model, parts["first_stage_model."], "first_stage_model.")
        for idx, sub in sorted(cond.items()):
            if idx >= len(embedders):
                unexpected.extend(f"conditioner.embedders.{idx}.{k}" for k in sub)
                continue
            emb = embedders[idx]
            owner = getattr(emb, "encoder", None)                         # VideoPredictionEmbedderWithEncoder: `.encoder.*`
            if owner is not None and any(True for _ in owner.parameters()):
                enc = {k[len("encoder."):]: v for k, v in sub.items() if k.startswith("encoder.")}
                unexpected.extend(f"conditioner.embedders.{idx}.{k}" for k in sub if not k.startswith("encoder."))
                _load(owner, enc, f"conditioner.embedders.{idx}.encoder.")
            elif any(True for _ in emb.parameters()):
                _load(emb, sub, f"conditioner.embedders.{idx}.")
            # else: a stand-in for a pretrained tower (PrecomputedEmbedder) -- its checkpoint keys have nowhere to go
        return missing, unexpected

    # ------------------------------------------------------------------ first stage (diffusion.py:117-151)
    def encode_first_stage(self, x, noise=None):
        from .vae import encode_first_stage
        return encode_first_stage(self.first_stage_model, x, self.scale_factor, self.en_and_decode_n_samples_a_time, noise=noise)

    def decode_first_stage(self, z):
        from .vae import decode_first_stage
        return decode_first_stage(self.first_stage_model, z, self.scale_factor, self.en_and_decode_n_samples_a_time)


def engine_from_config(config: dict) -> DiffusionEngine:
    """`instantiate_from_config(config.model)` of the drivers (sd_pipeline_vspw.py:553-580): `config` is the parsed YAML (a dict
    with a `model:` entry, or that entry itself)."""
    cfg = config.get("model", config)
    params = dict(cfg.get("params", {}))
    return DiffusionEngine(**params)
