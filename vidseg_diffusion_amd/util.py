"""The reference's plug-in/config helpers on the path (sgm/util.py): instantiate_from_config :168-185,
append_dims :192-199, default, load_target_features :277-296, load_xt :298-310,
get_modulate_timestep_frames :313-326 -- plus `install_sgm_aliases`, which lets the reference's YAML
`target:` strings (sgm.modules.diffusionmodules....) resolve to this package's classes."""
from __future__ import annotations

import importlib
import os
import sys
import types
from inspect import isfunction

import torch


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if isfunction(d) else d


def append_zero(x):
    return torch.cat([x, x.new_zeros([1])])


def append_dims(x, target_dims):
    dims_to_append = target_dims - x.ndim
    if dims_to_append < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * dims_to_append]


_ALIASES = {
    "sgm.modules.diffusionmodules.openaimodel": "vidseg_diffusion_amd.unet",
    "sgm.modules.diffusionmodules.video_model": "vidseg_diffusion_amd.video_unet",
    "sgm.modules.diffusionmodules.denoiser": "vidseg_diffusion_amd.sampling",
    "sgm.modules.diffusionmodules.denoiser_scaling": "vidseg_diffusion_amd.sampling",
    "sgm.modules.diffusionmodules.discretizer": "vidseg_diffusion_amd.sampling",
    "sgm.modules.diffusionmodules.guiders": "vidseg_diffusion_amd.sampling",
    "sgm.modules.diffusionmodules.sampling": "vidseg_diffusion_amd.sampling",
    "sgm.modules.diffusionmodules.wrappers": "vidseg_diffusion_amd.sampling",
    "sgm.util": "vidseg_diffusion_amd.util",
    "scripts.sampling.feature_extraction": "vidseg_diffusion_amd.feature_extraction",
    "scripts.sampling.process_output": "vidseg_diffusion_amd.process_output",
    "sgm.modules.encoders.modules": "vidseg_diffusion_amd.conditioner",
    "sgm.models.autoencoder": "vidseg_diffusion_amd.vae",
    "sgm.models.diffusion": "vidseg_diffusion_amd.engine",
    "sgm.modules.autoencoding.temporal_ae": "vidseg_diffusion_amd.vae",
}
# YAML targets that name a class re-exported by a package __init__ (svd.yaml / sd_2_1.yaml: `sgm.modules.GeneralConditioner`):
# resolved by get_obj_from_str only -- the package name itself is never replaced in sys.modules
_TARGET_ONLY = {"sgm.modules": "vidseg_diffusion_amd.conditioner"}


def get_obj_from_str(string, reload=False, invalidate_cache=True):
    module, cls = string.rsplit(".", 1)
    module = _ALIASES.get(module, _TARGET_ONLY.get(module, module))
    if invalidate_cache:
        importlib.invalidate_caches()
    if reload:
        importlib.reload(importlib.import_module(module))
    return getattr(importlib.import_module(module, package=None), cls)


def instantiate_from_config(config):
    if "target" not in config:
        if config == "__is_first_stage__":
            return None
        elif config == "__is_unconditional__":
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def install_sgm_aliases():
    """Register `sgm.*` / `scripts.sampling.feature_extraction` module names that resolve to this package, so
    `from sgm.util import instantiate_from_config` etc. in the unmodified drivers import the MI355X path."""
    for pkg in ("sgm", "sgm.modules", "sgm.modules.diffusionmodules", "sgm.modules.encoders", "sgm.modules.autoencoding", "sgm.models",
                "scripts", "scripts.sampling"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
    for alias, real in _ALIASES.items():
        try:
            sys.modules[alias] = importlib.import_module(real)
        except ImportError:
            pass


def _store_lookup(feature_maps_folder, exp_name, name):
    from .feature_extraction import FeatureStore
    return FeatureStore._stores.get(os.path.join(feature_maps_folder, exp_name, "feature_maps"), {}).get(name)


def load_target_features(feature_maps_folder, exp_name, timestep, injected_block_type, injected_feature_types, block_idx, device):
    """sgm/util.py:277-296, served from the in-HBM FeatureStore first, `.pt` files second."""
    path = os.path.join(feature_maps_folder, exp_name, "feature_maps")
    current = {}
    for feature_type in injected_feature_types:
        name = f"{injected_block_type}_block_{block_idx}_{feature_type}_time_{timestep}"
        t = _store_lookup(feature_maps_folder, exp_name, name)
        if t is None and os.path.exists(os.path.join(path, name + ".pt")):
            t = torch.load(os.path.join(path, name + ".pt")).detach()
        if t is not None:
            current[name] = t.to(device)
    if len(current) == 0:
        raise ValueError(f"No feature maps found for block {injected_block_type}_block_{block_idx} at timestep {timestep} in {path}")
    return current


def load_xt(feature_maps_folder, exp_name, timestep, device):
    """sgm/util.py:298-310."""
    name = f"xt_time_{timestep}"
    t = _store_lookup(feature_maps_folder, exp_name, name)
    path = os.path.join(feature_maps_folder, exp_name, "feature_maps", name + ".pt")
    if t is None and os.path.exists(path):
        t = torch.load(path).detach()
    if t is None:
        raise ValueError(f"No feature maps found for xt at timestep {timestep} in {os.path.dirname(path)}")
    return t.to(device)


def get_modulate_timestep_frames(start_timestep, end_timestep=None, num_frames=14, schedule="constant"):
    """sgm/util.py:313-326."""
    if schedule == "constant":
        return {}
    elif schedule == "linear":
        out = {t: [] for t in range(start_timestep, end_timestep - 1, -1)}
        for frame_id in range(num_frames):
            out[int(start_timestep + (end_timestep - start_timestep) * frame_id / (num_frames - 1))].append(frame_id)
        return out
    raise ValueError(f"Unknown modulate timestep frames schedule: {schedule}")


def load_raw_checkpoint(path):
    """The checkpoint's full state dict with its Lightning key prefixes untouched (sgm/models/diffusion.py:85-95):
    `.ckpt` -> torch.load(...)["state_dict"], `.safetensors` -> safetensors' load_file; other suffixes are not supported
    (the reference raises NotImplementedError too).  Host-side file parsing only."""
    if path.endswith("safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    if path.endswith("ckpt"):
        sd = torch.load(path, map_location="cpu")
        return sd["state_dict"] if "state_dict" in sd else sd
    raise NotImplementedError(f"checkpoint format of {path!r} (expected .ckpt or .safetensors)")


def load_checkpoint_state_dict(path, prefix="model.diffusion_model."):
    """The UNet part of an upstream checkpoint as a plain state dict for ``UNetModel/VideoUNet.load_state_dict`` --
    what ``load_model_from_config`` + ``DiffusionEngine.init_from_ckpt`` do for the network (sd_pipeline_vspw.py:553-580,
    sgm/models/diffusion.py:87-103): ``.safetensors`` via safetensors, ``.ckpt`` via torch.load(...)["state_dict"]; keys
    carry the ``model.diffusion_model.`` prefix of the Lightning module, which is stripped here.  Host-side file parsing
    only; weights reach the device in ``pack()``."""
    import torch
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path, device="cpu")
    else:
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd)
    out = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    if not out:                                                  # already a bare UNet state dict
        out = dict(sd)
    return out


def shared_prefix(modulate_params, is_modulate_step):
    """The Step-4 sweep's shared first-evaluation prefix (pipeline.modulation_sweep): `modulate_params["shared_prefix"]` =
    {"step": the sweep's first sampler step, "fork": the first modulated decoder block, "state": None until computed}.  Returns it when
    THIS evaluation is that step's modulated evaluation, else None (every other evaluation runs in full)."""
    if modulate_params is None or not is_modulate_step:
        return None
    sp = modulate_params.get("shared_prefix")
    if sp is None or modulate_params.get("timestep") != sp["step"]:
        return None
    return sp
