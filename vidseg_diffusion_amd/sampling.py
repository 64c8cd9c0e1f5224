"""Sampler, guider, denoiser, discretisation and wrapper classes with the reference's names, constructor
arguments and call signatures, so `instantiate_from_config` on the reference's YAML
(configs/inference/sd_2_1.yaml:6-15, 63-79; svd.yaml) builds these and the drivers call them unchanged:

    model.sampler(denoiser, x, cond=c, uc=uc, img_callback=cb, is_modulate=..., modulate_params=...,
                  uc_list=..., t_start=..., is_latent_blending=..., feature_height=..., feature_width=...)
    model.sampler.add_noise(x, cond, uc, num_steps, noise_level)
    model.denoiser(network, input, sigma, cond, ...)

Per-sample scalars (sigmas, c_in/c_out/c_skip, CFG scales) are host values, exactly the part the reference
also evaluates on tiny tensors; every operation on latents is a HIP kernel (ops.py).
"""
from __future__ import annotations

import os

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .util import append_zero, default, instantiate_from_config, load_xt


# ----------------------------------------------------------------------------- discretizer.py
class Discretization:
    def __call__(self, n, do_append_zero=True, device="cpu", flip=False):
        sigmas = self.get_sigmas(n, device="cpu")
        sigmas = append_zero(sigmas) if do_append_zero else sigmas
        return sigmas if not flip else torch.flip(sigmas, (0,))


class EDMDiscretization(Discretization):
    """discretizer.py:28-40."""

    def __init__(self, sigma_min=0.002, sigma_max=80.0, rho=7.0):
        self.sigma_min, self.sigma_max, self.rho = sigma_min, sigma_max, rho

    def get_sigmas(self, n, device="cpu"):
        ramp = torch.linspace(0, 1, n)
        min_inv_rho = self.sigma_min ** (1 / self.rho)
        max_inv_rho = self.sigma_max ** (1 / self.rho)
        return (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** self.rho


class LegacyDDPMDiscretization(Discretization):
    """discretizer.py:43-70 (+ make_beta_schedule 'linear', diffusionmodules/util.py:22-35)."""

    def __init__(self, linear_start=0.00085, linear_end=0.0120, num_timesteps=1000):
        self.num_timesteps = num_timesteps
        betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, num_timesteps, dtype=torch.float64) ** 2).numpy()
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0)

    def get_sigmas(self, n, device="cpu"):
        if n < self.num_timesteps:
            timesteps = np.linspace(self.num_timesteps - 1, 0, n, endpoint=False).astype(int)[::-1]
            alphas_cumprod = self.alphas_cumprod[timesteps]
        elif n == self.num_timesteps:
            alphas_cumprod = self.alphas_cumprod
        else:
            raise ValueError
        sigmas = torch.tensor((1 - alphas_cumprod) / alphas_cumprod, dtype=torch.float32) ** 0.5
        return torch.flip(sigmas, (0,))


# ----------------------------------------------------------------------------- denoiser_scaling.py
class EDMScaling:
    def __init__(self, sigma_data: float = 0.5):
        self.sigma_data = sigma_data

    def __call__(self, sigma):
        c_skip = self.sigma_data ** 2 / (sigma ** 2 + self.sigma_data ** 2)
        c_out = sigma * self.sigma_data / (sigma ** 2 + self.sigma_data ** 2) ** 0.5
        c_in = 1 / (sigma ** 2 + self.sigma_data ** 2) ** 0.5
        return c_skip, c_out, c_in, 0.25 * sigma.log()


class EpsScaling:
    """denoiser_scaling.py:29-37."""

    def __call__(self, sigma):
        return torch.ones_like(sigma), -sigma, 1 / (sigma ** 2 + 1.0) ** 0.5, sigma.clone()


class VScaling:
    def __call__(self, sigma):
        return 1.0 / (sigma ** 2 + 1.0), -sigma / (sigma ** 2 + 1.0) ** 0.5, 1.0 / (sigma ** 2 + 1.0) ** 0.5, sigma.clone()


class VScalingWithEDMcNoise:
    """denoiser_scaling.py:51-59."""

    def __call__(self, sigma):
        return (1.0 / (sigma ** 2 + 1.0), -sigma / (sigma ** 2 + 1.0) ** 0.5, 1.0 / (sigma ** 2 + 1.0) ** 0.5,
                0.25 * sigma.log())


# ----------------------------------------------------------------------------- denoiser.py
class Denoiser(nn.Module):
    """denoiser.py:13-46.  `sigma` is a per-sample vector; it is evaluated on the host."""
    HOST_MASTERS = True          # the sigma table stays an fp32 host table: engine.to(device) / .half() do not touch it

    def __init__(self, scaling_config: Dict):
        super().__init__()
        self.scaling = instantiate_from_config(scaling_config)

    def possibly_quantize_sigma(self, sigma):
        return sigma

    def possibly_quantize_c_noise(self, c_noise):
        return c_noise

    def forward(self, network, input, sigma, cond, is_modulate_step=False, is_injected_step=False, modulate_params=None,
                **additional_model_inputs):
        sigma = self.possibly_quantize_sigma(sigma.detach().float().cpu())
        c_skip, c_out, c_in, c_noise = self.scaling(sigma)
        c_noise = self.possibly_quantize_c_noise(c_noise.reshape(sigma.shape))
        t_host = c_noise.float()
        t_dev = t_host.to(input.device)      # (non_blocking measured in round 6: no gain -- 86.9 / 86.3 against 86.8 / 86.7 frames/s)
        t_dev._vidseg_host = t_host                                           # exact.ExactRunner embeds the timesteps on the host: no read-back
        net = network(ops.rows_axpby(input, c_in), t_dev, cond, is_modulate_step=is_modulate_step,
                      is_injected_step=is_injected_step, modulate_params=modulate_params, **additional_model_inputs)
        return ops.rows_axpby(net, c_out, input, c_skip)                     # net * c_out + input * c_skip


class DiscreteDenoiser(Denoiser):
    """denoiser.py:49-82: sigma snapped to the nearest of `num_idx` training sigmas, c_noise = its index."""

    def __init__(self, scaling_config, num_idx, discretization_config, do_append_zero=False, quantize_c_noise=True, flip=True):
        super().__init__(scaling_config)
        self.discretization = instantiate_from_config(discretization_config)
        self.register_buffer("sigmas", self.discretization(num_idx, do_append_zero=do_append_zero, flip=flip))
        self.quantize_c_noise = quantize_c_noise
        self.num_idx = num_idx

    def sigma_to_idx(self, sigma):
        dists = sigma - self.sigmas.cpu()[:, None]
        return dists.abs().argmin(dim=0).view(sigma.shape)

    def idx_to_sigma(self, idx):
        return self.sigmas.cpu()[idx]

    def possibly_quantize_sigma(self, sigma):
        return self.idx_to_sigma(self.sigma_to_idx(sigma))

    def possibly_quantize_c_noise(self, c_noise):
        return self.sigma_to_idx(c_noise) if self.quantize_c_noise else c_noise


# ----------------------------------------------------------------------------- guiders.py
class Guider:
    def _cat_cond(self, c, uc, keys):
        c_out = dict()
        for k in c:
            if k in keys:                                                   # uc FIRST (feature_maps[num_frames:] relies on it)
                c_out[k] = ops.window_cached(self, "_cat_" + k, (uc[k], c[k]), lambda k=k: torch.cat((uc[k], c[k]), 0))
            else:
                assert c[k] == uc[k]
                c_out[k] = c[k]
        return c_out


class VanillaCFG(Guider):
    """guiders.py:24-42."""

    def __init__(self, scale: float):
        self.scale = scale

    def __call__(self, x, sigma):
        return ops.cfg_combine(x, float(self.scale))

    def prepare_inputs(self, x, s, c, uc):
        return torch.cat([x] * 2), torch.cat([s] * 2), self._cat_cond(c, uc, ["vector", "crossattn", "concat"])


class IdentityGuider(Guider):
    def __call__(self, x, sigma):
        return x

    def prepare_inputs(self, x, s, c, uc):
        return x, s, dict(c)


class LinearPredictionGuider(Guider):
    """guiders.py:60-100: per-frame scale linspace(min, max, num_frames)."""

    def __init__(self, max_scale, num_frames, min_scale=1.0, additional_cond_keys=None):
        self.min_scale, self.max_scale, self.num_frames = min_scale, max_scale, num_frames
        self.scale = torch.linspace(min_scale, max_scale, num_frames).unsqueeze(0)
        additional_cond_keys = default(additional_cond_keys, [])
        if isinstance(additional_cond_keys, str):
            additional_cond_keys = [additional_cond_keys]
        self.additional_cond_keys = additional_cond_keys
        self._dev_scale = None

    def __call__(self, x, sigma):
        if self._dev_scale is None or self._dev_scale.device != x.device:
            self._dev_scale = self.scale.reshape(-1).float().to(x.device)
        return ops.cfg_combine(x, self._dev_scale, self.num_frames)

    def prepare_inputs(self, x, s, c, uc):
        return torch.cat([x] * 2), torch.cat([s] * 2), self._cat_cond(c, uc, ["vector", "crossattn", "concat"] + self.additional_cond_keys)


# ----------------------------------------------------------------------------- wrappers.py
class IdentityWrapper(nn.Module):
    def __init__(self, diffusion_model, compile_model: bool = False):
        super().__init__()
        self.diffusion_model = diffusion_model                                # no tracing compiler on this path

    def forward(self, *args, **kwargs):
        return self.diffusion_model(*args, **kwargs)


class OpenAIWrapper(IdentityWrapper):
    """wrappers.py:22-34."""

    def forward(self, x, t, c: dict, **kwargs):
        if "concat" in c:
            x = torch.cat((x, c["concat"].to(x.dtype)), dim=1)
        return self.diffusion_model(x, timesteps=t, context=c.get("crossattn", None), y=c.get("vector", None), **kwargs)


# ----------------------------------------------------------------------------- sampling.py
DEFAULT_GUIDER = {"target": "sgm.modules.diffusionmodules.guiders.IdentityGuider"}


class BaseDiffusionSampler:
    """sampling.py:25-82.  Sigma vectors stay on the host."""

    def __init__(self, discretization_config, num_steps=None, guider_config=None, verbose=False, device="cuda"):
        self.num_steps = num_steps
        self.discretization = instantiate_from_config(discretization_config)
        self.guider = instantiate_from_config(default(guider_config, DEFAULT_GUIDER))
        self.verbose = verbose
        self.device = device

    def prepare_sampling_loop(self, x, cond, uc=None, num_steps=None, inversion=False):
        sigmas = self.discretization(self.num_steps if num_steps is None else num_steps, device="cpu")
        if inversion:
            sigmas = sigmas.flip(0)
            sigmas[0] += 1e-8
        uc = default(uc, cond)
        x = ops.scale(x.float(), float(torch.sqrt(1.0 + sigmas[0] ** 2.0)))
        s_in = torch.ones([x.shape[0]])
        return x, s_in, sigmas, len(sigmas), cond, uc

    def denoise(self, x, denoiser, sigma, cond, uc, is_modulate_step=False, is_injected_step=False, modulate_params=None):
        denoised = denoiser(*self.guider.prepare_inputs(x, sigma, cond, uc), is_modulate_step=is_modulate_step,
                            is_injected_step=is_injected_step, modulate_params=modulate_params)
        return self.guider(denoised, sigma)

    def get_sigma_gen(self, num_sigmas):
        return range(num_sigmas - 1)


class SingleStepDiffusionSampler(BaseDiffusionSampler):
    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc, *args, **kwargs):
        raise NotImplementedError

    def euler_step(self, x, d, dt):
        raise NotImplementedError("euler_step is fused into ops.euler_update on this path")


class EDMSampler(SingleStepDiffusionSampler):
    """sampling.py:92-296 (Euler path; s_churn noise injection; is_smooth_latent through the engine's first stage)."""

    def __init__(self, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.s_churn, self.s_tmin, self.s_tmax, self.s_noise = s_churn, s_tmin, s_tmax, s_noise

    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc=None, gamma=0.0, is_modulate_step=False,
                     is_injected_step=False, modulate_params=None, is_smooth_latent=False, model=None, smooth_step_size=None):
        sigma_hat = sigma * (gamma + 1.0)
        if gamma > 0:
            eps = torch.randn_like(x) * self.s_noise
            x = ops.axpy(x, eps, float((sigma_hat[0] ** 2 - sigma[0] ** 2) ** 0.5))
        if sigma_hat.mean() < 1e-6:
            denoised = x
        else:
            denoised = self.denoise(x, denoiser, sigma_hat, cond, uc, is_modulate_step=is_modulate_step,
                                    is_injected_step=is_injected_step, modulate_params=modulate_params)
        if is_smooth_latent:                                                   # SAM:117-125: smooth in pixel space through the first stage
            if model is None:
                raise AssertionError("is_smooth_latent needs model= (the engine with decode_first_stage / encode_first_stage)")
            frames = model.decode_first_stage(denoised).contiguous()
            for frame_id in range(1, frames.shape[0] - 1):
                if (frame_id - smooth_step_size) % 3 == 0:
                    frames[frame_id] = ops.axpy(frames[frame_id - 1].contiguous(), frames[frame_id + 1].contiguous(), 1.0, 0.5)
            denoised = model.encode_first_stage(frames)
        x = ops.euler_update(x, denoised, sigma_hat, next_sigma)               # to_d + euler_step (SAM:125-131)
        return self.possible_correction_step(x, None, None, None, next_sigma, denoiser, cond, uc)

    def add_noise(self, x, cond, uc=None, num_steps=None, noise_level=0, noise=None):
        """sampling.py:133-144: (x + randn*sigma[noise_level]) / sqrt(1 + sigma_0^2).  `noise` lets tests inject the draw."""
        sigmas = self.discretization(self.num_steps if num_steps is None else num_steps, device="cpu")
        eps = torch.randn_like(x) if noise is None else noise
        return ops.axpy(x.float(), eps.float(), float(sigmas[noise_level]), 1.0 / float(torch.sqrt(1.0 + sigmas[0] ** 2.0)))

    def __call__(self, denoiser, x, cond, uc=None, num_steps=None, callback=None, img_callback=None, is_modulate=False,
                 modulate_params=None, uc_list=None, t_start=None, t_end=None, is_latent_blending=False, feature_height=None,
                 feature_width=None, is_smooth_latent=False, model=None, step_hook=None):
        """step_hook(i) (not in the reference): called with the loop index BEFORE step i's network evaluation(s) -- the feature pass
        switches the Q/K taps on for exactly the step it dumps, whatever the number of denoiser calls a step makes."""
        x, s_in, sigmas, num_sigmas, cond, uc = self.prepare_sampling_loop(x, cond, uc, num_steps)
        if is_modulate:
            if len(modulate_params["modulate_timestep_frames"]) == 0:
                modulate_timestep = modulate_params["modulate_timestep"]
            else:
                modulate_timestep = modulate_params["modulate_timestep_frames"].keys()
            is_injected_features = modulate_params["is_injected_features"]
        else:
            is_injected_features = False
        if t_start is None:
            t_start = 0
        if t_end is None:
            t_end = num_sigmas
        for i in list(self.get_sigma_gen(num_sigmas))[t_start:(t_end + 1)]:
            gamma = min(self.s_churn / (num_sigmas - 1), 2 ** 0.5 - 1) if self.s_tmin <= sigmas[i] <= self.s_tmax else 0.0
            is_modulate_step = bool(is_modulate and i in modulate_timestep)
            is_injected_step = bool(is_modulate and is_injected_features and i >= min(modulate_timestep))
            if modulate_params is not None:
                modulate_params["timestep"] = i
            if is_modulate and i in modulate_timestep:
                if len(modulate_params["modulate_timestep_frames"]) > 0:
                    modulate_params["modulate_timestep_frames_group"] = modulate_params["modulate_timestep_frames"][i]
                else:
                    modulate_params["modulate_timestep_frames_group"] = list(range(modulate_params["num_frames"]))
            if uc_list is not None:
                uc = uc_list[i]
            smooth_step = is_smooth_latent and i in (23, 24)                   # SAM:199-212: steps 23 / 24 with offsets 1 / 2
            if step_hook is not None:
                step_hook(i)
            x = self.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoiser, x, cond, uc, gamma,
                                  is_modulate_step=is_modulate_step, is_injected_step=is_injected_step,
                                  modulate_params=modulate_params, is_smooth_latent=smooth_step, model=model,
                                  smooth_step_size=(i - 22) if smooth_step else None)
            if is_latent_blending:                                             # sampling.py:229-250
                if modulate_params["latent_mask_start"] <= i <= modulate_params["latent_mask_end"]:
                    xh, xw = x.shape[-2], x.shape[-1]
                    ori_xt = load_xt(modulate_params["feature_folder"], modulate_params["exp_name"], modulate_params["timestep"],
                                     x.device).float()
                    masks = torch.stack(modulate_params["feature_masks"], dim=0).float()
                    fh = 28 if feature_height is None else feature_height
                    fw = 52 if feature_width is None else feature_width
                    x = latent_blend(x, ori_xt, masks.reshape(masks.shape[0], fh, fw))
            if callback:
                callback(i)
            if img_callback:
                if is_modulate:
                    if i >= min(modulate_timestep):
                        img_callback(x, i)
                else:
                    img_callback(x, i)
        return x

    def inversion(self, denoiser, x, cond, uc=None, num_steps=None):
        """sampling.py:264-296."""
        x, s_in, sigmas, num_sigmas, cond, uc = self.prepare_sampling_loop(x, cond, uc, num_steps, inversion=True)
        latents_list = [x]
        for i in self.get_sigma_gen(num_sigmas):
            gamma = min(self.s_churn / (num_sigmas - 1), 2 ** 0.5 - 1) if self.s_tmin <= sigmas[i] <= self.s_tmax else 0.0
            x = self.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoiser, x, cond, uc, gamma)
            latents_list.append(x)
        x = ops.scale(x, 1.0 / float(torch.sqrt(1.0 + sigmas[-1] ** 2.0)))
        latents_list[-1] = x                           # SAM:294 divides in place: the list's last entry is the rescaled tensor too
        return x, latents_list


class EulerEDMSampler(EDMSampler):
    """sampling.py:495-499."""

    def possible_correction_step(self, euler_step, x, d, dt, next_sigma, denoiser, cond, uc):
        return euler_step


def latent_blend(x, xt, masks_fhw):
    """x*m + xt*(1-m) with m nearest-upsampled to the latent grid (sampling.py:240-249)."""
    from ._lib import call, ptr, stream
    F, C, h, w = x.shape
    out = x.contiguous().clone()
    xtc, mc = xt.contiguous(), masks_fhw.contiguous()
    call("vidseg_latent_blend", ptr(out), ptr(xtc), ptr(mc), F, C, h, w, mc.shape[1], mc.shape[2], stream())
    return out
