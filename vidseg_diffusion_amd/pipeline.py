"""Steps 1-3b of the reference drivers' `sample()` on MI355X:

    scripts/sampling/sd_pipeline_vspw.py:228-409 -- windowing (last window re-anchored, :240-245), per-window
    reseed (:255, :619-623), Step 1 add_noise (:341), Step 2 feature pass with the dump callback (:103-139, :357),
    Step 3 match_gt_mask with the 3-block aggregate when is_aggre_attn (:365-385), Step 3b correct_low_res_mask
    on block 7 when is_refine_mask (:398-405), window-to-window state (ref_mask / ref_feature_map /
    ref_unique_labels, :224-226, :381-387, :401).

The VAE encode and the conditioner sit outside the built path: the engine takes
latents and conditioning tensors.  The dump hand-off is the in-HBM FeatureStore instead of `.pt` files.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from . import feature_extraction as FE
from .sampling import Denoiser, DiscreteDenoiser, EulerEDMSampler, OpenAIWrapper


def seed_everything(seed):
    """sd_pipeline_vspw.py:619-623."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    np.random.seed(seed)


def window_slices(num_frames_total: int, batch_size: int):
    """sd_pipeline_vspw.py:228-245: num_batches = len // bs + 1; the window that reaches the end is re-anchored to
    the last `batch_size` frames (a clip of exactly k*bs frames re-processes its last bs frames once more)."""
    out = []
    for batch_id in range(num_frames_total // batch_size + 1):
        start = batch_id * batch_size
        end = min((batch_id + 1) * batch_size, num_frames_total)
        if end == num_frames_total:
            start = max(end - batch_size, 0)
        out.append((start, end))
    return out


def build_sd_engine(unet, num_steps=25, scale=5.0):
    """The sampler/denoiser wiring of configs/inference/sd_2_1.yaml:6-15, :63-79 (num_steps patched by the
    driver, sd_pipeline_vspw.py:666-668)."""
    dd = "sgm.modules.diffusionmodules."
    denoiser = DiscreteDenoiser(scaling_config={"target": dd + "denoiser_scaling.EpsScaling"}, num_idx=1000,
                                discretization_config={"target": dd + "discretizer.LegacyDDPMDiscretization"})
    sampler = EulerEDMSampler(discretization_config={"target": dd + "discretizer.LegacyDDPMDiscretization"},
                              guider_config={"target": dd + "guiders.VanillaCFG", "params": {"scale": scale}},
                              num_steps=num_steps, s_churn=0, s_tmin=0, s_tmax=999, s_noise=1, device="cuda")
    return Engine(model=OpenAIWrapper(unet), denoiser=denoiser, sampler=sampler)


def build_svd_engine(video_unet, num_frames=14, num_steps=25, min_scale=1.0, max_scale=2.5):
    """configs/inference/svd.yaml:8-12 (Denoiser + VScalingWithEDMcNoise), :135-147 (EulerEDMSampler, EDMDiscretization
    sigma_max 700, LinearPredictionGuider 1.0 -> 2.5 with num_frames injected by the driver, svd_pipeline_vspw.py:563-565)."""
    dd = "sgm.modules.diffusionmodules."
    denoiser = Denoiser(scaling_config={"target": dd + "denoiser_scaling.VScalingWithEDMcNoise"})
    sampler = EulerEDMSampler(discretization_config={"target": dd + "discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
                              guider_config={"target": dd + "guiders.LinearPredictionGuider",
                                             "params": {"max_scale": max_scale, "min_scale": min_scale, "num_frames": num_frames}},
                              num_steps=num_steps, s_churn=0, s_tmin=0, s_tmax=999, s_noise=1, device="cuda")
    return Engine(model=OpenAIWrapper(video_unet), denoiser=denoiser, sampler=sampler, video=True)


@dataclass
class Engine:
    """The slice of DiffusionEngine's attribute protocol the drivers touch (sgm/models/diffusion.py)."""
    model: OpenAIWrapper
    denoiser: Denoiser
    sampler: EulerEDMSampler
    video: bool = False


@dataclass
class WindowState:
    ref_mask: Optional[np.ndarray] = None
    ref_feature_map: Optional[torch.Tensor] = None
    ref_unique_labels: Optional[np.ndarray] = None


def save_feature_maps(engine, store_folder, exp_name, i, xt=None, block_filter=None):
    """ddim_sampler_callback -> save_feature_maps (sd_pipeline_vspw.py:103-139) into the FeatureStore."""
    blocks = engine.model.diffusion_model.output_blocks
    for idx, block in enumerate(blocks):
        # SD driver tests "SpatialTransformer", SVD driver "SpatialVideoTransformer" (SDP:112, SVP:111)
        if len(block) > 1 and ("SpatialTransformer" in str(type(block[1])) or "SpatialVideoTransformer" in str(type(block[1]))):
            if block_filter is not None and idx not in block_filter:
                continue
            tb = block[1].transformer_blocks[0]
            for an, a in (("self", tb.attn1), ("cross", tb.attn2)):
                FE.FeatureStore.put(store_folder, exp_name, f"output_block_{idx}_spatial_{an}_attn_k_time_{i}", a.k)
                FE.FeatureStore.put(store_folder, exp_name, f"output_block_{idx}_spatial_{an}_attn_q_time_{i}", a.q)
            if hasattr(block[1], "time_stack"):                                   # SVP:116-119
                ts = block[1].time_stack[0]
                for an, a in (("self", ts.attn1), ("cross", ts.attn2)):
                    FE.FeatureStore.put(store_folder, exp_name, f"output_block_{idx}_temporal_{an}_attn_k_time_{i}", a.k)
                    FE.FeatureStore.put(store_folder, exp_name, f"output_block_{idx}_temporal_{an}_attn_q_time_{i}", a.q)
    if xt is not None:
        FE.FeatureStore.put(store_folder, exp_name, f"xt_time_{i}", xt)


def make_denoiser(engine: Engine, num_frames: int):
    """The driver's denoiser closure (sd_pipeline_vspw.py:324-332); SVD adds image_only_indicator / num_video_frames
    (svd_pipeline_vspw.py:307-311)."""
    extra = {"image_only_indicator": torch.zeros(2, num_frames), "num_video_frames": num_frames} if engine.video else {}

    def denoiser(inp, sigma, cc, is_modulate_step=False, is_injected_step=False, modulate_params=None):
        return engine.denoiser(engine.model, inp, sigma, cc, is_modulate_step=is_modulate_step, is_injected_step=is_injected_step,
                               modulate_params=modulate_params, **extra)
    return denoiser


def segment_window(engine: Engine, latent: torch.Tensor, c: dict, uc: dict, *, num_masks=20, num_steps=25, t_start=22,
                   feature_timestep="24", is_aggre_attn=True, is_refine_mask=False, seed=17, state: WindowState = None,
                   frame_names=None, feature_folder="features_outputs_VSPW", exp_name="exp", gt_mask_path=None, noise=None,
                   keep_all_steps=True):
    """One 14-frame window: latent [F,4,h,w] fp32 (VAE output * 0.18215) -> cluster-id masks int64 [F, h/2 * w/2].

    Returns (labels [F, N] int64 numpy, state) -- `state` carries ref_mask/ref_feature_map/ref_unique_labels to the
    next window exactly like the driver's loop variables."""
    state = state or WindowState()
    F, _, lh, lw = latent.shape
    seed_everything(seed)                                                           # SDP:255
    sampler = engine.sampler

    denoiser = make_denoiser(engine, F)

    x = sampler.add_noise(latent, cond=c, uc=uc, num_steps=num_steps, noise_level=t_start, noise=noise)   # Step 1, SDP:341
    want = int(feature_timestep)

    def callback(xt, i):                                                            # SDP:103-105
        if i >= t_start and (keep_all_steps or i == want):
            save_feature_maps(engine, feature_folder, exp_name, i, xt=xt)

    sampler(denoiser, x, cond=c, uc=uc, img_callback=callback, is_modulate=False, modulate_params=None, uc_list=None,
            t_start=t_start, is_latent_blending=False)                              # Step 2, SDP:357
    if is_aggre_attn:
        block_name = "output_block_8,output_block_7,output_block_6"                   # SDP:367-370 / SVP:351-354
    else:
        block_name = "output_block_8" if engine.video else "output_block_7"
    fh, fw = lh // 2, lw // 2                                                       # H // (F*2), SDP:374-375
    unique_labels, ref_mask, ref_fm = FE.feature_extraction_main(
        "match_gt_mask", num_masks, t_start, block_name, exp_name, exp_name, "spatial_self_attn_q", fh, fw, feature_timestep,
        frame_name_list=frame_names, base_folder=feature_folder, num_frames=F, ref_mask=state.ref_mask,
        ref_feature_map=state.ref_feature_map, ref_unique_labels=state.ref_unique_labels, gt_mask_path=gt_mask_path)
    if state.ref_unique_labels is None:
        state.ref_unique_labels = unique_labels                                     # SDP:386-387
    if is_refine_mask:                                                              # Step 3b, SDP:398-405
        folder = os.path.join(feature_folder, exp_name, "match_gt_mask",
                              "_".join(block_name.split(",")) + f"_spatial_self_attn_q_masks_{num_masks}")
        _, ref_mask, _ = FE.feature_extraction_main(
            "correct_low_res_mask", num_masks, t_start, "output_block_7", exp_name, exp_name, "spatial_self_attn_q", fh, fw,
            feature_timestep, frame_name_list=frame_names, base_folder=feature_folder, num_frames=F, ref_mask=ref_mask,
            ref_feature_map=ref_fm, ref_unique_labels=state.ref_unique_labels, gt_mask_path=gt_mask_path, mask_folder=folder)
    state.ref_mask, state.ref_feature_map = ref_mask, ref_fm
    return np.asarray(ref_mask).reshape(F, fh * fw), state


def segment_clip(engine, latents, c_fn, *, batch_size=14, **kw):
    """Whole clip: windows processed in order with the state chained (sd_pipeline_vspw.py:228-409).
    `c_fn(start, end)` returns (c, uc) for a window.  Returns a list of (start, end, labels)."""
    state = WindowState()
    out = []
    for (s, e) in window_slices(latents.shape[0], batch_size):
        c, uc = c_fn(s, e)
        labels, state = segment_window(engine, latents[s:e].contiguous(), c, uc, state=state, **kw)
        out.append((s, e, labels))
    return out
