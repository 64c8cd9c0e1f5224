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
from . import ops
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


def save_feature_maps(engine, store_folder, exp_name, i, xt=None, block_filter=None, pad_uncond=False):
    """ddim_sampler_callback -> save_feature_maps (sd_pipeline_vspw.py:103-139) into the FeatureStore.
    pad_uncond: the taps come from a conditional-half-only evaluation ([F, N, C]); they are stored as the second half of a
    [2F, N, C] tensor whose first (unconditional) half is never read by Steps 3-3b (FE:550-551 keeps `feature_maps[num_frames:]`)."""
    def put(name, t):
        if t is None:
            raise FE.VidsegError(f"save_feature_maps: {name} has no tap at step {i} (taps were off for this evaluation)")
        if pad_uncond:
            full = torch.empty((2 * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            full[t.shape[0]:] = t
            t = full
        FE.FeatureStore.put(store_folder, exp_name, name, t)

    blocks = engine.model.diffusion_model.output_blocks
    for idx, block in enumerate(blocks):
        # SD driver tests "SpatialTransformer", SVD driver "SpatialVideoTransformer" (SDP:112, SVP:111)
        if len(block) > 1 and ("SpatialTransformer" in str(type(block[1])) or "SpatialVideoTransformer" in str(type(block[1]))):
            if block_filter is not None and idx not in block_filter:
                continue
            tb = block[1].transformer_blocks[0]
            for an, a in (("self", tb.attn1), ("cross", tb.attn2)):
                put(f"output_block_{idx}_spatial_{an}_attn_k_time_{i}", a.k)
                put(f"output_block_{idx}_spatial_{an}_attn_q_time_{i}", a.q)
            if hasattr(block[1], "time_stack"):                                   # SVP:116-119
                ts = block[1].time_stack[0]
                for an, a in (("self", ts.attn1), ("cross", ts.attn2)):
                    put(f"output_block_{idx}_temporal_{an}_attn_k_time_{i}", a.k)
                    put(f"output_block_{idx}_temporal_{an}_attn_q_time_{i}", a.q)
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


def first_latent(engine: Engine, denoiser, latent, c, uc, num_steps, t_start, noise, inversion_type):
    """Step 1 of sample() (sd_pipeline_vspw.py:233-236, 340-345): returns (x at the first sampled step, t_start)."""
    sampler, net = engine.sampler, engine.model.diffusion_model
    if inversion_type == "add_noise":
        return sampler.add_noise(latent, cond=c, uc=uc, num_steps=num_steps, noise_level=t_start, noise=noise), t_start   # SDP:341
    if inversion_type != "inversion":
        raise ValueError(f"Unknown inversion type: {inversion_type}")                # SDP:345
    mode0 = getattr(net, "tap_mode", None)
    if mode0 is not None:                                                           # nothing reads the inversion's own Q/K
        net.tap_mode = "none"
        net._set_taps()
    try:
        x, _ = sampler.inversion(denoiser, latent, cond=c, uc=uc, num_steps=num_steps)            # SDP:343
    finally:
        if mode0 is not None:
            net.tap_mode = mode0
            net._set_taps()
    return x, 0                                                                     # SDP:235-236: t_start = 0


def feature_pass(engine: Engine, latent: torch.Tensor, c: dict, uc: dict, *, num_steps=25, t_start=22, feature_timestep="24", seed=17,
                 feature_folder="features_outputs_VSPW", exp_name="exp", noise=None, keep_all_steps=True, masks_only=False,
                 inversion_type="add_noise"):
    """Steps 1-2 of one window (sd_pipeline_vspw.py:255, 336-357): reseed, add_noise, the Euler steps of the UNet with the dump
    callback.  Everything is enqueued on the current HIP stream; returns the handle `analyse_window` needs.

    inversion_type (sd_pipeline_vspw.py:233-236, 340-345; svd_pipeline_vspw.py likewise): "add_noise" (the drivers' default) noises
    the latent to step `t_start`; "inversion" runs the EDM-form DDIM inversion `sampler.inversion` over all num_steps sigma pairs
    (sampling.py:264-296; num_steps - 1 network evaluations, the first pair skips the network) and then the feature pass from
    t_start = 0 -- num_steps more evaluations, the callback firing at every step.

    masks_only=True (opt-in, not the reference's schedule): the evaluation at `feature_timestep` -- whose only consumers in
    Steps 3-3b are the conditional half's Q taps of decoder blocks 6-8 -- runs on the conditional half alone and stops after
    output block 8; the unconditional half, blocks 9-11, the output conv, the CFG combine and the Euler update of that step (all
    dead for the masks) are skipped.  Needs keep_all_steps=False and feature_timestep = the last step; Step 4 (which reads every
    step's dumps and x_t) needs the full pass.  The taps are the same arithmetic in another fp32 summation order (the half batch
    changes split-K choices): 1.2e-3 normalised rms from the full pass at full size, i.e. the distance either has from an fp32
    evaluation -- the masks are as close to the oracle's, not bit-identical to the full pass's."""
    F, _, lh, lw = latent.shape
    ops.new_window()                                                                # per-window caches (ops.window_cached) start empty
    seed_everything(seed)                                                           # SDP:255
    sampler = engine.sampler
    denoiser = make_denoiser(engine, F)
    net = engine.model.diffusion_model
    x, t_start = first_latent(engine, denoiser, latent, c, uc, num_steps, t_start, noise, inversion_type)     # Step 1
    want = int(feature_timestep)

    def callback(xt, i):                                                            # SDP:103-105
        if i >= t_start and (keep_all_steps or i == want):
            save_feature_maps(engine, feature_folder, exp_name, i, xt=xt)

    if masks_only:
        if keep_all_steps or want != num_steps - 1 or want < t_start:
            raise ValueError("masks_only needs keep_all_steps=False and feature_timestep = the last step")
        if want > t_start:
            # nothing reads the Q / K taps of the steps before `want` (no callback): the fp16 tap copies of those evaluations are not
            # written, and the exact mode's one-key cross-attentions fold into the preceding projection (exact._NK1_IDENTITY) -- round 5;
            # the network's outputs do not depend on the taps
            mode0 = getattr(net, "tap_mode", None)
            if mode0 is not None:
                net.tap_mode = "none"
                net._set_taps()
            try:
                x = sampler(denoiser, x, cond=c, uc=uc, img_callback=None, is_modulate=False, modulate_params=None, uc_list=None,
                            t_start=t_start, t_end=want - 1, is_latent_blending=False)
            finally:
                if mode0 is not None:
                    net.tap_mode = mode0
                    net._set_taps()
        else:                                                                       # one step only: the loop's entry scaling, SAM:45-59
            x, _, _, _, _, _ = sampler.prepare_sampling_loop(x, c, uc, num_steps)
        _taps_only_eval(engine, sampler, x, c, F, num_steps, want)
        save_feature_maps(engine, feature_folder, exp_name, want, xt=None, block_filter=(6, 7, 8), pad_uncond=True)
    else:
        hook, mode0 = None, getattr(net, "tap_mode", None)
        if not keep_all_steps and mode0 is not None:
            # only step `want` is dumped: the fp16 Q/K tap copies of the other steps (1.5 GB per evaluation at config 2) would be
            # written and never read.  The network's outputs do not depend on the taps.  Keyed on the sampler's own loop index.
            def hook(i):
                net.tap_mode = mode0 if i == want else "none"
                net._set_taps()
        try:
            sampler(denoiser, x, cond=c, uc=uc, img_callback=callback, is_modulate=False, modulate_params=None, uc_list=None,
                    t_start=t_start, is_latent_blending=False, step_hook=hook)      # Step 2, SDP:357
        finally:
            if hook is not None:
                net.tap_mode = mode0
                net._set_taps()
    done = torch.cuda.Event()
    done.record(torch.cuda.current_stream())
    return dict(F=F, fh=lh // 2, fw=lw // 2, t_start=t_start, feature_timestep=feature_timestep, seed=seed, feature_folder=feature_folder,
                exp_name=exp_name, done=done)


def _taps_only_eval(engine: Engine, sampler, x, c, F, num_steps, step):
    """The network call of sampler step `step` (denoiser.py:23-46 scalings included) on the conditional half only, stopped after
    output block 8: leaves the Q/K taps of blocks <= 8 on the attention modules as [F, N, C]."""
    sigmas = sampler.discretization(sampler.num_steps if num_steps is None else num_steps, device="cpu")
    den = engine.denoiser
    sigma = den.possibly_quantize_sigma((torch.ones([F]) * sigmas[step]).float())
    _, _, c_in, c_noise = den.scaling(sigma)
    c_noise = den.possibly_quantize_c_noise(c_noise.reshape(sigma.shape))
    extra = {"image_only_indicator": torch.zeros(1, F), "num_video_frames": F} if engine.video else {}
    t_host = c_noise.float()
    t_dev = t_host.to(x.device)
    t_dev._vidseg_host = t_host
    engine.model(ops.rows_axpby(x, c_in), t_dev, c, stop_after_block=8, **extra)


def analyse_window(engine: Engine, h: dict, *, num_masks=20, is_aggre_attn=True, is_refine_mask=False, state: WindowState = None,
                   frame_names=None, gt_mask_path=None):
    """Steps 3-3b of one window (sd_pipeline_vspw.py:365-405) on the dumps `feature_pass` left in the FeatureStore.  May run
    on another stream than the feature pass (it first waits for the pass's completion event); numpy's global RandomState is
    re-seeded like the window's seed_everything did -- nothing between that call and KMeans consumes it."""
    state = state or WindowState()
    torch.cuda.current_stream().wait_event(h["done"])
    np.random.seed(h["seed"])
    F, fh, fw, t_start, feature_timestep = h["F"], h["fh"], h["fw"], h["t_start"], h["feature_timestep"]
    feature_folder, exp_name = h["feature_folder"], h["exp_name"]
    if is_aggre_attn:
        block_name = "output_block_8,output_block_7,output_block_6"                   # SDP:367-370 / SVP:351-354
    else:
        block_name = "output_block_8" if engine.video else "output_block_7"
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


def segment_window(engine: Engine, latent: torch.Tensor, c: dict, uc: dict, *, num_masks=20, num_steps=25, t_start=22,
                   feature_timestep="24", is_aggre_attn=True, is_refine_mask=False, seed=17, state: WindowState = None,
                   frame_names=None, feature_folder="features_outputs_VSPW", exp_name="exp", gt_mask_path=None, noise=None,
                   keep_all_steps=True, masks_only=False, inversion_type="add_noise"):
    """One 14-frame window: latent [F,4,h,w] fp32 (VAE output * 0.18215) -> cluster-id masks int64 [F, h/2 * w/2].
    masks_only: see feature_pass (opt-in pruning of the work Steps 3-3b never read; implies keep_all_steps=False).

    Returns (labels [F, N] int64 numpy, state) -- `state` carries ref_mask/ref_feature_map/ref_unique_labels to the
    next window exactly like the driver's loop variables."""
    h = feature_pass(engine, latent, c, uc, num_steps=num_steps, t_start=t_start, feature_timestep=feature_timestep, seed=seed,
                     feature_folder=feature_folder, exp_name=exp_name, noise=noise, keep_all_steps=keep_all_steps and not masks_only,
                     masks_only=masks_only, inversion_type=inversion_type)
    return analyse_window(engine, h, num_masks=num_masks, is_aggre_attn=is_aggre_attn, is_refine_mask=is_refine_mask, state=state,
                          frame_names=frame_names, gt_mask_path=gt_mask_path)


def hand_to_stream(lane, *things):
    """The caller allocated these tensors (dicts of tensors) on ITS stream; a lane reads them later, possibly after the caller has
    dropped them.  record_stream tells the caching allocator not to reuse their blocks before the lane's queued work is done."""
    for t in things:
        if isinstance(t, dict):
            hand_to_stream(lane, *t.values())
        elif torch.is_tensor(t) and t.is_cuda:
            t.record_stream(lane)


class WindowPipeline:
    """Software pipeline over windows: the UNet feature pass of window w+1 is enqueued BEFORE the analysis of window w runs on
    a second HIP stream, so the latency-bound K-means / 4-NN / tracking kernels (and their host polls) of one window hide
    behind the MFMA-bound feature pass of the next.  Results are identical to the sequential loop: the feature passes are
    independent, the analysis chain keeps its order (sd_pipeline_vspw.py:228-409).

    lanes > 1: the feature passes of `lanes` consecutive windows are in flight at once, each on its own HIP stream (its own
    split-K / GroupNorm scratch, ops.workspace).  The passes are the same launches on the same data -- only which kernels
    share the chip at a given moment changes: the 28 = 4 * 7 samples of a CFG window leave every power-of-two tile grid at
    7/8 of a round of 256 CUs (448 tiles of 256 rows at the 64x64 level, 224 at 32x32, ...), and the short launches of the
    16x16 / 8x8 levels fill a fraction of the chip; a second window's kernels take the idle CUs.

        pipe = WindowPipeline(engine, **analysis_kwargs)
        for ...: out = pipe.push(latent, c, uc, **feature_kwargs)   # -> labels of the window pushed `lanes` calls ago (None before)
        rest = pipe.flush()                                         # -> labels of the last window (drain() returns all of them)
    """

    def __init__(self, engine: Engine, chain=True, lanes=1, **analysis_kw):
        """chain=False treats every pushed window as the first window of its own clip (K-means every time)."""
        self.engine, self.analysis_kw, self.chain = engine, analysis_kw, chain
        self.side = torch.cuda.Stream()
        self.lanes = [torch.cuda.Stream() for _ in range(lanes)] if lanes > 1 else [None]
        self.state = WindowState()
        self.pending = []
        self.count = 0

    def _analyse(self, h):
        with torch.cuda.stream(self.side):
            labels, state = analyse_window(self.engine, h, state=self.state if self.chain else WindowState(), **self.analysis_kw)
        if self.chain:
            self.state = state
        if os.environ.get("VIDSEG_DEBUG_HASH"):                                      # run-to-run determinism hunts (tools/determinism_check.py)
            import hashlib
            import sys
            fm = state.ref_feature_map
            fh_ = hashlib.sha256(fm.detach().cpu().numpy().tobytes()).hexdigest()[:12] if torch.is_tensor(fm) else "-"
            from . import analysis as A_
            km = A_.LAST_KMEANS
            extra = ""
            if km is not None:
                extra = (f" best {km.best_restart} n_iter {list(km.all_n_iter)} inertia {[f'{v:.17g}' for v in km.all_inertia]} "
                         f"labels {[hashlib.sha256(r.tobytes()).hexdigest()[:6] for r in km.all_labels.cpu().numpy()]}")
            print(f"HASH {h['exp_name']} features {fh_} masks {hashlib.sha256(np.ascontiguousarray(labels).tobytes()).hexdigest()[:12]}{extra}",
                  file=sys.stderr, flush=True)
        FE.FeatureStore.clear(h["feature_folder"], h["exp_name"])                   # the window's dumps are no longer needed
        return labels

    def push(self, latent, c, uc, **feature_kw):
        lane = self.lanes[self.count % len(self.lanes)]
        self.count += 1
        if lane is None:
            h = feature_pass(self.engine, latent, c, uc, **feature_kw)               # enqueue first: the GPU never waits for the host
        else:
            lane.wait_stream(torch.cuda.current_stream())                            # the inputs were produced on the caller's stream
            hand_to_stream(lane, latent, c, uc, feature_kw.get("noise"))
            with torch.cuda.stream(lane):
                h = feature_pass(self.engine, latent, c, uc, **feature_kw)
        self.pending.append(h)
        return self._analyse(self.pending.pop(0)) if len(self.pending) > len(self.lanes) else None

    def drain(self):
        """Analyse everything still in flight, in order; returns the list of label maps."""
        out = [self._analyse(h) for h in self.pending]
        self.pending = []
        self.side.synchronize()
        return out

    def flush(self):
        out = self.drain()
        return out[-1] if out else None


_FEATURE_KEYS = ("num_steps", "t_start", "feature_timestep", "seed", "feature_folder", "noise", "keep_all_steps", "masks_only",
                 "inversion_type")


def segment_clip(engine, latents, c_fn, *, batch_size=14, overlap=True, lanes=1, exp_name="exp", **kw):
    """Whole clip: windows processed in order with the state chained (sd_pipeline_vspw.py:228-409).
    `c_fn(start, end)` returns (c, uc) for a window.  Returns a list of (start, end, labels).
    overlap=True runs the windows through `WindowPipeline` (analysis of window w concurrent with the feature pass of window
    w+1, `lanes` feature passes in flight; identical results); the dumps of window b live under exp_name + f"_w{b}" while
    they are needed."""
    fkw = {k: v for k, v in kw.items() if k in _FEATURE_KEYS}
    akw = {k: v for k, v in kw.items() if k not in _FEATURE_KEYS}
    slices = window_slices(latents.shape[0], batch_size)
    out = []
    if not overlap:
        state = WindowState()
        for b, (s, e) in enumerate(slices):
            c, uc = c_fn(s, e)
            labels, state = segment_window(engine, latents[s:e].contiguous(), c, uc, state=state, exp_name=f"{exp_name}_w{b}", **kw)
            out.append((s, e, labels))
        return out
    pipe = WindowPipeline(engine, lanes=lanes, **akw)
    done = []
    for b, (s, e) in enumerate(slices):
        c, uc = c_fn(s, e)
        prev = pipe.push(latents[s:e].contiguous(), c, uc, exp_name=f"{exp_name}_w{b}", **fkw)
        if prev is not None:
            done.append(prev)
    done += pipe.drain()
    return [(*slices[b], lab) for b, lab in enumerate(done)]


# ----------------------------------------------------------------------------------------------------------------------
# The drivers' conditioning block: value_dict -> get_batch -> conditioner.get_unconditional_conditioning -> per-frame repeat
# ----------------------------------------------------------------------------------------------------------------------
def get_unique_embedder_keys_from_conditioner(conditioner):
    """sd_pipeline_vspw.py:527-528 (a list in embedder order here: the set's order is arbitrary and nothing downstream depends on it)."""
    seen = []
    for e in conditioner.embedders:
        if e.input_key not in seen:
            seen.append(e.input_key)
    return seen


def sd_window_conditioning(conditioner, num_frames, *, prompt="", negative_prompt="", device="cuda"):
    """The SD driver's conditioning of one window (sd_pipeline_vspw.py:268-318 with get_batch :530-551): `txt` = [prompt] * num_frames
    through the conditioner (the OpenCLIP text tower), the unconditional half forced to zeros; T is None in that driver, so nothing is
    repeated over frames.  Returns (c, uc) with `crossattn` [num_frames, 77, context_dim] on `device`."""
    batch = {"txt": [prompt] * num_frames}
    batch_uc = {"txt": [negative_prompt] * num_frames}
    c, uc = conditioner.get_unconditional_conditioning(batch, batch_uc=batch_uc, force_uc_zero_embeddings=["txt"], force_cond_zero_embeddings=None)
    for k in c:
        if k != "crossattn":                                                 # SDP:310-313 (`math.prod(1)` rows of the other keys)
            c[k], uc[k] = c[k][:1].to(device), uc[k][:1].to(device)
    return c, uc


def svd_window_conditioning(conditioner, frames, num_frames=None, *, fps_id=6, motion_bucket_id=127, cond_aug=0.02, noise=None):
    """The SVD driver's conditioning of one window (svd_pipeline_vspw.py:263-302 with get_batch :510-541): the window's first frame is
    the conditioning image -- `cond_frames_without_noise` (OpenCLIP image tower -> crossattn) and `cond_frames` = image + cond_aug * noise
    (first-stage encoder -> concat) --, fps / motion bucket / cond_aug as the `vector` of N = num_frames rows; the unconditional half
    zeroes both image embeddings; crossattn and concat are then repeated over the frames.  frames: fp32 [T, 3, H, W] in [-1, 1] on the
    device.  `noise`: the N(0, 1) draw of SVP:279 (torch.randn_like(image) under the driver's seed) or None to draw it here.
    Returns (c, uc, additional_model_inputs)."""
    T = frames.shape[0] if num_frames is None else num_frames
    dev = frames.device
    image = frames[:1].contiguous()
    if noise is None:
        noise = torch.randn(image.shape, device=dev, dtype=image.dtype)
    value = {"motion_bucket_id": motion_bucket_id, "fps_id": fps_id, "cond_aug": cond_aug, "cond_frames_without_noise": image,
             "cond_frames": image + cond_aug * noise}
    N = (1, T)
    batch = {}
    for key in get_unique_embedder_keys_from_conditioner(conditioner):
        if key in ("fps_id", "motion_bucket_id", "cond_aug"):
            batch[key] = torch.tensor([value[key]], device=dev).repeat(N[0] * N[1])
        elif key in ("cond_frames", "cond_frames_without_noise"):
            batch[key] = value[key].expand(N[0], -1, -1, -1).contiguous()       # "1 ... -> b ...", b = N[0]
        else:
            batch[key] = value[key]
    batch_uc = {k: v.clone() for k, v in batch.items() if isinstance(v, torch.Tensor)}
    c, uc = conditioner.get_unconditional_conditioning(batch, batch_uc=batch_uc,
                                                       force_uc_zero_embeddings=["cond_frames", "cond_frames_without_noise"])
    for k in ("crossattn", "concat"):                                       # "b ... -> b t ..." then "(b t) ..." (SVP:297-301)
        for d in (c, uc):
            d[k] = d[k][:, None].expand(-1, T, *d[k].shape[1:]).reshape(-1, *d[k].shape[1:]).contiguous()
    extra = {"image_only_indicator": torch.zeros(2, T, device=dev), "num_video_frames": T}
    return c, uc, extra


# ----------------------------------------------------------------------------------------------------------------------
# Step 4 of sample(): the modulation sweep (sd_pipeline_vspw.py:412-507, svd_pipeline_vspw.py:396-487)
# ----------------------------------------------------------------------------------------------------------------------
_BLOCK_SCALE = {0: 1, 1: 1, 2: 1, 3: 2, 4: 2, 5: 2, 6: 4, 7: 4, 8: 4, 9: 8, 10: 8, 11: 8}


def load_feature_masks(masks_path, mask_id, num_frames=14, feature_timestep="24", modulate_block_idx=7, base_height=8, base_width=8,
                       frame_name_list=None, device=None):
    """sd_pipeline_vspw.py:64-101: the binary mask of label `mask_id` per frame as float64 [h*w] in {0, 1}, at the token
    resolution of the modulated decoder block (base * {1,2,4,8}).  Source: the MaskStore entry for `masks_path` (the label
    maps Step 3/3b left there) or, if absent, the reference's PNG folder.  When the mask resolution equals the target the
    PIL resize of the reference is the identity; other block groups go through the same PIL resize (host, tiny)."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    rh, rw = base_height * _BLOCK_SCALE[modulate_block_idx], base_width * _BLOCK_SCALE[modulate_block_idx]
    entry = FE.MaskStore.get(masks_path)
    out = []
    if entry is not None:
        labels = entry[0]
        labels = labels.cpu().numpy() if isinstance(labels, torch.Tensor) else np.asarray(labels)
        labels = labels.reshape(labels.shape[0], -1)
        if labels.shape[0] < num_frames:
            raise FileNotFoundError(f"{masks_path}: only {labels.shape[0]} frames of masks")
        n = labels.shape[1]
        for f in range(num_frames):
            m = (labels[f] == mask_id).astype(np.uint8) * 255
            if n != rh * rw:
                from PIL import Image
                src_w = int(round((n * rw / rh) ** 0.5))
                img = Image.fromarray(m.reshape(n // src_w, src_w)).resize((rw, rh))
                m = np.array(img).reshape(-1)
            out.append(torch.from_numpy(m / 255.0).to(device))
        return out
    from PIL import Image
    for f in range(num_frames):
        name = frame_name_list[f] if frame_name_list is not None else f
        img = Image.open(os.path.join(masks_path, f"kmeans_time_{feature_timestep}_frame_{name}", f"mask_{mask_id}.png"))
        out.append(torch.from_numpy(np.array(img.resize((rw, rh))) / 255.0).reshape(-1).to(device))
    return out


_SWEEP_STREAMS = {}


def _sweep_lanes(device, n):
    """The sweep's lane streams, created once per device (ops.workspace binds scratch per stream: fresh streams per call would leak it)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    got = _SWEEP_STREAMS.setdefault(key, [])
    while len(got) < n:
        got.append(torch.cuda.Stream(device=device))
    return got[:n]


def modulation_sweep(engine: Engine, latent, c, uc, unique_labels, masks_folder, *, t_start=22, num_steps=25, feature_timestep="24",
                     modulate_block_idx=(7,), modulate_layer_type=("spatial",), modulate_attn_type=("cross_attn",),
                     modulate_timestep=None, modulate_schedule="constant", modulate_lambda_start=50.0, modulate_lambda_end=50.0,
                     is_injected_features=True, is_latent_blending=True, feature_folder="features_outputs_VSPW", exp_name="exp",
                     frame_names=None, noise=None, seed=17, share_prefix=True, lanes=None, keep_taps=False):
    """Step 4 for one window: 2*K modulated sampler passes (+lambda then -lambda, one per label in `unique_labels`), each
    with the dumped Q/K injected, lambda*mask added to the chosen attention outputs of the chosen decoder block(s) at the
    modulation timestep(s) and, if asked, the latent blended with the feature pass's x_t outside the mask.  The feature pass
    (Step 2, `segment_window(..., keep_all_steps=True)`) must have left its dumps in the FeatureStore under
    (feature_folder, exp_name) and Step 3 its label maps under `masks_folder`.
    Returns {(sign, label): final latent fp32 [F,4,h,w]} -- what the reference hands to decode_first_stage (SDP:150-151).

    share_prefix: all 2*K passes start from the same noised latent (SDP:341 with the window's seed), so their FIRST network evaluation
    is one and the same computation up to the first modulated attention -- encoder, middle block, the decoder blocks before
    min(modulate_block_idx) and that block's ResBlock (with injection: the same dumps at the same step).  It is computed by the first
    pass and resumed by the other 2*K - 1 (exact.ExactRunner.forward; bit-identical latents, tests/test_gpu_exact.py); the later
    evaluations of a pass see that pass's own x and run in full.  Applies when the first sampled step is a modulated one (the
    drivers' default: modulate_timestep = t_start, SDP:233-234); both the exact runner and the 16-bit forward carry the fork.
    lanes: passes in flight at once, each on its own HIP stream (default: 2 for the SD network, 1 for SVD, whose launches fill the chip);
    same launches on the same data, bit-identical latents.  keep_taps: write the Q/K taps in every evaluation as a feature pass does
    (nothing reads them in Step 4; round 5's behaviour, kept for bench.py's A/B)."""
    F, _, lh, lw = latent.shape
    modulate_timestep = [t_start] if modulate_timestep is None else [int(t) for t in modulate_timestep]
    blocks = [int(b) for b in modulate_block_idx]
    video = engine.video
    if is_injected_features:                                            # SDP:420-428 / SVP:404-411
        types = (["temporal_cross_attn_k", "temporal_cross_attn_q", "temporal_self_attn_k", "temporal_self_attn_q"] if video else
                 ["spatial_cross_attn_k", "spatial_cross_attn_q", "spatial_self_attn_k", "spatial_self_attn_q"])
        inj = dict(injected_block_types=["output"], injected_feature_types=types,
                   input_block_indices=[3, 4, 5, 6, 7, 8, 10, 11] if video else [3, 4, 5, 6, 7, 8, 9, 10, 11],
                   output_block_indices=list(range(1, 12)))
    else:
        inj = dict(injected_block_types=None, injected_feature_types=None, input_block_indices=None, output_block_indices=None)
    scale = _BLOCK_SCALE[blocks[0]]
    base_h, base_w = lh // 8, lw // 8                                   # H // (F*8) of the driver (latent = image / 8)
    sampler, denoiser = engine.sampler, make_denoiser(engine, F)
    ops.new_window()
    seed_everything(seed)
    x0 = sampler.add_noise(latent, cond=c, uc=uc, num_steps=num_steps, noise_level=t_start, noise=noise)   # same start as Step 2
    # (s_churn > 0 would draw fresh noise into every pass's first input, SAM:103-109: then nothing is shared)
    shared = {"step": t_start, "fork": min(blocks), "state": None} \
        if share_prefix and t_start in modulate_timestep and float(getattr(sampler, "s_churn", 0.0)) == 0.0 else None
    out = {}

    def run(sign, mask_id):
        masks = load_feature_masks(masks_folder, mask_id, num_frames=F, feature_timestep=feature_timestep,
                                   modulate_block_idx=blocks[0], base_height=base_h, base_width=base_w,
                                   frame_name_list=frame_names, device=latent.device)
        mp = {"feature_masks": masks, "modulate_block_idx": blocks, "modulate_layer_type": list(modulate_layer_type),
              "modulate_attn_type": list(modulate_attn_type), "modulate_timestep": modulate_timestep,
              "modulate_schedule": modulate_schedule, "modulate_lambda_start": sign * modulate_lambda_start,
              "modulate_lambda_end": sign * modulate_lambda_end, "num_frames": F, "modulate_uc": True,
              "is_injected_features": is_injected_features, **inj, "feature_folder": feature_folder, "exp_name": exp_name,
              "injected_features_group": {}, "modulate_layer_frames": {}, "modulate_block_frames": {},
              "modulate_timestep_frames": {}, "modulate_lambda_layers": {}, "latent_mask_start": min(modulate_timestep),
              "latent_mask_end": num_steps if video else min(modulate_timestep) + 1}                     # SVP:467 / SDP:484
        if shared is not None:
            mp["shared_prefix"] = shared
        return sampler(denoiser, x0.clone(), cond=c, uc=uc, img_callback=None, is_modulate=True, modulate_params=mp, uc_list=None,
                       t_start=t_start, is_latent_blending=is_latent_blending, feature_height=base_h * scale, feature_width=base_w * scale)

    jobs = [(sign, int(v)) for sign in (1.0, -1.0) for v in np.asarray(unique_labels).reshape(-1)]        # SDP:436-442
    nl = max(1, int(lanes)) if lanes is not None else (1 if video else 2)   # SVD's launches fill the chip: two in flight measured slower (feature passes: -7 %)
    # nothing reads the Q/K taps of a modulated pass (the drivers hand Step 4 no dump callback, SDP:146-149): the taps are switched off for
    # the sweep -- no tap stores in the projection epilogues, and the cross-attention k | v of the un-injected blocks come from the window cache
    net = engine.model.diffusion_model
    mode0 = getattr(net, "tap_mode", None)
    switch = mode0 is not None and mode0 != "none" and not keep_taps    # keep_taps=True: the round-5 behaviour (bench.py's A/B leg)
    if switch:
        net.tap_mode = "none"
        net._set_taps()
    try:
        if nl == 1 or len(jobs) < 3:
            for sign, mask_id in jobs:
                out[(int(sign), mask_id)] = run(sign, mask_id)
            return out
        return _sweep_on_lanes(jobs, run, out, shared, nl, latent, x0, c, uc)
    finally:
        if switch:
            net.tap_mode = mode0
            net._set_taps()


def _sweep_on_lanes(jobs, run, out, shared, nl, latent, x0, c, uc):
    # lanes > 1: the passes are independent given the feature pass's dumps, so `lanes` of them are in flight at once, each on its own HIP
    # stream with its own scratch (the same launches on the same data as the sequential sweep: bit-identical latents, -3.8 % per pass).
    # The first pass runs on the caller's stream: it leaves the shared prefix and the window's cached context projections
    # (ops.window_cached), which the lanes read after waiting for it.  (Round 6 found 1-3 of 40 passes NOT bit-stable in this form and
    # traced it to k_x_attention_f32's packed-fp32 VALU ops under CU sharing -- fixed there, profiles/r06_e_sweep_lanes_race.txt.)
    main = torch.cuda.current_stream()
    out[(int(jobs[0][0]), jobs[0][1])] = run(*jobs[0])
    streams = _sweep_lanes(latent.device, nl)
    state = [] if shared is None or shared["state"] is None else [*shared["state"][0], shared["state"][1]]
    for st in streams:
        st.wait_stream(main)
        hand_to_stream(st, x0, latent, c, uc, *state)
    for j, (sign, mask_id) in enumerate(jobs[1:]):
        with torch.cuda.stream(streams[j % nl]):
            out[(int(sign), mask_id)] = run(sign, mask_id)
    for st in streams:
        main.wait_stream(st)
    for v in out.values():
        v.record_stream(main)
    return out


def segmentation_map_window(engine: Engine, first_stage_model, latent, c, uc, unique_labels, masks_folder, *, label_maps=None,
                            filter_difference=False, filter_s=0.7, scale_factor=0.18215, n_samples=None, **sweep_kw):
    """Steps 4-5 for one window, HBM-resident: the 2*K modulated sampler passes (`modulation_sweep`, SDP:416-515), each final
    latent through `decode_first_stage` (SDP:150-152), the +lambda / -lambda difference map per label and the arg-max over labels
    (`process_output.get_seg_map_main`, SDP:517-523).  Returns (uint8 [F, H, W] raw segmentation map, the sweep's latents).
    The reference writes PNG frames and JPEG difference maps in between; the JPEG round trip is the one step not reproduced."""
    from . import process_output as PO
    from .vae import decode_first_stage
    labels = [int(v) for v in np.asarray(unique_labels).reshape(-1)]
    lat = modulation_sweep(engine, latent, c, uc, labels, masks_folder, **sweep_kw)
    maps, maxima = [], []
    for lab in labels:                                                  # both signs of one label at a time: 2 decoded windows live
        pos = decode_first_stage(first_stage_model, lat[(1, lab)], scale_factor, n_samples)
        neg = decode_first_stage(first_stage_model, lat[(-1, lab)], scale_factor, n_samples)
        m, mx = PO.difference_map(pos, neg)
        maps.append(m)
        maxima.append(mx)
    maps, maxima = torch.stack(maps), torch.stack(maxima)
    weights = None
    if filter_difference:
        if label_maps is None:
            raise ValueError("filter_difference needs the Step 3 label maps [F, h, w]")
        weights = PO.mask_weights(label_maps, labels, maps.shape[-2:])
    return PO.seg_map(maps, maxima, labels, weights, filter_s), lat


def segment_clip_from_frames(engine: Engine, first_stage_model, frames, c_fn, *, scale_factor=0.18215, batch_size=14, **kw):
    """Image-in variant of `segment_clip`: frames fp32 [T, 3, H, W] in [-1, 1] on the device are encoded window by window
    with `vae.encode_first_stage` (sgm/models/diffusion.py:138-151; sd_pipeline_vspw.py:294-307) before Steps 1-3b."""
    from .vae import encode_first_stage
    state = WindowState()
    out = []
    for (s, e) in window_slices(frames.shape[0], batch_size):
        latent = encode_first_stage(first_stage_model, frames[s:e].contiguous(), scale_factor)
        c, uc = c_fn(s, e)
        labels, state = segment_window(engine, latent, c, uc, state=state, **kw)
        out.append((s, e, labels))
    return out
