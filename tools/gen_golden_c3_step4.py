"""Generate tests/golden/c3_step4_w0.npz: Step 4 of BASELINE configs[2] ("is_refine_mask + latent blending") for ONE label of window 0
through the REFERENCE at full size.

SVD img2vid VideoUNet at full width (1524.6 M parameters), 14 frames at 576x1024 (latent 14x4x72x128), CFG batch 28, on exactly
the inputs `bench.py --config svd` builds for window 0.  svd_pipeline_vspw.py:396-487 for one label: the feature pass (Step 2, its x_t of
every step written as `xt_time_<i>.pt` like the driver's callback, :102-142), then the +lambda and the -lambda modulated sampler pass
with the SVD driver's settings -- lambda * mask added to the rows of the spatial AND temporal self-attention outputs of decoder block 8
at the modulation step (attention.py:646-663, video_attention.py:197-216), no feature injection, LATENT BLENDING with the feature
pass's x_t outside the mask after every step (sampling.py:229-250).  The mask is the label the reference itself produced for this
window: tests/golden/c3_t17_w0.npz `corrected_labels` == its smallest label (36x64 tokens = block 8's resolution).

t_start = modulate_timestep = 17, the SVD driver's own schedule (svd_pipeline_vspw.py:236-237, 598): eight Euler steps per pass, 3 x 8 = 24
CFG evaluations of the whole 28-frame batch in ONE network call each (~65 min on the build host, 16 GB).

The fixture holds, per pass, x after every step (every fourth latent row / column, fp32) with its full norm, the final latent (every
second row / column) with its norm; the mask; sha256 of every input.

    python tools/gen_golden_c3_step4.py [--threads N] [--state DIR]      # resumes from DIR pass by pass
"""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_golden_c3_window import F, LH, LW, NUM_STEPS, svd_inputs  # noqa: E402
from ref_import import import_reference  # noqa: E402
from vidseg_diffusion_amd import synthetic  # noqa: E402

T_START, LAM, BLOCK = 17, 50.0, 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--state", default="/tmp/vidseg_c3_step4_state", help="directory that keeps the finished passes (the run resumes from it)")
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    import_reference()
    import sgm.modules.diffusionmodules.sampling as SAM
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    SAM.F = types.SimpleNamespace(to_pil_image=lambda t: None)      # torchvision stub: sampling.py:246 builds an unused PIL image
    torch.set_grad_enabled(False)
    t_all = time.time()
    net = VideoUNet(use_checkpoint=False, spatial_transformer_attn_type="softmax", **synthetic.SVD_FULL).eval().to("cpu")
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synthetic.fill_state_dict(shapes, seed=1234)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    del sd
    wid = 0
    lat, c, uc, noise = svd_inputs(wid)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "c3_t17_w0.npz"))
    labels = gold["corrected_labels"].astype(np.int64).reshape(F, -1)
    fh, fw = LH // 2, LW // 2
    assert labels.shape[1] == fh * fw
    label = int(np.unique(labels)[0])
    masks_np = (labels == label).astype(np.float64)                    # what pipeline.load_feature_masks hands to Step 4 (SVP:64-101)
    rec = dict(state_dict_signature=synthetic.state_dict_signature(shapes), weight_seed=1234, F=F, lat_h=LH, lat_w=LW, t_start=T_START,
               num_steps=NUM_STEPS, seed=17, window_id=wid, label=label, lam=LAM, block=BLOCK, masks=masks_np.astype(np.uint8),
               latent_sha256=synthetic.sha256_of(lat.numpy()), noise_sha256=synthetic.sha256_of(noise.numpy()),
               ctx_sha256=synthetic.sha256_of(c["crossattn"].numpy()), vector_sha256=synthetic.sha256_of(c["vector"].numpy()))

    # Memory: every attention / transformer / ResBlock module of the reference keeps its last q, k, v, attn1_out, attn2_out, ff_out ...
    # as attributes (the dumps the driver's callback reads), ~3 GB per transformer at the 28-frame CFG batch: 65 GB over the network
    # (three attempts were OOM-killed).  Step 4 with injection off reads none of them, so a forward hook drops each module's stash as
    # soon as that module returns -- no arithmetic changes; the batch-28 evaluation then peaks at 15.5 GB and the whole CFG batch goes
    # through ONE network call per evaluation (which the modulation hooks need: they index rows i and i + num_masks / [:half_hw], [half_hw:]).
    stash = ("q", "k", "v", "attn1_out", "attn2_out", "ff_out", "features_after_temporal", "video_features", "in_layers_features",
             "out_layers_features")

    def drop(mod, inp, out):
        for a in stash:
            if isinstance(getattr(mod, a, None), torch.Tensor):
                setattr(mod, a, None)

    for m in net.modules():
        m.register_forward_hook(drop)
    orig_forward = net.forward

    def forward(*a, **kw):
        out = orig_forward(*a, **kw)
        print(f"  network evaluation done ({'modulated' if kw.get('is_modulate_step') else 'plain'}, batch {out.shape[0]}), {time.time() - t_all:.0f} s", flush=True)
        return out

    net.forward = forward
    dd = "sgm.modules.diffusionmodules."
    den_m = Denoiser(scaling_config={"target": dd + "denoiser_scaling.VScalingWithEDMcNoise"})
    sampler = SAM.EulerEDMSampler(discretization_config={"target": dd + "discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
                                  guider_config={"target": dd + "guiders.LinearPredictionGuider",
                                                 "params": {"max_scale": 2.5, "min_scale": 1.0, "num_frames": F}},
                                  num_steps=NUM_STEPS, s_churn=0, s_tmin=0, s_tmax=999, s_noise=1, device="cpu")
    model = OpenAIWrapper(net)
    extra = {"image_only_indicator": torch.zeros(2, F), "num_video_frames": F}

    def denoiser(inp, sigma, cc, is_modulate_step=False, is_injected_step=False, modulate_params=None):
        return den_m(model, inp, sigma, cc, is_modulate_step=is_modulate_step, is_injected_step=is_injected_step,
                     modulate_params=modulate_params, **extra)

    torch.manual_seed(100 + wid)
    noised = sampler.add_noise(lat.clone(), cond=c, uc=uc, num_steps=NUM_STEPS, noise_level=T_START)
    sig = sampler.discretization(NUM_STEPS, device="cpu")
    assert torch.equal(noised, (lat + noise * sig[T_START]) / torch.sqrt(1.0 + sig[0] ** 2.0)), "torch.randn_like under manual_seed != Generator draw"
    # Resumable: every finished pass is kept under --state (the feature pass's xt_time_<i>.pt -- which the modulated passes read anyway --
    # and one .npz per pass), so an interrupted run continues with the next pass instead of starting over.
    base = args.state
    fm = os.path.join(base, "exp", "feature_maps")
    os.makedirs(fm, exist_ok=True)

    def pass_file(tag):
        return os.path.join(base, f"pass_{tag}.npz")

    if not os.path.exists(pass_file("feat")):
        feat_steps = []

        def dump_cb(xt, i):                                       # the driver's callback keeps x_t of every step (SVP:123)
            torch.save(xt.clone(), f"{fm}/xt_time_{i}.pt")
            feat_steps.append(xt.clone().numpy())

        feat_final = sampler(denoiser, noised.clone(), cond=c, uc=uc, img_callback=dump_cb, t_start=T_START)
        np.savez(pass_file("feat"), feat_final=feat_final.numpy(), feat_steps_sub=np.stack(feat_steps)[:, :, :, ::4, ::4].astype(np.float32),
                 feat_step_norms=np.array([np.linalg.norm(x.astype(np.float64)) for x in feat_steps]))
    pf = np.load(pass_file("feat"))
    feat_final = pf["feat_final"]
    rec.update(feat_final_sub=feat_final[:, :, ::2, ::2].astype(np.float32), feat_final_norm=np.float64(np.linalg.norm(feat_final.astype(np.float64))),
               feat_steps_sub=pf["feat_steps_sub"], feat_step_norms=pf["feat_step_norms"])
    for tag, lam in (("pos", LAM), ("neg", -LAM)):
        if not os.path.exists(pass_file(tag)):
            mp = {"feature_masks": [torch.from_numpy(m) for m in masks_np], "modulate_block_idx": [BLOCK],
                  "modulate_layer_type": ["spatial", "temporal"], "modulate_attn_type": ["self_attn"], "modulate_timestep": [T_START],
                  "modulate_schedule": "constant", "modulate_lambda_start": lam, "modulate_lambda_end": lam, "num_frames": F,
                  "modulate_uc": True, "is_injected_features": False, "injected_feature_types": None, "injected_block_types": None,
                  "input_block_indices": None, "output_block_indices": None, "feature_folder": base, "exp_name": "exp",
                  "injected_features_group": {}, "modulate_layer_frames": {}, "modulate_block_frames": {}, "modulate_timestep_frames": {},
                  "modulate_lambda_layers": {}, "latent_mask_start": T_START, "latent_mask_end": NUM_STEPS}
            xs = []
            final = sampler(denoiser, noised.clone(), cond=c, uc=uc, img_callback=lambda xt, i: xs.append(xt.clone().numpy()),
                            is_modulate=True, modulate_params=mp, t_start=T_START, is_latent_blending=True, feature_height=fh, feature_width=fw,
                            model=None)
            fin = final.numpy().astype(np.float32)
            np.savez(pass_file(tag), steps_sub=np.stack(xs)[:, :, :, ::4, ::4].astype(np.float32),
                     step_norms=np.array([np.linalg.norm(x.astype(np.float64)) for x in xs]), final_norm=np.float64(np.linalg.norm(fin.astype(np.float64))),
                     final_sub=fin[:, :, ::2, ::2], effect=np.float64(np.abs(fin - feat_final).mean() / np.abs(feat_final).mean()))
        pm = np.load(pass_file(tag))
        rec[f"mod_{tag}_steps_sub"], rec[f"mod_{tag}_step_norms"] = pm["steps_sub"], pm["step_norms"]
        rec[f"mod_{tag}_final_norm"], rec[f"mod_{tag}_final_sub"] = pm["final_norm"], pm["final_sub"]
        print(f"pass {tag}: modulated vs plain final latent, mean |difference| / mean |plain| = {float(pm['effect']):.4f}", flush=True)
    rec["versions"] = np.array([f"torch {torch.__version__}", f"numpy {np.__version__}"])
    path = os.path.join(ROOT, "tests", "golden", "c3_step4_w0.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB in", f"{time.time() - t_all:.0f} s")


if __name__ == "__main__":
    main()
