"""tests/golden/svd_modulated_narrow.npz: the REFERENCE's SVD feature pass (dumping .pt files like the driver's callback,
svd_pipeline_vspw.py:102-142) followed by one modulated (spatial + temporal self-attention, block 8) + injected (temporal
q/k) + latent-blended pass (Step 4, :399-487) on the narrow VideoUNet.  Build-container only."""
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ref_import import import_reference  # noqa: E402
from vidseg_diffusion_amd import synthetic  # noqa: E402

T_START = 22          # 3 steps keep the fixture small; the hooks are step-independent


def main():
    import_reference()
    import sgm.modules.diffusionmodules.sampling as SAM
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    SAM.F = types.SimpleNamespace(to_pil_image=lambda t: None)      # torchvision stub: sampling.py:246 builds an unused PIL image
    torch.set_grad_enabled(False)
    net = VideoUNet(use_checkpoint=False, spatial_transformer_attn_type="softmax", **synthetic.SVD_NARROW).eval().to("cpu")
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=4321).items()})
    Fn, fh, fw = 3, 8, 8
    g = np.random.Generator(np.random.PCG64(31))
    lat = synthetic.latent_clip(Fn, 16, 16, seed=19)
    c = dict(crossattn=np.repeat(g.standard_normal((1, 1, 64)).astype(np.float32), Fn, 0),
             concat=np.repeat(g.standard_normal((1, 4, 16, 16)).astype(np.float32) * 0.5, Fn, 0),
             vector=np.repeat(g.standard_normal((1, 64)).astype(np.float32), Fn, 0))
    uc = dict(crossattn=np.zeros_like(c["crossattn"]), concat=np.zeros_like(c["concat"]), vector=c["vector"].copy())
    dd = "sgm.modules.diffusionmodules."
    den_m = Denoiser(scaling_config={"target": dd + "denoiser_scaling.VScalingWithEDMcNoise"})
    sampler = SAM.EulerEDMSampler(discretization_config={"target": dd + "discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
                                  guider_config={"target": dd + "guiders.LinearPredictionGuider",
                                                 "params": {"max_scale": 2.5, "min_scale": 1.0, "num_frames": Fn}},
                                  num_steps=25, s_churn=0, s_tmin=0, s_tmax=999, s_noise=1, device="cpu")
    model = OpenAIWrapper(net)
    extra = {"image_only_indicator": torch.zeros(2, Fn), "num_video_frames": Fn}

    def denoiser(inp, sigma, cc, is_modulate_step=False, is_injected_step=False, modulate_params=None):
        return den_m(model, inp, sigma, cc, is_modulate_step=is_modulate_step, is_injected_step=is_injected_step,
                     modulate_params=modulate_params, **extra)

    cond = {k: torch.from_numpy(v) for k, v in c.items()}
    ucond = {k: torch.from_numpy(v) for k, v in uc.items()}
    torch.manual_seed(9)
    noise = torch.randn(Fn, 4, 16, 16)
    torch.manual_seed(9)
    noised = sampler.add_noise(torch.from_numpy(lat).clone(), cond=cond, uc=ucond, num_steps=25, noise_level=T_START)
    base = tempfile.mkdtemp(prefix="vidseg_svdmod_")
    fm = os.path.join(base, "exp", "feature_maps")
    os.makedirs(fm)

    def dump_cb(xt, i):                                           # SVP:106-123; fp32 dumps (the CPU SDPA needs one dtype)
        for idx, blk in enumerate(net.output_blocks):
            if len(blk) > 1 and "SpatialVideoTransformer" in str(type(blk[1])):
                for kind, tb in (("spatial", blk[1].transformer_blocks[0]), ("temporal", blk[1].time_stack[0])):
                    torch.save(tb.attn1.k.clone(), f"{fm}/output_block_{idx}_{kind}_self_attn_k_time_{i}.pt")
                    torch.save(tb.attn1.q.clone(), f"{fm}/output_block_{idx}_{kind}_self_attn_q_time_{i}.pt")
                    torch.save(tb.attn2.k.clone(), f"{fm}/output_block_{idx}_{kind}_cross_attn_k_time_{i}.pt")
                    torch.save(tb.attn2.q.clone(), f"{fm}/output_block_{idx}_{kind}_cross_attn_q_time_{i}.pt")
        torch.save(xt.clone(), f"{fm}/xt_time_{i}.pt")

    feat_final = sampler(denoiser, noised.clone(), cond=cond, uc=ucond, img_callback=dump_cb, t_start=T_START)
    masks_np = (g.uniform(size=(Fn, fh * fw)) > 0.6).astype(np.float64)
    rec = dict(latent=lat, noise=noise.numpy(), noised=noised.numpy(), feat_final=feat_final.numpy(), masks=masks_np,
               **{f"c_{k}": v for k, v in c.items()}, **{f"uc_{k}": v for k, v in uc.items()})
    for lam in (50.0, -2000.0):            # the driver's default and a large one whose effect dominates bf16 rounding
        mp = {"feature_masks": [torch.from_numpy(m) for m in masks_np], "modulate_block_idx": [8],
              "modulate_layer_type": ["spatial", "temporal"], "modulate_attn_type": ["self_attn"], "modulate_timestep": [T_START],
              "modulate_schedule": "constant", "modulate_lambda_start": lam, "modulate_lambda_end": lam, "num_frames": Fn,
              "modulate_uc": True, "is_injected_features": True,
              "injected_feature_types": ["temporal_cross_attn_k", "temporal_cross_attn_q", "temporal_self_attn_k", "temporal_self_attn_q"],
              "injected_block_types": ["output"], "input_block_indices": [3, 4, 5, 6, 7, 8, 10, 11],
              "output_block_indices": [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11], "feature_folder": base, "exp_name": "exp",
              "injected_features_group": {}, "modulate_layer_frames": {}, "modulate_block_frames": {}, "modulate_timestep_frames": {},
              "modulate_lambda_layers": {}, "latent_mask_start": T_START, "latent_mask_end": 25}
        xs = []
        final = sampler(denoiser, noised.clone(), cond=cond, uc=ucond, img_callback=lambda xt, i: xs.append(xt.clone().numpy()),
                        is_modulate=True, modulate_params=mp, t_start=T_START, is_latent_blending=True, feature_height=fh,
                        feature_width=fw, model=None)
        tag = "pos" if lam > 0 else "neg"
        rec[f"lam_{tag}"] = np.float64(lam)
        rec[f"mod_{tag}_x_steps"] = np.stack(xs)
        rec[f"mod_{tag}_final"] = final.numpy()
    shutil.rmtree(base, ignore_errors=True)
    path = os.path.join(ROOT, "tests", "golden", "svd_modulated_narrow.npz")
    np.savez_compressed(path, **rec)
    d = np.abs(rec["mod_pos_final"] - rec["feat_final"]).mean() / np.abs(rec["feat_final"]).mean()
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; modulated vs plain rel diff", float(d))


if __name__ == "__main__":
    main()
