"""tests/golden/sd_modulated_narrow.npz: the REFERENCE's feature pass (dumping .pt files like the driver's callback,
sd_pipeline_vspw.py:103-139) followed by one modulated + injected + latent-blended pass (Step 4, :416-507) on the
narrow SD UNet.  Build-container only."""
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ref_import import import_reference  # noqa: E402
from vidseg_diffusion_amd import synthetic  # noqa: E402


def main():
    import_reference()
    import sgm.modules.diffusionmodules.sampling as SAM
    from sgm.modules.diffusionmodules.denoiser import DiscreteDenoiser
    from sgm.modules.diffusionmodules.openaimodel import UNetModel
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    import types
    SAM.F = types.SimpleNamespace(to_pil_image=lambda t: None)      # torchvision stub: sampling.py:246 builds an unused PIL image
    torch.set_grad_enabled(False)
    net = UNetModel(use_checkpoint=False, **synthetic.SD21_NARROW).eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1234).items()})
    Fn, fh, fw = 2, 8, 8
    g = np.random.Generator(np.random.PCG64(21))
    lat = synthetic.latent_clip(Fn, 16, 16, seed=13)
    c = g.standard_normal((Fn, 7, 64)).astype(np.float32)
    dd = "sgm.modules.diffusionmodules."
    den_m = DiscreteDenoiser(scaling_config={"target": dd + "denoiser_scaling.EpsScaling"}, num_idx=1000,
                             discretization_config={"target": dd + "discretizer.LegacyDDPMDiscretization"})
    sampler = SAM.EulerEDMSampler(discretization_config={"target": dd + "discretizer.LegacyDDPMDiscretization"},
                                  guider_config={"target": dd + "guiders.VanillaCFG", "params": {"scale": 5}}, num_steps=25,
                                  s_churn=0, s_tmin=0, s_tmax=999, s_noise=1, device="cpu")
    model = OpenAIWrapper(net)

    def denoiser(inp, sigma, cc, is_modulate_step=False, is_injected_step=False, modulate_params=None):
        return den_m(model, inp, sigma, cc, is_modulate_step=is_modulate_step, is_injected_step=is_injected_step,
                     modulate_params=modulate_params)

    cond, ucond = {"crossattn": torch.from_numpy(c)}, {"crossattn": torch.zeros(Fn, 7, 64)}
    torch.manual_seed(5)
    noise = torch.randn(Fn, 4, 16, 16)
    torch.manual_seed(5)
    noised = sampler.add_noise(torch.from_numpy(lat).clone(), cond=cond, uc=ucond, num_steps=25, noise_level=22)
    base = tempfile.mkdtemp(prefix="vidseg_mod_")
    fm = os.path.join(base, "exp", "feature_maps")
    os.makedirs(fm)

    def dump_cb(xt, i):                                           # SDP:107-120
        for idx, blk in enumerate(net.output_blocks):
            if len(blk) > 1 and "SpatialTransformer" in str(type(blk[1])):
                tb = blk[1].transformer_blocks[0]
                torch.save(tb.attn1.k.clone(), f"{fm}/output_block_{idx}_spatial_self_attn_k_time_{i}.pt")
                torch.save(tb.attn1.q.clone(), f"{fm}/output_block_{idx}_spatial_self_attn_q_time_{i}.pt")
                torch.save(tb.attn2.k.clone(), f"{fm}/output_block_{idx}_spatial_cross_attn_k_time_{i}.pt")
                torch.save(tb.attn2.q.clone(), f"{fm}/output_block_{idx}_spatial_cross_attn_q_time_{i}.pt")
        torch.save(xt.clone(), f"{fm}/xt_time_{i}.pt")

    feat_final = sampler(denoiser, noised.clone(), cond=cond, uc=ucond, img_callback=dump_cb, t_start=22)
    masks_np = (g.uniform(size=(Fn, fh * fw)) > 0.6).astype(np.float64)
    rec = dict(latent=lat, c=c, noise=noise.numpy(), noised=noised.numpy(), feat_final=feat_final.numpy(), masks=masks_np)
    for lam in (50.0, -50.0):
        mp = {"feature_masks": [torch.from_numpy(m) for m in masks_np], "modulate_block_idx": [7], "modulate_layer_type": ["spatial"],
              "modulate_attn_type": ["cross_attn"], "modulate_timestep": [22], "modulate_schedule": "constant",
              "modulate_lambda_start": lam, "modulate_lambda_end": lam, "num_frames": Fn, "modulate_uc": True,
              "is_injected_features": True,
              "injected_feature_types": ["spatial_cross_attn_k", "spatial_cross_attn_q", "spatial_self_attn_k", "spatial_self_attn_q"],
              "injected_block_types": ["output"], "input_block_indices": [3, 4, 5, 6, 7, 8, 9, 10, 11],
              "output_block_indices": [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11], "feature_folder": base, "exp_name": "exp",
              "injected_features_group": {}, "modulate_layer_frames": {}, "modulate_block_frames": {}, "modulate_timestep_frames": {},
              "modulate_lambda_layers": {}, "latent_mask_start": 22, "latent_mask_end": 23}
        xs = []
        final = sampler(denoiser, noised.clone(), cond=cond, uc=ucond, img_callback=lambda xt, i: xs.append(xt.clone().numpy()),
                        is_modulate=True, modulate_params=mp, t_start=22, is_latent_blending=True, feature_height=fh, feature_width=fw)
        tag = "pos" if lam > 0 else "neg"
        rec[f"mod_{tag}_x_steps"] = np.stack(xs)
        rec[f"mod_{tag}_final"] = final.numpy()
    shutil.rmtree(base, ignore_errors=True)
    path = os.path.join(ROOT, "tests", "golden", "sd_modulated_narrow.npz")
    np.savez_compressed(path, **rec)
    d = np.abs(rec["mod_pos_final"] - rec["feat_final"]).mean() / np.abs(rec["feat_final"]).mean()
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; modulated vs plain rel diff", float(d))


if __name__ == "__main__":
    main()
