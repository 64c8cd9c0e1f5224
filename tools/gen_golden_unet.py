"""Generate tests/golden/unet_sd_narrow.npz by running the REFERENCE's UNetModel and
EulerEDMSampler/DiscreteDenoiser/VanillaCFG/OpenAIWrapper (imported read-only from /root/reference) on a
narrow-width instance of the SD 2.1 topology with the deterministic synthetic weights of
vidseg_diffusion_amd.synthetic.fill_state_dict.  Build-container only.

    python tools/gen_golden_unet.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ref_import import import_reference  # noqa: E402
from vidseg_diffusion_amd import synthetic  # noqa: E402


def main():
    import_reference()
    from sgm.modules.diffusionmodules.openaimodel import UNetModel
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    from sgm.modules.diffusionmodules.denoiser import DiscreteDenoiser
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    torch.set_grad_enabled(False)
    cfg = dict(synthetic.SD21_NARROW)
    net = UNetModel(use_checkpoint=False, **cfg).eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synthetic.fill_state_dict(shapes, seed=1234)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    rec = dict(state_dict_signature=synthetic.state_dict_signature(shapes), weight_seed=1234)

    # --- single forward ---------------------------------------------------------------------
    g = np.random.Generator(np.random.PCG64(5))
    x = g.standard_normal((4, 4, 16, 16)).astype(np.float32)
    t = np.array([500.0, 500.0, 999.0, 3.0], dtype=np.float32)
    ctx = g.standard_normal((4, 7, 64)).astype(np.float32)
    out = net(torch.from_numpy(x), timesteps=torch.from_numpy(t), context=torch.from_numpy(ctx))
    rec.update(fw_x=x, fw_t=t, fw_ctx=ctx, fw_out=out.numpy())
    for i, blk in enumerate(net.output_blocks):
        if len(blk) > 1 and "SpatialTransformer" in str(type(blk[1])):
            tb = blk[1].transformer_blocks[0]
            for an, a in (("self", tb.attn1), ("cross", tb.attn2)):
                rec[f"fw_output_block_{i}_spatial_{an}_attn_q"] = a.q.half().numpy()
                rec[f"fw_output_block_{i}_spatial_{an}_attn_k"] = a.k.half().numpy()

    # the "mid-block spatial features" every ResBlock leaves behind (openaimodel.py:349-350, 367-368): in_layers(x) before the emb
    # add, out_layers(h) before the skip add -- NCHW, stored in fp16 for six blocks across the levels
    for name in synthetic.RESBLOCK_FEATURE_PROBES:
        rb = net.get_submodule(name)
        rec[f"fw_rb_{name}_in"] = rb.in_layers_features.half().numpy()
        rec[f"fw_rb_{name}_out"] = rb.out_layers_features.half().numpy()

    # --- sampler: add_noise + Euler steps 22..24 with CFG ------------------------------------
    Fn = 2
    lat = synthetic.latent_clip(Fn, 16, 16, seed=9)
    c = g.standard_normal((Fn, 7, 64)).astype(np.float32)
    uc = np.zeros_like(c)
    dd = "sgm.modules.diffusionmodules."
    denoiser_m = DiscreteDenoiser(scaling_config={"target": dd + "denoiser_scaling.EpsScaling"}, num_idx=1000,
                                  discretization_config={"target": dd + "discretizer.LegacyDDPMDiscretization"})
    sampler = EulerEDMSampler(discretization_config={"target": dd + "discretizer.LegacyDDPMDiscretization"},
                              guider_config={"target": dd + "guiders.VanillaCFG", "params": {"scale": 5}}, num_steps=25,
                              s_churn=0, s_tmin=0, s_tmax=999, s_noise=1, device="cpu")
    model = OpenAIWrapper(net)

    def denoiser(inp, sigma, cc, is_modulate_step=False, is_injected_step=False, modulate_params=None):
        return denoiser_m(model, inp, sigma, cc, is_modulate_step=is_modulate_step, is_injected_step=is_injected_step,
                          modulate_params=modulate_params)

    cond, ucond = {"crossattn": torch.from_numpy(c)}, {"crossattn": torch.from_numpy(uc)}
    torch.manual_seed(17)
    noise = torch.randn(Fn, 4, 16, 16)
    torch.manual_seed(17)
    noised = sampler.add_noise(torch.from_numpy(lat).clone(), cond=cond, uc=ucond, num_steps=25, noise_level=22)
    xs, taps = [], {}

    def cb(xt, i):
        xs.append(xt.clone().numpy())
        if i == 24:
            for b in (6, 7, 8):
                taps[b] = net.output_blocks[b][1].transformer_blocks[0].attn1.q.half().numpy()

    final = sampler(denoiser, noised.clone(), cond=cond, uc=ucond, img_callback=cb, t_start=22)
    rec.update(sm_latent=lat, sm_c=c, sm_noise=noise.numpy(), sm_noised=noised.numpy(), sm_x_steps=np.stack(xs),
               sm_final=final.numpy(), sm_sigmas=sampler.discretization(25, device="cpu").numpy())
    for b in (6, 7, 8):
        rec[f"sm_q_block_{b}_time_24"] = taps[b]
    # --- a3b: EDM-form "DDIM inversion" (sampling.py:264-296), then it is the feature pass from t_start = 0
    inv, lat_list = sampler.inversion(denoiser, torch.from_numpy(lat).clone(), cond=cond, uc=ucond, num_steps=25)
    rec.update(inv_final=inv.numpy(), inv_step5=lat_list[5].numpy(), inv_step24=lat_list[24].numpy())
    rec["versions"] = np.array([f"torch {torch.__version__}"])
    path = os.path.join(ROOT, "tests", "golden", "unet_sd_narrow.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; out abs mean", float(np.abs(out.numpy()).mean()))


if __name__ == "__main__":
    main()
