"""tests/golden/svd_sampler_narrow.npz: the REFERENCE's EulerEDMSampler + Denoiser(VScalingWithEDMcNoise) +
LinearPredictionGuider + OpenAIWrapper + VideoUNet (narrow) for steps 17..24.  Build-container only."""
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
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    torch.set_grad_enabled(False)
    net = VideoUNet(use_checkpoint=False, spatial_transformer_attn_type="softmax", **synthetic.SVD_NARROW).eval().to("cpu")
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=4321).items()})
    Fn = 3
    g = np.random.Generator(np.random.PCG64(8))
    lat = synthetic.latent_clip(Fn, 16, 16, seed=11)
    c = dict(crossattn=np.repeat(g.standard_normal((1, 1, 64)).astype(np.float32), Fn, 0),
             concat=np.repeat(g.standard_normal((1, 4, 16, 16)).astype(np.float32) * 0.5, Fn, 0),
             vector=np.repeat(g.standard_normal((1, 64)).astype(np.float32), Fn, 0))
    uc = dict(crossattn=np.zeros_like(c["crossattn"]), concat=np.zeros_like(c["concat"]), vector=c["vector"].copy())
    dd = "sgm.modules.diffusionmodules."
    den_m = Denoiser(scaling_config={"target": dd + "denoiser_scaling.VScalingWithEDMcNoise"})
    sampler = EulerEDMSampler(discretization_config={"target": dd + "discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
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
    torch.manual_seed(3)
    noise = torch.randn(Fn, 4, 16, 16)
    torch.manual_seed(3)
    noised = sampler.add_noise(torch.from_numpy(lat).clone(), cond=cond, uc=ucond, num_steps=25, noise_level=17)
    xs, taps = [], {}

    def cb(xt, i):
        xs.append(xt.clone().numpy())
        if i == 24:
            taps["q8"] = net.output_blocks[8][1].transformer_blocks[0].attn1.q.half().numpy()
            taps["tq8"] = net.output_blocks[8][1].time_stack[0].attn1.q.half().numpy()

    final = sampler(denoiser, noised.clone(), cond=cond, uc=ucond, img_callback=cb, t_start=17)
    rec = dict(sm_latent=lat, sm_noise=noise.numpy(), sm_noised=noised.numpy(), sm_x_steps=np.stack(xs), sm_final=final.numpy(),
               sm_sigmas=sampler.discretization(25, device="cpu").numpy(), sm_q8=taps["q8"], sm_tq8=taps["tq8"],
               **{f"c_{k}": v for k, v in c.items()}, **{f"uc_{k}": v for k, v in uc.items()})
    path = os.path.join(ROOT, "tests", "golden", "svd_sampler_narrow.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", len(xs), "steps")


if __name__ == "__main__":
    main()
