"""Generate tests/golden/sd_inversion_window_narrow.npz: ONE window through the REFERENCE with `inversion_type = "inversion"`
(scripts/sampling/sd_pipeline_vspw.py:233-236, 340-345, 357): EulerEDMSampler.inversion over all 25 sigma pairs
(sgm/modules/diffusionmodules/sampling.py:264-296), then the feature pass from t_start = 0 (25 CFG evaluations, the dump callback
firing at every step), then the reference's own feature_extraction_main("match_gt_mask") on the step-24 dumps of decoder blocks
8/7/6 and "correct_low_res_mask" on block 7.  Narrow-width SD 2.1 topology (synthetic.SD21_NARROW), 4 frames at 16x16 latents, K = 4.
Build container only.

    python tools/gen_golden_inversion_window.py
"""
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ref_import import REF, import_reference  # noqa: E402
from vidseg_diffusion_amd import synthetic  # noqa: E402

F, LAT, K, NUM_STEPS = 4, 16, 4, 25
BLOCKS = ["output_block_8", "output_block_7", "output_block_6"]


def main():
    fe = import_reference()
    from sgm.modules.diffusionmodules.denoiser import DiscreteDenoiser
    from sgm.modules.diffusionmodules.openaimodel import UNetModel
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    torch.set_grad_enabled(False)
    cfg = dict(synthetic.SD21_NARROW)
    net = UNetModel(use_checkpoint=False, **cfg).eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1234).items()})
    rec = dict(state_dict_signature=synthetic.state_dict_signature(shapes), weight_seed=1234, F=F, lat=LAT, K=K, num_steps=NUM_STEPS, seed=17)
    lat = synthetic.latent_clip(F, LAT, LAT, seed=21)
    g = np.random.Generator(np.random.PCG64(11))
    c = g.standard_normal((F, 7, cfg["context_dim"])).astype(np.float32)
    uc = np.zeros_like(c)
    dd = "sgm.modules.diffusionmodules."
    denoiser_m = DiscreteDenoiser(scaling_config={"target": dd + "denoiser_scaling.EpsScaling"}, num_idx=1000,
                                  discretization_config={"target": dd + "discretizer.LegacyDDPMDiscretization"})
    sampler = EulerEDMSampler(discretization_config={"target": dd + "discretizer.LegacyDDPMDiscretization"},
                              guider_config={"target": dd + "guiders.VanillaCFG", "params": {"scale": 5.0}}, num_steps=NUM_STEPS,
                              s_churn=0, s_tmin=0, s_tmax=999, s_noise=1, device="cpu")
    model = OpenAIWrapper(net)

    def denoiser(inp, sigma, cc, is_modulate_step=False, is_injected_step=False, modulate_params=None):
        return denoiser_m(model, inp, sigma, cc, is_modulate_step=is_modulate_step, is_injected_step=is_injected_step,
                          modulate_params=modulate_params)

    cond, ucond = {"crossattn": torch.from_numpy(c)}, {"crossattn": torch.from_numpy(uc)}
    latent, _ = sampler.inversion(denoiser, torch.from_numpy(lat).clone(), cond=cond, uc=ucond, num_steps=NUM_STEPS)    # SDP:343
    rec.update(latent=lat, c=c, inverted=latent.numpy().copy())
    xs, taps, dumped = {}, {}, []
    attn = {b: net.output_blocks[b][1].transformer_blocks[0].attn1 for b in (6, 7, 8)}

    def cb(xt, i):                                            # the driver dumps at every i >= t_start = 0 (SDP:103-105)
        dumped.append(i)
        if i in (0, 12, 24):
            xs[i] = xt.clone().numpy()
        if i == NUM_STEPS - 1:
            for b, a in attn.items():
                taps[b] = a.q.half()

    final = sampler(denoiser, latent.clone(), cond=cond, uc=ucond, img_callback=cb, t_start=0)                           # SDP:357 with t_start = 0
    assert dumped == list(range(NUM_STEPS))
    rec.update(x_final=final.numpy(), x_step0=xs[0], x_step12=xs[12], x_step24=xs[24], dumped_steps=np.array(dumped))
    for b in (6, 7, 8):
        rec[f"q{b}"] = taps[b].numpy()
    base = tempfile.mkdtemp(prefix="vidseg_inv_")
    exp = "exp"
    fm = os.path.join(base, exp, "feature_maps")
    os.makedirs(fm)
    for name, b in zip(BLOCKS, (8, 7, 6)):
        torch.save(taps[b], os.path.join(fm, f"{name}_spatial_self_attn_q_time_24.pt"))
    names = [f"{i:05d}" for i in range(F)]
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        np.random.seed(17)
        ul, ref_mask, ref_fm = fe.feature_extraction_main(
            "match_gt_mask", K, 0, ",".join(BLOCKS), exp, exp, "spatial_self_attn_q", LAT // 2, LAT // 2, "24", frame_name_list=names,
            base_folder=base, num_frames=F, ref_mask=None, ref_feature_map=None, ref_unique_labels=None, gt_mask_path=None)
        rec["match_labels"] = np.asarray(ref_mask).astype(np.int16)
        folder = os.path.join(base, exp, "match_gt_mask", "_".join(BLOCKS) + f"_spatial_self_attn_q_masks_{K}")
        _, ref_mask2, _ = fe.feature_extraction_main(
            "correct_low_res_mask", K, 0, "output_block_7", exp, exp, "spatial_self_attn_q", LAT // 2, LAT // 2, "24", frame_name_list=names,
            base_folder=base, num_frames=F, ref_mask=ref_mask, ref_feature_map=ref_fm, ref_unique_labels=ul, gt_mask_path=None,
            mask_folder=folder)
        rec["corrected_labels"] = np.asarray(ref_mask2).astype(np.int16)
    finally:
        os.chdir(cwd)
        shutil.rmtree(base, ignore_errors=True)
    rec["versions"] = np.array([f"torch {torch.__version__}"])
    path = os.path.join(ROOT, "tests", "golden", "sd_inversion_window_narrow.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; labels", np.bincount(rec["match_labels"].reshape(-1)))


if __name__ == "__main__":
    main()
