"""Generate tests/golden/sd_swan_narrow.npz: ONE window at the geometry of the reference's only shipped input, through the REFERENCE.

`input_video/swan` is 854x480, which the SD driver crops to 832x448 (sd_pipeline_vspw.py:196-214): latent 56x104, decoder grid 28x52 --
the defaults `feature_height=28, feature_width=52` of the sampler (sgm/modules/diffusionmodules/sampling.py:239-242).  Token counts per
level: 5824 / 1456 / 364 / 91 -- none a multiple of 64, the lowest level 7x13 with odd sides.  Narrow-width SD 2.1 topology
(synthetic.SD21_NARROW), 14 frames, CFG batch 28, t_start = 22 (three Euler steps), Q taps of decoder blocks 6/7/8 at step 24, the
reference's own feature_extraction_main("match_gt_mask") with the 3-block aggregate and K = 20, then "correct_low_res_mask" on block 7.
Build container only (about two minutes on 8 cores).

Stored: the reference's labels (Step 3, Step 3b) with the ten K-means restarts, the final latent and the norm of x after every step, the step-24 Q taps of
the conditional half subsampled to every 8th token / 2nd channel (full-tensor norms beside them), input hashes.

    python tools/gen_golden_swan.py
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
from gen_golden_c2_window import RestartRecorder  # noqa: E402
from ref_import import REF, import_reference  # noqa: E402
from vidseg_diffusion_amd import synthetic  # noqa: E402

F, LH, LW, K, T_START, NUM_STEPS = 14, 56, 104, 20, 22, 25
BLOCKS = ["output_block_8", "output_block_7", "output_block_6"]


def inputs():
    lat = synthetic.region_clip(F, LH, LW, num_regions=K, seed=5, amp=2.0, noise=0.05)
    g = np.random.Generator(np.random.PCG64(12))
    c = g.standard_normal((F, 77, synthetic.SD21_NARROW["context_dim"])).astype(np.float32)
    noise = torch.randn((F, 4, LH, LW), generator=torch.Generator().manual_seed(300))
    return lat, c, noise


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
    rec = dict(state_dict_signature=synthetic.state_dict_signature(shapes), weight_seed=1234, F=F, lat_h=LH, lat_w=LW, K=K, t_start=T_START,
               num_steps=NUM_STEPS, seed=17)
    lat, c, noise = inputs()
    rec.update(latent_sha256=synthetic.sha256_of(lat), c_sha256=synthetic.sha256_of(c), noise_sha256=synthetic.sha256_of(noise.numpy()))
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

    cond, ucond = {"crossattn": torch.from_numpy(c)}, {"crossattn": torch.zeros(c.shape)}
    torch.manual_seed(300)                                   # add_noise draws torch.randn_like(x) (sampling.py:139): the same stream
    noised = sampler.add_noise(torch.from_numpy(lat).clone(), cond=cond, uc=ucond, num_steps=NUM_STEPS, noise_level=T_START)
    sig = sampler.discretization(NUM_STEPS, device="cpu")
    assert torch.equal(noised, (torch.from_numpy(lat) + noise * sig[T_START]) / torch.sqrt(1.0 + sig[0] ** 2.0))
    attn = {b: net.output_blocks[b][1].transformer_blocks[0].attn1 for b in (6, 7, 8)}
    # one plain forward of frame 0's CFG pair at the geometry (the fast pin of the CPU suite; the window below takes a minute)
    fw_x = torch.cat([noised[:1], noised[:1]]) * 0.5
    fw_t = torch.full((2,), 958.0)
    fw_ctx = torch.cat([ucond["crossattn"][:1], cond["crossattn"][:1]])
    fw_out = net(fw_x, timesteps=fw_t, context=fw_ctx)
    rec.update(fw_x=fw_x.numpy(), fw_t=fw_t.numpy(), fw_out=fw_out.numpy(), fw_q7=attn[7].q.half().numpy()[:, ::4],
               fw_q7_norm=np.float64(np.linalg.norm(attn[7].q.double().numpy())))
    xs, taps = [], {}

    def cb(xt, i):
        xs.append(xt.clone().numpy())
        if i == NUM_STEPS - 1:
            for b, a in attn.items():
                assert a.q.shape == (2 * F, (LH // 2) * (LW // 2), 2 * cfg["model_channels"]), a.q.shape
                taps[b] = a.q.half()

    final = sampler(denoiser, noised.clone(), cond=cond, uc=ucond, img_callback=cb, t_start=T_START)
    rec.update(x_final=final.numpy().astype(np.float32), x_step_norms=np.array([np.linalg.norm(x.astype(np.float64)) for x in xs]))
    for b in (6, 7, 8):
        q = taps[b].numpy()
        rec[f"q{b}_sub"] = q[F:, ::8, ::2]                    # conditional half, every 8th token, every 2nd channel
        rec[f"q{b}_norm"] = np.float64(np.linalg.norm(q[F:].astype(np.float64)))
    base = tempfile.mkdtemp(prefix="vidseg_swan_")
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
        with RestartRecorder() as recd:
            ul, ref_mask, ref_fm = fe.feature_extraction_main(
                "match_gt_mask", K, T_START, ",".join(BLOCKS), exp, exp, "spatial_self_attn_q", LH // 2, LW // 2, "24", frame_name_list=names,
                base_folder=base, num_frames=F, ref_mask=None, ref_feature_map=None, ref_unique_labels=None, gt_mask_path=None)
        rec["restart_labels"] = np.stack([r[0] for r in recd.runs])
        rec["restart_inertia"] = np.array([r[1] for r in recd.runs], dtype=np.float64)
        rec["match_labels"] = np.asarray(ref_mask).astype(np.int16)
        folder = os.path.join(base, exp, "match_gt_mask", "_".join(BLOCKS) + f"_spatial_self_attn_q_masks_{K}")
        _, ref_mask2, _ = fe.feature_extraction_main(
            "correct_low_res_mask", K, T_START, "output_block_7", exp, exp, "spatial_self_attn_q", LH // 2, LW // 2, "24", frame_name_list=names,
            base_folder=base, num_frames=F, ref_mask=ref_mask, ref_feature_map=ref_fm, ref_unique_labels=ul, gt_mask_path=None,
            mask_folder=folder)
        rec["corrected_labels"] = np.asarray(ref_mask2).astype(np.int16)
    finally:
        os.chdir(cwd)
        shutil.rmtree(base, ignore_errors=True)
    import sklearn
    rec["versions"] = np.array([f"torch {torch.__version__}", f"sklearn {sklearn.__version__}", f"numpy {np.__version__}"])
    path = os.path.join(ROOT, "tests", "golden", "sd_swan_narrow.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; labels", np.bincount(rec["match_labels"].reshape(-1)), "changed by Step 3b:",
          int((rec["match_labels"] != rec["corrected_labels"]).sum()))


if __name__ == "__main__":
    main()
