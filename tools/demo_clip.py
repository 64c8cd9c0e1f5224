"""End-to-end walk through the mirrored driver on synthetic data: frames -> first stage -> Steps 1-3b (masks) -> Steps 4-5
(modulation sweep, decodes, segmentation map), the engine built from a reference-schema config like
scripts/sampling/sd_pipeline_vspw.py does.  usage: python tools/demo_clip.py [--full] [--masks K] [--frames F]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vidseg_diffusion_amd import feature_extraction as FE  # noqa: E402
from vidseg_diffusion_amd import synthetic, util  # noqa: E402
from vidseg_diffusion_amd.pipeline import segment_window, segmentation_map_window  # noqa: E402


# open_clip model-config schema, narrow: the text tower's width is the narrow UNet's context_dim
NARROW_CLIP = {"embed_dim": 64, "text": {"context_length": 77, "vocab_size": 49408, "width": 64, "heads": 1, "layers": 2},
               "vision": {"image_size": 28, "patch_size": 14, "width": 128, "head_width": 64, "layers": 1, "mlp_ratio": 4.0}}


def model_config(full):
    dd = "sgm.modules.diffusionmodules."
    mc, ctx, ch = (320, 1024, 128) if full else (64, 64, 64)
    vae = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=ch, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
               attn_resolutions=[], dropout=0.0)
    return {"target": "sgm.models.diffusion.DiffusionEngine", "params": {
        "scale_factor": 0.18215, "disable_first_stage_autocast": True,
        "denoiser_config": {"target": dd + "denoiser.DiscreteDenoiser", "params": {
            "num_idx": 1000, "scaling_config": {"target": dd + "denoiser_scaling.EpsScaling"},
            "discretization_config": {"target": dd + "discretizer.LegacyDDPMDiscretization"}}},
        "network_config": {"target": dd + "openaimodel.UNetModel", "params": dict(
            use_checkpoint=True, in_channels=4, out_channels=4, model_channels=mc, attention_resolutions=[4, 2, 1], num_res_blocks=2,
            channel_mult=[1, 2, 4, 4], num_head_channels=64, use_linear_in_transformer=True, transformer_depth=1, context_dim=ctx)},
        "conditioner_config": {"target": "sgm.modules.GeneralConditioner", "params": {"emb_models": [
            {"is_trainable": False, "input_key": "txt", "target": "sgm.modules.encoders.modules.FrozenOpenCLIPEmbedder",
             "params": {"freeze": True, "layer": "penultimate", "arch": "ViT-H-14" if full else NARROW_CLIP}}]}},
        "first_stage_config": {"target": "sgm.models.autoencoder.AutoencoderKL", "params": {"embed_dim": 4, "ddconfig": vae}},
        "sampler_config": {"target": dd + "sampling.EulerEDMSampler", "params": {
            "num_steps": 25, "discretization_config": {"target": dd + "discretizer.LegacyDDPMDiscretization"},
            "guider_config": {"target": dd + "guiders.VanillaCFG", "params": {"scale": 5.0}}}}}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="full-size SD 2.1 UNet and first stage (default: narrow)")
    ap.add_argument("--masks", type=int, default=5)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--size", type=int, default=256)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    eng = util.instantiate_from_config(model_config(a.full))
    sd = {}
    for prefix, mod, seed in (("model.diffusion_model.", eng.model.diffusion_model, 1234), ("first_stage_model.", eng.first_stage_model, 99),
                              ("conditioner.embedders.0.", eng.conditioner.embedders[0], 7)):        # the OpenCLIP text tower's keys
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        sd.update({prefix + k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=seed).items()})
    missing, unexpected = eng.load_state_dict(sd)
    assert not missing and not unexpected
    F, S = a.frames, a.size
    g = np.random.Generator(np.random.PCG64(3))
    frames = torch.from_numpy(np.clip(g.standard_normal((F, 3, S, S)) * 0.4, -1, 1).astype(np.float32)).to(dev)
    tc = time.perf_counter()
    # the drivers' conditioning (sd_pipeline_vspw.py:270-318): the empty prompt through the OpenCLIP text tower, zeros for the unconditional half
    c, uc = eng.conditioner.get_unconditional_conditioning({"txt": [""] * F}, batch_uc={"txt": [""] * F}, force_uc_zero_embeddings=["txt"])
    torch.cuda.synchronize()
    print(f"conditioner: crossattn {tuple(c['crossattn'].shape)} in {1e3 * (time.perf_counter() - tc):.1f} ms (incl. packing the tower's weights)")
    t0 = time.perf_counter()
    z = eng.encode_first_stage(frames)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    FE.FeatureStore.clear()
    FE.MaskStore.clear()
    base, exp = "/nonexistent/demo", "clip"
    labels, _ = segment_window(eng, z, c, uc, num_masks=a.masks, t_start=22, is_aggre_attn=True, is_refine_mask=True, seed=17,
                               feature_folder=base, exp_name=exp, keep_all_steps=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    uniq = np.unique(labels)
    folder = os.path.join(base, exp, "correct_low_res_mask", f"output_block_8_output_block_7_output_block_6_spatial_self_attn_q_masks_{a.masks}_corrected")
    if FE.MaskStore.get(folder) is None:
        folder = os.path.join(base, exp, "match_gt_mask", f"output_block_8_output_block_7_output_block_6_spatial_self_attn_q_masks_{a.masks}")
    seg, _ = segmentation_map_window(eng, eng.first_stage_model, z, c, uc, uniq, folder, t_start=22, feature_folder=base, exp_name=exp,
                                     seed=17, filter_difference=True, label_maps=labels.reshape(F, S // 16, S // 16))
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"frames {tuple(frames.shape)} -> latent {tuple(z.shape)} in {1e3 * (t1 - t0):.1f} ms; masks {labels.shape} ({len(uniq)} labels) in "
          f"{1e3 * (t2 - t1):.1f} ms; segmentation map {tuple(seg.shape)} ({len(torch.unique(seg))} labels) in {t3 - t2:.2f} s")


if __name__ == "__main__":
    main()
