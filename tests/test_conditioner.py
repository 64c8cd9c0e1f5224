"""The arithmetic part of the conditioner (vidseg_diffusion_amd/conditioner.py) vs the reference's own classes
(tests/golden/conditioner.npz, tools/gen_golden_conditioner.py): ConcatTimestepEmbedderND bit for bit, GeneralConditioner's
key routing / concatenation order / force-zero / (c, uc) pair on SVD's five-embedder layout (svd.yaml:38-96)."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden", "conditioner.npz")


def test_concat_timestep_embedder_matches_reference():
    from vidseg_diffusion_amd.conditioner import ConcatTimestepEmbedderND
    g = np.load(G)
    e = ConcatTimestepEmbedderND(256)
    assert np.array_equal(e(torch.from_numpy(g["fps_id"])).numpy(), g["emb_fps"])
    assert np.array_equal(e(torch.from_numpy(g["cond_aug"])).numpy(), g["emb_aug"])
    two = torch.stack([torch.from_numpy(g["fps_id"]), torch.from_numpy(g["motion"])], 1)
    assert np.array_equal(e(two).numpy(), g["emb_two"])


def test_general_conditioner_matches_reference():
    from vidseg_diffusion_amd.conditioner import GeneralConditioner
    g = np.load(G)
    cfgs = [{"target": "sgm.modules.encoders.modules.FrozenOpenCLIPImagePredictionEmbedder", "input_key": "cond_frames_without_noise"},
            {"target": "sgm.modules.encoders.modules.ConcatTimestepEmbedderND", "params": {"outdim": 256}, "input_key": "fps_id"},
            {"target": "sgm.modules.encoders.modules.ConcatTimestepEmbedderND", "params": {"outdim": 256}, "input_key": "motion_bucket_id"},
            {"target": "vidseg_diffusion_amd.conditioner.PrecomputedEmbedder", "input_key": "cond_frames"},
            {"target": "sgm.modules.encoders.modules.ConcatTimestepEmbedderND", "params": {"outdim": 256}, "input_key": "cond_aug"}]
    cond = GeneralConditioner(cfgs)
    assert [e.input_key for e in cond.embedders] == ["cond_frames_without_noise", "fps_id", "motion_bucket_id", "cond_frames", "cond_aug"]
    batch = {k[len("batch_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("batch_")}
    c, uc = cond.get_unconditional_conditioning(batch, force_uc_zero_embeddings=["cond_frames", "cond_frames_without_noise"])
    assert set(c) == {"crossattn", "vector", "concat"}
    for k in c:
        assert np.array_equal(c[k].numpy(), g["c_" + k]), k
        assert np.array_equal(uc[k].numpy(), g["uc_" + k]), k
    assert not uc["crossattn"].any() and not uc["concat"].any() and torch.equal(uc["vector"], c["vector"])


def test_video_prediction_embedder_layout():
    """(b t) c h w -> b () (t c) h w -> (b n_copies) (t c) h w with the posterior mode * scale_factor, on a stand-in encoder."""
    from vidseg_diffusion_amd.conditioner import VideoPredictionEmbedderWithEncoder

    class Enc:
        def moments(self, x):                                    # [B, h, w, 2z] NHWC: mean = 2x of the first channel, logvar junk
            b, _, hh, ww = x.shape
            m = x[:, :1].permute(0, 2, 3, 1).repeat(1, 1, 1, 4) * 2.0
            return torch.cat([m, torch.full_like(m, 7.0)], -1)
    emb = VideoPredictionEmbedderWithEncoder(n_cond_frames=1, n_copies=3, is_ae=True, scale_factor=0.5, encoder=Enc(),
                                             en_and_decode_n_samples_a_time=1)
    vid = torch.arange(2 * 3 * 4 * 4, dtype=torch.float32).reshape(2, 3, 4, 4)
    out = emb(vid)
    assert out.shape == (6, 4, 4, 4)
    for b in range(2):
        for cpy in range(3):
            assert torch.equal(out[b * 3 + cpy], vid[b, :1].repeat(4, 1, 1))     # 2.0 * 0.5 = 1


def test_yaml_targets_resolve_to_the_mirror():
    from vidseg_diffusion_amd import conditioner, util, vae
    assert util.get_obj_from_str("sgm.modules.GeneralConditioner") is conditioner.GeneralConditioner
    assert util.get_obj_from_str("sgm.modules.encoders.modules.ConcatTimestepEmbedderND") is conditioner.ConcatTimestepEmbedderND
    assert util.get_obj_from_str("sgm.models.autoencoder.AutoencodingEngine") is vae.AutoencodingEngine
    assert util.get_obj_from_str("sgm.modules.autoencoding.temporal_ae.VideoDecoder") is vae.VideoDecoder
    cond = util.instantiate_from_config({"target": "sgm.modules.GeneralConditioner", "params": {"emb_models": [
        {"target": "sgm.modules.encoders.modules.ConcatTimestepEmbedderND", "params": {"outdim": 8}, "input_key": "fps_id"}]}})
    assert cond({"fps_id": torch.tensor([1.0, 2.0])})["vector"].shape == (2, 8)


# a narrow OpenCLIP config in open_clip's model_config schema (the `arch` parameter of the embedders accepts a name or such a dict)
NARROW_CLIP = {"embed_dim": 64, "text": {"context_length": 77, "vocab_size": 49408, "width": 64, "heads": 1, "layers": 2},
               "vision": {"image_size": 28, "patch_size": 14, "width": 128, "head_width": 64, "layers": 1, "mlp_ratio": 4.0}}


def _narrow_model_config():
    """The schema of configs/inference/sd_2_1.yaml (own text, narrow sizes): every `target:` is the reference's dotted path."""
    dd = "sgm.modules.diffusionmodules."
    vae = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
               attn_resolutions=[], dropout=0.0)
    return {"model": {"target": "sgm.models.diffusion.DiffusionEngine", "params": {
        "scale_factor": 0.18215, "disable_first_stage_autocast": True,
        "denoiser_config": {"target": dd + "denoiser.DiscreteDenoiser", "params": {
            "num_idx": 1000, "scaling_config": {"target": dd + "denoiser_scaling.EpsScaling"},
            "discretization_config": {"target": dd + "discretizer.LegacyDDPMDiscretization"}}},
        "network_config": {"target": dd + "openaimodel.UNetModel", "params": dict(
            use_checkpoint=True, in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1], num_res_blocks=2,
            channel_mult=[1, 2, 4, 4], num_head_channels=64, use_linear_in_transformer=True, transformer_depth=1, context_dim=64)},
        "conditioner_config": {"target": "sgm.modules.GeneralConditioner", "params": {"emb_models": [
            {"is_trainable": False, "input_key": "txt", "target": "sgm.modules.encoders.modules.FrozenOpenCLIPEmbedder",
             "params": {"freeze": True, "layer": "penultimate", "arch": NARROW_CLIP}}]}},
        "first_stage_config": {"target": "sgm.models.autoencoder.AutoencoderKL", "params": {
            "embed_dim": 4, "monitor": "val/rec_loss", "ddconfig": vae, "lossconfig": {"target": "torch.nn.Identity"}}},
        "sampler_config": {"target": dd + "sampling.EulerEDMSampler", "params": {
            "num_steps": 25, "discretization_config": {"target": dd + "discretizer.LegacyDDPMDiscretization"},
            "guider_config": {"target": dd + "guiders.VanillaCFG", "params": {"scale": 5.0}}}}}}}


def test_diffusion_engine_from_reference_style_config():
    """The drivers' `instantiate_from_config(config.model)` on a config written in the reference's schema: every target resolves to
    this package, and the attribute protocol the drivers poke is there."""
    from vidseg_diffusion_amd import conditioner, engine, sampling, unet, util, vae
    cfg = _narrow_model_config()
    eng = util.instantiate_from_config(cfg["model"])
    assert isinstance(eng, engine.DiffusionEngine) and isinstance(eng.model, sampling.OpenAIWrapper)
    assert isinstance(eng.model.diffusion_model, unet.UNetModel) and isinstance(eng.denoiser, sampling.DiscreteDenoiser)
    assert isinstance(eng.sampler, sampling.EulerEDMSampler) and isinstance(eng.first_stage_model, vae.AutoencoderKL)
    assert isinstance(eng.conditioner, conditioner.GeneralConditioner) and eng.conditioner.embedders[0].input_key == "txt"
    assert eng.scale_factor == 0.18215 and eng.en_and_decode_n_samples_a_time is None and eng.video is False
    assert len(list(eng.model.diffusion_model.output_blocks)) == 12 and callable(eng.encode_first_stage) and callable(eng.decode_first_stage)
    # checkpoint-style keys are routed by prefix; the OpenCLIP tower's keys reach the text tower, strangers are reported
    from vidseg_diffusion_amd import openclip
    assert isinstance(eng.conditioner.embedders[0], openclip.FrozenOpenCLIPEmbedder) and eng.conditioner.embedders[0].layer == "penultimate"
    sd = {"model.diffusion_model." + k: torch.zeros(v.shape) for k, v in list(eng.model.diffusion_model.state_dict().items())[:3]}
    sd["conditioner.embedders.0.model.ln_final.weight"] = torch.full((64,), 2.0)
    sd["something.else"] = torch.zeros(1)
    missing, unexpected = eng.load_state_dict(sd)
    assert unexpected == ["something.else"] and len(missing) > 0
    assert all(k.startswith("model.diffusion_model.") or k.startswith("conditioner.embedders.0.model.") for k in missing)
    assert "conditioner.embedders.0.model.ln_final.weight" not in missing and "conditioner.embedders.0.model.ln_final.bias" in missing
    assert torch.equal(eng.conditioner.embedders[0].model.ln_final.weight, torch.full((64,), 2.0))
    assert engine.engine_from_config(cfg).scale_factor == 0.18215


def test_engine_follows_the_drivers_to_eval_calls():
    """`instantiate_from_config(config.model).to(device).eval()` (svd_pipeline_vspw.py:566-569) and `.half()`: the engine's parameters are
    host masters -- still meta before a checkpoint arrives --, so moving / casting the module is a no-op that hands the engine back."""
    from vidseg_diffusion_amd import util
    eng = util.instantiate_from_config(_narrow_model_config()["model"])
    assert eng.to("cuda").eval() is eng and eng.half() is eng and eng.cuda() is eng
    assert any(p.is_meta for p in eng.parameters())                      # nothing was materialised or moved


def test_sd21_checkpoint_layout_fills_the_text_tower():
    """the checkpoint the SD driver loads (v2-1_512-ema-pruned.safetensors, sd_pipeline_vspw.py:663) is in the LDM layout:
    `model.diffusion_model.*`, `first_stage_model.*` and the text encoder under `cond_stage_model.model.*` (open_clip names, with the
    `attn_mask` buffer).  Those keys must reach the first FrozenOpenCLIPEmbedder's tower -- every parameter of it, nothing left on meta."""
    from vidseg_diffusion_amd import util
    eng = util.instantiate_from_config(_narrow_model_config()["model"])
    tower = eng.conditioner.embedders[0].model
    g = torch.Generator().manual_seed(3)
    sd = {"cond_stage_model.model." + k: torch.randn(v.shape, generator=g) for k, v in tower.state_dict().items()}
    sd["cond_stage_model.model.attn_mask"] = torch.zeros(77, 77)
    sd.update({"model.diffusion_model." + k: torch.zeros(v.shape) for k, v in eng.model.diffusion_model.state_dict().items()})
    sd.update({"first_stage_model." + k: torch.zeros(v.shape) for k, v in eng.first_stage_model.state_dict().items()})
    sd["model_ema.decay"] = torch.zeros(1)
    missing, unexpected = eng.load_state_dict(sd)
    assert missing == [] and unexpected == ["model_ema.decay"]
    assert not any(p.is_meta for p in tower.parameters())
    assert torch.equal(tower.ln_final.weight, sd["cond_stage_model.model.ln_final.weight"])
    assert torch.equal(tower.transformer.resblocks[1].mlp.c_fc.weight, sd["cond_stage_model.model.transformer.resblocks.1.mlp.c_fc.weight"])
    # a checkpoint that ALSO carries the embedder's own keys keeps those; the legacy copy is then reported, not loaded
    eng2 = util.instantiate_from_config(_narrow_model_config()["model"])
    sd2 = {"cond_stage_model.model.ln_final.weight": torch.full((64,), 5.0), "conditioner.embedders.0.model.ln_final.weight": torch.full((64,), 2.0)}
    _, unexpected2 = eng2.load_state_dict(sd2)
    assert unexpected2 == ["cond_stage_model.model.ln_final.weight"]
    assert torch.equal(eng2.conditioner.embedders[0].model.ln_final.weight, torch.full((64,), 2.0))


def test_engine_to_moves_ordinary_torch_embedders_only():
    """`.to()` / `.half()` on the engine leave the host-master modules (UNet, first stage, OpenCLIP towers) alone but must
    still reach an ordinary torch embedder that GeneralConditioner built through instantiate_from_config and that owns real parameters."""
    from vidseg_diffusion_amd import util
    cfg = _narrow_model_config()
    cfg["model"]["params"]["conditioner_config"]["params"]["emb_models"].append(
        {"is_trainable": False, "input_key": "vec", "ucg_rate": 0.0, "target": "tests_plain_embedder.PlainEmbedder", "params": {"dim": 8}})
    import sys
    import types
    from vidseg_diffusion_amd.conditioner import AbstractEmbModel

    class PlainEmbedder(AbstractEmbModel):
        def __init__(self, dim):
            super().__init__()
            self.proj = torch.nn.Linear(dim, dim)
            self.register_buffer("scale", torch.ones(dim))

        def forward(self, x):
            return self.proj(x) * self.scale

    mod = types.ModuleType("tests_plain_embedder")
    mod.PlainEmbedder = PlainEmbedder
    sys.modules["tests_plain_embedder"] = mod
    try:
        eng = util.instantiate_from_config(cfg["model"])
    finally:
        del sys.modules["tests_plain_embedder"]
    plain = eng.conditioner.embedders[1]
    assert plain.proj.weight.dtype == torch.float32
    assert eng.double() is eng
    assert plain.proj.weight.dtype == torch.float64 and plain.proj.bias.dtype == torch.float64 and plain.scale.dtype == torch.float64
    assert any(p.is_meta for p in eng.model.diffusion_model.parameters())              # the host masters were not touched
    assert all(p.is_meta for p in eng.conditioner.embedders[0].model.parameters())
    assert eng.float() is eng and plain.proj.weight.dtype == torch.float32


@pytest.mark.parametrize("name", ["sd_2_1", "svd"])
def test_the_reference_yaml_files_build_the_engine(name):
    """The reference's own configs/inference/{sd_2_1,svd}.yaml (read-only, build container only: skipped where /root/reference is absent)
    with the drivers' patches (svd_pipeline_vspw.py:556-565: init_device, num_steps, guider num_frames) go through instantiate_from_config
    unchanged: every `target:` resolves to this package, at full size, nothing materialised (parameters on the meta device)."""
    path = f"/root/reference/configs/inference/{name}.yaml"
    if not os.path.exists(path):
        pytest.skip("the reference tree is not on this box")
    yaml = pytest.importorskip("yaml")
    from vidseg_diffusion_amd import engine, openclip, unet, util, video_unet
    with open(path) as fh:
        cfg = yaml.safe_load(fh)["model"]
    p = cfg["params"]
    p.pop("ckpt_path", None)
    if name == "svd":
        p["conditioner_config"]["params"]["emb_models"][0]["params"]["open_clip_embedding_config"]["params"]["init_device"] = "cuda"
        p["sampler_config"]["params"]["num_steps"] = 25
        p["sampler_config"]["params"]["guider_config"]["params"]["num_frames"] = 14
    eng = util.instantiate_from_config(cfg).to("cuda").eval()
    assert isinstance(eng, engine.DiffusionEngine)
    net = eng.model.diffusion_model
    if name == "svd":
        assert isinstance(net, video_unet.VideoUNet) and eng.video
        assert isinstance(eng.conditioner.embedders[0], openclip.FrozenOpenCLIPImagePredictionEmbedder) and len(eng.conditioner.embedders) == 5
        assert eng.conditioner.embedders[0].open_clip.model.visual.cfg["width"] == 1280
    else:
        assert isinstance(net, unet.UNetModel) and not eng.video
        e0 = eng.conditioner.embedders[0]
        assert isinstance(e0, openclip.FrozenOpenCLIPEmbedder) and e0.layer == "penultimate" and e0.model.cfg["layers"] == 24
    assert sum(p.numel() for p in net.parameters()) > 8e8


def _narrow_svd_config():
    """The schema of configs/inference/svd.yaml (own text, narrow sizes)."""
    dd = "sgm.modules.diffusionmodules."
    ae = dict(attn_type="vanilla", double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    emb = "sgm.modules.encoders.modules."
    cfg = {"target": "sgm.models.diffusion.DiffusionEngine", "params": {
        "scale_factor": 0.18215, "disable_first_stage_autocast": True,
        "denoiser_config": {"target": dd + "denoiser.Denoiser", "params": {"scaling_config": {"target": dd + "denoiser_scaling.VScalingWithEDMcNoise"}}},
        "network_config": {"target": dd + "video_model.VideoUNet", "params": dict(
            adm_in_channels=96, num_classes="sequential", use_checkpoint=True, in_channels=8, out_channels=4, model_channels=64,
            attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_head_channels=64, use_linear_in_transformer=True,
            transformer_depth=1, context_dim=64, spatial_transformer_attn_type="softmax-xformers", extra_ff_mix_layer=True,
            use_spatial_context=True, merge_strategy="learned_with_images", video_kernel_size=[3, 1, 1])},
        "conditioner_config": {"target": "sgm.modules.GeneralConditioner", "params": {"emb_models": [
            {"is_trainable": False, "input_key": "cond_frames_without_noise", "target": emb + "FrozenOpenCLIPImagePredictionEmbedder",
             "params": {"n_cond_frames": 1, "n_copies": 1, "open_clip_embedding_config": {"target": emb + "FrozenOpenCLIPImageEmbedder",
                                                                                    "params": {"freeze": True, "arch": NARROW_CLIP}}}},
            {"input_key": "fps_id", "is_trainable": False, "target": emb + "ConcatTimestepEmbedderND", "params": {"outdim": 32}},
            {"input_key": "motion_bucket_id", "is_trainable": False, "target": emb + "ConcatTimestepEmbedderND", "params": {"outdim": 32}},
            {"input_key": "cond_frames", "is_trainable": False, "target": emb + "VideoPredictionEmbedderWithEncoder", "params": {
                "disable_encoder_autocast": True, "n_cond_frames": 1, "n_copies": 1, "is_ae": True,
                "encoder_config": {"target": "sgm.models.autoencoder.AutoencoderKLModeOnly", "params": {
                    "embed_dim": 4, "monitor": "val/rec_loss", "ddconfig": dict(ae, attn_type="vanilla-xformers"), "lossconfig": {"target": "torch.nn.Identity"}}}}},
            {"input_key": "cond_aug", "is_trainable": False, "target": emb + "ConcatTimestepEmbedderND", "params": {"outdim": 32}}]}},
        "first_stage_config": {"target": "sgm.models.autoencoder.AutoencodingEngine", "params": {
            "loss_config": {"target": "torch.nn.Identity"},
            "regularizer_config": {"target": "sgm.modules.autoencoding.regularizers.DiagonalGaussianRegularizer"},
            "encoder_config": {"target": dd + "model.Encoder", "params": ae},
            "decoder_config": {"target": "sgm.modules.autoencoding.temporal_ae.VideoDecoder", "params": dict(ae, video_kernel_size=[3, 1, 1])}}},
        "sampler_config": {"target": dd + "sampling.EulerEDMSampler", "params": {
            "num_steps": 25, "discretization_config": {"target": dd + "discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
            "guider_config": {"target": dd + "guiders.LinearPredictionGuider", "params": {"max_scale": 2.5, "min_scale": 1.0, "num_frames": 14}}}}}}
    return cfg


def test_svd_style_config_builds_the_video_engine():
    """The schema of configs/inference/svd.yaml (own text, narrow sizes): VideoUNet, v-prediction denoiser, the five-embedder
    conditioner, AutoencodingEngine with the VideoDecoder, LinearPredictionGuider."""
    from vidseg_diffusion_amd import conditioner, sampling, util, vae, video_unet
    cfg = _narrow_svd_config()
    eng = util.instantiate_from_config(cfg)
    assert eng.video and isinstance(eng.model.diffusion_model, video_unet.VideoUNet) and isinstance(eng.denoiser, sampling.Denoiser)
    assert isinstance(eng.first_stage_model, vae.AutoencodingEngine) and eng.first_stage_model.decoder.video
    assert isinstance(eng.sampler.guider, sampling.LinearPredictionGuider)
    kinds = [type(e).__name__ for e in eng.conditioner.embedders]
    assert kinds == ["FrozenOpenCLIPImagePredictionEmbedder", "ConcatTimestepEmbedderND", "ConcatTimestepEmbedderND", "VideoPredictionEmbedderWithEncoder",
                     "ConcatTimestepEmbedderND"]
    assert isinstance(eng.conditioner.embedders[3].encoder, vae.AutoencoderKL)
    # the vector conditioning the video UNet's label_emb takes: three 32-wide sinusoids = adm_in_channels 96
    b = {"fps_id": torch.full((3,), 6.0), "motion_bucket_id": torch.full((3,), 127.0), "cond_aug": torch.full((3,), 0.02)}
    vec = torch.cat([eng.conditioner.embedders[i](b[k]) for i, k in ((1, "fps_id"), (2, "motion_bucket_id"), (4, "cond_aug"))], 1)
    assert vec.shape == (3, 96)


def _prefixed_checkpoint(eng, seed=3):
    """A synthetic checkpoint with the released files' key layout for every parameter the engine owns."""
    from vidseg_diffusion_amd import synthetic
    sd = {}
    for prefix, mod in (("model.diffusion_model.", eng.model.diffusion_model), ("first_stage_model.", eng.first_stage_model)):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        sd.update({prefix + k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=seed).items()})
    for i, emb in enumerate(eng.conditioner.embedders):
        if hasattr(emb, "encoder"):
            shapes = {k: tuple(v.shape) for k, v in emb.encoder.state_dict().items()}
            sd.update({f"conditioner.embedders.{i}.encoder.{k}": torch.from_numpy(v)
                       for k, v in synthetic.fill_state_dict(shapes, seed=seed + 1).items()})
    from vidseg_diffusion_amd import openclip
    e0 = eng.conditioner.embedders[0]                                             # the OpenCLIP tower (text: SD, image: SVD)
    assert isinstance(e0, (openclip.FrozenOpenCLIPEmbedder, openclip.FrozenOpenCLIPImagePredictionEmbedder))
    shapes = {k: tuple(v.shape) for k, v in e0.state_dict().items()}
    sd.update({"conditioner.embedders.0." + k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=seed + 2).items()})
    if isinstance(e0, openclip.FrozenOpenCLIPImagePredictionEmbedder):
        # what `del model.transformer` (modules.py:596) leaves of the text half in the released SVD checkpoints: not the image tower's, not reported
        sd["conditioner.embedders.0.open_clip.model.positional_embedding"] = torch.zeros(77, 64)
        sd["conditioner.embedders.0.open_clip.model.logit_scale"] = torch.zeros(())
    return sd


@pytest.mark.parametrize("kind", ["sd", "svd"])
def test_ckpt_path_restores_every_parameter(tmp_path, kind):
    """sgm/models/diffusion.py:85-101 -- `ckpt_path=` in the config (svd.yaml sets it): the raw checkpoint keys
    (`model.diffusion_model.*`, `first_stage_model.*`, `conditioner.embedders.3.encoder.*`) reach their modules, nothing stays
    on the meta device, nothing is reported missing or unexpected."""
    from safetensors.torch import save_file
    from vidseg_diffusion_amd import util
    cfg = _narrow_model_config()["model"] if kind == "sd" else _narrow_svd_config()
    eng0 = util.instantiate_from_config(cfg)
    sd = _prefixed_checkpoint(eng0)
    path = str(tmp_path / f"{kind}.safetensors")
    save_file({k: v.contiguous() for k, v in sd.items()}, path)
    cfg = dict(cfg, params=dict(cfg["params"], ckpt_path=path))
    eng = util.instantiate_from_config(cfg)
    assert not [n for n, p in eng.named_parameters() if p.is_meta]
    missing, unexpected = eng.init_from_ckpt(path)
    assert missing == [] and unexpected == []
    k0 = next(iter(eng.model.diffusion_model.state_dict()))
    assert torch.equal(eng.model.diffusion_model.state_dict()[k0], sd["model.diffusion_model." + k0])
    if kind == "svd":
        k3 = next(iter(eng.conditioner.embedders[3].encoder.state_dict()))
        assert torch.equal(eng.conditioner.embedders[3].encoder.state_dict()[k3], sd["conditioner.embedders.3.encoder." + k3])
    # the .ckpt route (torch.save({"state_dict": ...})) and an unknown suffix
    p2 = str(tmp_path / f"{kind}.ckpt")
    torch.save({"state_dict": sd}, p2)
    assert util.instantiate_from_config(dict(cfg, params=dict(cfg["params"], ckpt_path=p2))).init_from_ckpt(p2) == ([], [])
    with pytest.raises(NotImplementedError):
        eng.init_from_ckpt(str(tmp_path / "weights.bin"))
