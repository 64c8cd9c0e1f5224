"""The conditioner's OpenCLIP towers on the HIP path (vidseg_diffusion_amd/openclip.py, csrc/clip_ops.hip) against oracle/openclip.py
(itself pinned against transformers' CLIP in tests/test_oracle_openclip.py).  fp16 build only (the towers run in the exact mode).

Bars (floating point, against the oracle's fp32 / a float64 evaluation of the same inputs):
    attention / LayerNorm / GELU / preprocess kernels      |err| <= 3e-6 * max|ref|      (fp32 arithmetic, other summation order)
    narrow towers (3-4 blocks)                                |err| <= 2e-5 * max|ref|
    full ViT-H-14 towers (23 / 32 blocks, synthetic weights)  |err| <= 2e-5 * max|ref|      (measured 2.6e-6 / 2.3e-6)
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from conftest import act_mode
from tools_openclip import fill, text_shapes, visual_shapes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NARROW = {"embed_dim": 64,
          "text": {"context_length": 77, "vocab_size": 49408, "width": 128, "heads": 2, "layers": 4},
          "vision": {"image_size": 56, "patch_size": 14, "width": 320, "head_width": 80, "layers": 3, "mlp_ratio": 4.0}}


@pytest.fixture(scope="module")
def C():
    assert torch.cuda.is_available()
    from vidseg_diffusion_amd import _lib, openclip
    _lib.lib()
    if act_mode()[0] != "f16":
        pytest.skip("the OpenCLIP towers run in the exact mode, which exists in the fp16 build only")
    return openclip


def rel(got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    return float((got - ref).abs().max() / ref.abs().max())


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal(shape).astype(np.float32)) * scale


@pytest.mark.parametrize("B,N,heads,d,causal", [(2, 77, 2, 64, True), (1, 257, 4, 80, False), (3, 17, 1, 128, False), (1, 5, 3, 4, True)])
def test_attention_kernel(C, B, N, heads, d, causal):
    W = heads * d
    qkv = rnd((B * N, 3 * W), 11, 1.5)
    got = C.attention(qkv.to(DEV), B, N, heads, causal)
    q, k, v = (t.double().reshape(B, N, heads, d).transpose(1, 2) for t in qkv.chunk(3, -1))
    s = q @ k.transpose(-1, -2) / math.sqrt(d)
    if causal:
        s = s + torch.full((N, N), float("-inf"), dtype=torch.float64).triu_(1)
    ref = (s.softmax(-1) @ v).transpose(1, 2).reshape(B * N, W)
    assert rel(got, ref) <= 3e-6


def test_attention_kernel_rejects_what_it_cannot_do(C):
    from vidseg_diffusion_amd._lib import VidsegError
    with pytest.raises(VidsegError):
        C.attention(torch.zeros((4, 3 * 132), device=DEV), 1, 4, 1, False)          # head width 132 > 128
    with pytest.raises(VidsegError):
        C.attention(torch.zeros((1100, 3 * 64), device=DEV), 1, 1100, 1, False)     # 1100 keys > 1024


def test_layernorm_and_gelu_kernels(C):
    x = rnd((300, 1280), 5, 3.0) + 0.7
    g, b = rnd((1280,), 6) * 0.1 + 1, rnd((1280,), 7) * 0.05
    got = C.layernorm(x.to(DEV), g.to(DEV), b.to(DEV))
    assert rel(got, TF.layer_norm(x.double(), (1280,), g.double(), b.double(), 1e-5)) <= 3e-6
    y = rnd((77, 512), 8, 2.5)
    s = C.gelu_split3(y.to(DEV)).cpu()
    assert s.dtype == torch.float16 and tuple(s.shape) == (77, 3 * 512)
    assert rel(s[:, :512].double() + s[:, 512:1024].double(), TF.gelu(y.double())) <= 3e-6


@pytest.mark.parametrize("H,W,size,antialias", [(100, 180, 56, True), (576, 1024, 224, True), (576, 1024, 224, False), (40, 48, 56, True)])
def test_preprocess_patches_vs_oracle(C, H, W, size, antialias):
    from oracle import openclip as O
    img = torch.tanh(rnd((2, 3, H, W), 21))
    ref = O.preprocess(img, size, antialias)                                     # [B, 3, S, S]
    P, G = 14, size // 14
    ref_p = ref.reshape(2, 3, G, P, G, P).permute(0, 2, 4, 1, 3, 5).reshape(2 * G * G, 3 * P * P)
    got = C.preprocess_patches(img.to(DEV), size, P, 640, antialias).cpu()
    assert tuple(got.shape) == (2 * G * G, 640) and not got[:, 3 * P * P:].any()
    assert rel(got[:, :3 * P * P], ref_p) <= 3e-6


def _text(C, arch, seed, layer):
    t = arch["text"]
    sd = fill(text_shapes(t["vocab_size"], t["context_length"], t["width"], t["layers"], arch["embed_dim"]), seed)
    return C.FrozenOpenCLIPEmbedder(arch=arch, layer=layer, state_dict=sd), sd


@pytest.mark.parametrize("layer", ["penultimate", "last"])
def test_narrow_text_tower_vs_oracle(C, layer):
    from oracle import openclip as O
    emb, sd = _text(C, NARROW, 31, layer)
    tokens = torch.randint(1, 49000, (3, 77), generator=torch.Generator().manual_seed(2))
    tokens[0] = C.tokenize("")[0]
    got = emb(tokens)
    assert tuple(got.shape) == (3, 77, 128) and got.dtype == torch.float32 and got.is_cuda
    assert rel(got, O.text_encode(sd, tokens, NARROW["text"]["heads"], layer)) <= 2e-5
    # the drivers' call: a list of empty prompts (sd_pipeline_vspw.py:536) -> the same row for each
    two = emb(["", ""])
    assert tuple(two.shape) == (2, 77, 128) and torch.equal(two[0], two[1]) and rel(two[0], got[0]) <= 1e-5   # distinct rows are evaluated once
    # an embedding computed elsewhere passes through; another prompt needs the BPE vocabulary this image does not have
    pre = torch.randn(2, 77, 128, device=DEV)
    assert emb(pre) is pre
    from vidseg_diffusion_amd._lib import VidsegError
    with pytest.raises(VidsegError, match="BPE vocabulary"):
        emb(["a photo of a cat"])


def test_narrow_visual_tower_and_prediction_embedder_vs_oracle(C):
    from oracle import openclip as O
    v = NARROW["vision"]
    grid = v["image_size"] // v["patch_size"]
    sd = fill(visual_shapes(v["width"], v["layers"], v["patch_size"], grid, NARROW["embed_dim"]), 41)
    emb = C.FrozenOpenCLIPImagePredictionEmbedder(
        open_clip_embedding_config={"target": "sgm.modules.encoders.modules.FrozenOpenCLIPImageEmbedder",
                                    "params": {"arch": NARROW, "freeze": True, "state_dict": {"visual." + k: t for k, t in sd.items()}}},
        n_cond_frames=1, n_copies=2)
    img = torch.tanh(rnd((2, 3, 100, 180), 42))
    got = emb(img.to(DEV))
    ref = O.image_embed(sd, img, v["width"] // v["head_width"], v["patch_size"], v["image_size"])
    assert tuple(got.shape) == (4, 1, 64)                                         # "(b t) d -> b t d", then "b t d -> (b s) t d"
    assert torch.equal(got[0], got[1]) and torch.equal(got[2], got[3])
    assert rel(got[::2, 0], ref) <= 2e-5


def test_unloaded_tower_refuses_to_run(C):
    from vidseg_diffusion_amd._lib import VidsegError
    emb = C.FrozenOpenCLIPEmbedder(arch=NARROW, layer="penultimate")
    with pytest.raises(VidsegError, match="never loaded"):
        emb([""])
    with pytest.raises(VidsegError, match="HIP device"):
        emb.model.encode(C.tokenize(""), 1)                                       # tokens on the CPU: no CPU path


def test_full_vit_h_14_towers_vs_oracle(C):
    """The sizes the reference runs: SD 2.1's empty prompt through 23 of the text tower's 24 blocks (sd_2_1.yaml:39-43) and one
    576 x 1024 conditioning frame through the image tower's 32 blocks (svd.yaml:43-50), synthetic weights."""
    from oracle import openclip as O
    a = C.ARCHS["ViT-H-14"]
    emb, sd = _text(C, a, 51, "penultimate")
    tokens = C.tokenize([""])
    got = emb([""])
    assert tuple(got.shape) == (1, 77, 1024)
    e = rel(got, O.text_encode(sd, tokens, a["text"]["heads"], "penultimate"))
    print(f"ViT-H-14 text tower, empty prompt, penultimate: max err / max |ref| = {e:.2e}")
    assert e <= 2e-5
    emb.model.release()
    del emb, sd
    v = a["vision"]
    sd = fill(visual_shapes(v["width"], v["layers"], v["patch_size"], v["image_size"] // v["patch_size"], a["embed_dim"]), 52)
    im = C.FrozenOpenCLIPImageEmbedder(state_dict={"visual." + k: t for k, t in sd.items()})
    img = torch.tanh(rnd((1, 3, 576, 1024), 53))
    got = im(img.to(DEV))
    assert tuple(got.shape) == (1, 1024)
    e = rel(got, O.image_embed(sd, img, v["width"] // v["head_width"], v["patch_size"], v["image_size"]))
    print(f"ViT-H-14 image tower, 576 x 1024 frame: max err / max |ref| = {e:.2e}")
    assert e <= 2e-5


def test_general_conditioner_with_the_towers(C):
    """svd.yaml's conditioner schema with the image tower in place: `crossattn` comes out of the tower, the unconditional one is zeros."""
    from vidseg_diffusion_amd.conditioner import GeneralConditioner
    v = NARROW["vision"]
    sd = fill(visual_shapes(v["width"], v["layers"], v["patch_size"], v["image_size"] // v["patch_size"], NARROW["embed_dim"]), 41)
    m = "sgm.modules.encoders.modules."
    cond = GeneralConditioner([
        {"is_trainable": False, "input_key": "cond_frames_without_noise", "target": m + "FrozenOpenCLIPImagePredictionEmbedder",
         "params": {"n_cond_frames": 1, "n_copies": 1, "open_clip_embedding_config": {
             "target": m + "FrozenOpenCLIPImageEmbedder", "params": {"arch": NARROW, "freeze": True}}}},
        {"input_key": "fps_id", "is_trainable": False, "target": m + "ConcatTimestepEmbedderND", "params": {"outdim": 32}}])
    miss, unexpected = cond.embedders[0].load_state_dict({"open_clip.model.visual." + k: t for k, t in sd.items()})
    assert not miss and not unexpected
    batch = {"cond_frames_without_noise": torch.tanh(rnd((1, 3, 64, 96), 3)).to(DEV), "fps_id": torch.full((1,), 6.0, device=DEV)}
    c, uc = cond.get_unconditional_conditioning(batch, force_uc_zero_embeddings=["cond_frames_without_noise"])
    assert tuple(c["crossattn"].shape) == (1, 1, 64) and c["crossattn"].abs().max() > 0 and not uc["crossattn"].any()
    assert torch.equal(c["vector"], uc["vector"]) and tuple(c["vector"].shape) == (1, 32)


def _engine(kind):
    """A narrow engine from a reference-schema config (tests/test_conditioner.py) with every parameter -- the towers' included --
    restored through the checkpoint key route."""
    import test_conditioner as TC
    from vidseg_diffusion_amd import util
    cfg = TC._narrow_model_config()["model"] if kind == "sd" else TC._narrow_svd_config()
    eng = util.instantiate_from_config(cfg)
    sd = TC._prefixed_checkpoint(eng)
    missing, unexpected = eng.load_state_dict(sd)
    assert not missing and not unexpected
    return eng, sd


def test_sd_driver_conditioning_block(C):
    """sd_pipeline_vspw.py:268-318: [""] * num_frames through the conditioner -> crossattn [F, 77, context_dim], zeros for the
    unconditional half; the rows are the text tower's embedding of the empty prompt (oracle)."""
    from oracle import openclip as O
    from vidseg_diffusion_amd.pipeline import sd_window_conditioning
    eng, sd = _engine("sd")
    c, uc = sd_window_conditioning(eng.conditioner, 5, device=DEV)
    assert set(c) == {"crossattn"} and tuple(c["crossattn"].shape) == (5, 77, 64) and not uc["crossattn"].any()
    tower = {k[len("conditioner.embedders.0.model."):]: v for k, v in sd.items() if k.startswith("conditioner.embedders.0.model.")}
    ref = O.text_encode(tower, C.tokenize([""]), 1, "penultimate")
    assert rel(c["crossattn"][3:4], ref) <= 2e-5 and torch.equal(c["crossattn"][0], c["crossattn"][4])


def test_svd_driver_conditioning_block(C):
    """svd_pipeline_vspw.py:263-302: first frame -> image tower (crossattn) and noise-augmented first-stage latent (concat), the three
    scalars -> vector; both image embeddings zeroed in the unconditional half; crossattn / concat repeated over the frames."""
    from oracle import openclip as O
    from vidseg_diffusion_amd.pipeline import svd_window_conditioning
    eng, sd = _engine("svd")
    T = 6
    frames = torch.tanh(rnd((T, 3, 64, 128), 9)).to(DEV)       # 8 x 16 latent: the first stage's mid attention takes token counts % 64 == 0
    noise = rnd((1, 3, 64, 128), 10).to(DEV)
    c, uc, extra = svd_window_conditioning(eng.conditioner, frames, fps_id=6, motion_bucket_id=127, cond_aug=0.02, noise=noise)
    assert tuple(c["crossattn"].shape) == (T, 1, 64) and tuple(c["vector"].shape) == (T, 96) and tuple(c["concat"].shape) == (T, 4, 8, 16)
    assert not uc["crossattn"].any() and not uc["concat"].any() and torch.equal(uc["vector"], c["vector"])
    assert tuple(extra["image_only_indicator"].shape) == (2, T) and extra["num_video_frames"] == T
    pre = "conditioner.embedders.0.open_clip.model.visual."
    tower = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    import test_conditioner as TC
    v = TC.NARROW_CLIP["vision"]
    ref = O.image_embed(tower, frames[:1].cpu(), v["width"] // v["head_width"], v["patch_size"], v["image_size"])
    assert rel(c["crossattn"][2], ref) <= 2e-5 and torch.equal(c["crossattn"][0], c["crossattn"][T - 1])
    lat = eng.conditioner.embedders[3](frames[:1] + 0.02 * noise)                  # the `cond_frames` embedder on its own
    assert torch.equal(c["concat"][T - 1], lat[0]) and torch.equal(c["vector"][0], c["vector"][T - 1])
