"""Oracle of the first stage (encoder, decoder) vs the reference's own outputs (tests/golden/vae_{encoder,decoder}_narrow.npz)."""
import os

import numpy as np
import torch

from oracle.vae import VAEDecoderOracle, VAEEncoderOracle
from vidseg_diffusion_amd import synthetic

VAE_NARROW = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2, 4, 4],
                  num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def narrow_state_dict():
    """Encode side: the mirror's encoder.* / quant_conv.* keys, filled like tools/gen_golden_vae.py does."""
    from vidseg_diffusion_amd.vae import AutoencoderKL
    net = AutoencoderKL(embed_dim=4, ddconfig=VAE_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items() if k.split(".")[0] in ("encoder", "quant_conv")}
    return net, shapes, {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=2468, gain=1.0).items()}


def narrow_decoder_state_dict(pq_bias):
    from vidseg_diffusion_amd.vae import AutoencoderKL
    net = AutoencoderKL(embed_dim=4, ddconfig=VAE_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items() if k.split(".")[0] in ("decoder", "post_quant_conv")}
    sd = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1357, gain=1.0).items()}
    sd["post_quant_conv.bias"] = torch.from_numpy(np.asarray(pq_bias, dtype=np.float32))
    return net, shapes, sd


def test_mirror_keys_and_oracle_match_reference():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vae_encoder_narrow.npz"))
    net, shapes, sd = narrow_state_dict()
    assert synthetic.state_dict_signature(shapes) == str(g["state_dict_signature"])     # same keys and shapes as the reference modules
    o = VAEEncoderOracle(sd)
    mom = o.moments(torch.from_numpy(g["x"]))
    assert np.abs(mom.numpy() - g["moments"]).max() <= 2e-5 * np.abs(g["moments"]).max()
    z = o.encode(torch.from_numpy(g["x"]), torch.from_numpy(g["noise"]), 0.18215)
    assert np.abs(z.numpy() - g["z"]).max() <= 2e-5 * np.abs(g["z"]).max()


def test_decoder_mirror_keys_and_oracle_match_reference():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vae_decoder_narrow.npz"))
    net, shapes, sd = narrow_decoder_state_dict(g["pq_bias"])
    assert synthetic.state_dict_signature(shapes) == str(g["state_dict_signature"])     # decoder.* / post_quant_conv.* keys and shapes
    out = VAEDecoderOracle(sd).decode(torch.from_numpy(g["z"]), 0.18215)
    assert out.shape == g["out"].shape
    assert np.abs(out.numpy() - g["out"]).max() <= 2e-5 * np.abs(g["out"]).max()


def narrow_video_decoder_state_dict(g):
    """The mirror's AutoencodingEngine with a VideoDecoder, filled like tools/gen_golden_vae.py --video-decoder does."""
    from vidseg_diffusion_amd.vae import AutoencodingEngine
    dd = dict(VAE_NARROW, attn_type="vanilla")
    net = AutoencodingEngine(encoder_config={"target": "sgm.modules.diffusionmodules.model.Encoder", "params": dd},
                             decoder_config={"target": "sgm.modules.autoencoding.temporal_ae.VideoDecoder",
                                             "params": dict(dd, video_kernel_size=[3, 1, 1])})
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items() if k.startswith("decoder.")}
    sdn = synthetic.fill_state_dict(shapes, seed=9753, gain=1.0)
    for i, k in enumerate(g["extra_keys"]):
        sdn[str(k)] = g["extra_%d" % i]
    return net, shapes, {k: torch.from_numpy(np.asarray(v)) for k, v in sdn.items()}


def test_video_decoder_mirror_keys_and_oracle_match_reference():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vae_video_decoder_narrow.npz"))
    net, shapes, sd = narrow_video_decoder_state_dict(g)
    assert synthetic.state_dict_signature(shapes) == str(g["state_dict_signature"])     # temporal_ae.VideoDecoder keys and shapes
    out = VAEDecoderOracle(sd).decode(torch.from_numpy(g["z"]), 0.18215, timesteps=int(g["T"]))
    assert np.abs(out.numpy() - g["out"]).max() <= 2e-5 * np.abs(g["out"]).max()
    # the time path matters: decoding the frames as independent images gives something else
    img = VAEDecoderOracle(sd).decode(torch.from_numpy(g["z"]), 0.18215)
    assert np.abs(img.numpy() - g["out"]).max() > 1e-2 * np.abs(g["out"]).max()


def test_encoder_only_checkpoint_loads():
    net, shapes, sd = narrow_state_dict()
    sd["loss.logvar"] = torch.zeros(1)                                                  # training-only keys of a checkpoint are ignored
    missing, unexpected = net.load_state_dict(sd)
    assert not unexpected and all(k.split(".")[0] in ("decoder", "post_quant_conv") for k in missing)
