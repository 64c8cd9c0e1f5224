"""Oracle of the first-stage encoder vs the reference's own output (tests/golden/vae_encoder_narrow.npz)."""
import os

import numpy as np
import torch

from oracle.vae import VAEEncoderOracle
from vidseg_diffusion_amd import synthetic

VAE_NARROW = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2, 4, 4],
                  num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def narrow_state_dict():
    from vidseg_diffusion_amd.vae import AutoencoderKL
    net = AutoencoderKL(embed_dim=4, ddconfig=VAE_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    return net, shapes, {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=2468, gain=1.0).items()}


def test_mirror_keys_and_oracle_match_reference():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vae_encoder_narrow.npz"))
    net, shapes, sd = narrow_state_dict()
    assert synthetic.state_dict_signature(shapes) == str(g["state_dict_signature"])     # same keys and shapes as the reference modules
    o = VAEEncoderOracle(sd)
    mom = o.moments(torch.from_numpy(g["x"]))
    assert np.abs(mom.numpy() - g["moments"]).max() <= 2e-5 * np.abs(g["moments"]).max()
    z = o.encode(torch.from_numpy(g["x"]), torch.from_numpy(g["noise"]), 0.18215)
    assert np.abs(z.numpy() - g["z"]).max() <= 2e-5 * np.abs(g["z"]).max()


def test_checkpoint_with_decoder_keys_loads():
    net, shapes, sd = narrow_state_dict()
    sd["decoder.conv_in.weight"] = torch.zeros(1)
    sd["post_quant_conv.weight"] = torch.zeros(1)
    missing, unexpected = net.load_state_dict(sd)
    assert not missing and not unexpected
