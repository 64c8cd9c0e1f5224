"""tests/golden/vae_encoder_narrow.npz: the REFERENCE's first-stage Encoder (sgm/modules/diffusionmodules/model.py) +
quant_conv + DiagonalGaussianDistribution on a narrow configuration (ch 64, same topology as sd_2_1.yaml:44-61).
Build-container only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ref_import import import_reference  # noqa: E402
from vidseg_diffusion_amd import synthetic  # noqa: E402

VAE_NARROW = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2, 4, 4],
                  num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def main():
    import_reference()
    from sgm.modules.diffusionmodules.model import Encoder
    from sgm.modules.distributions.distributions import DiagonalGaussianDistribution
    torch.set_grad_enabled(False)
    enc = Encoder(**VAE_NARROW).eval()
    quant = torch.nn.Conv2d(8, 8, 1)
    shapes = {"encoder." + k: tuple(v.shape) for k, v in enc.state_dict().items()}
    shapes.update({"quant_conv." + k: tuple(v.shape) for k, v in quant.state_dict().items()})
    sd = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=2468, gain=1.0).items()}
    enc.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")})
    quant.load_state_dict({k[len("quant_conv."):]: v for k, v in sd.items() if k.startswith("quant_conv.")})
    g = np.random.Generator(np.random.PCG64(77))
    x = np.clip(g.standard_normal((2, 3, 64, 64)).astype(np.float32) * 0.5, -1, 1)
    mom = quant(enc(torch.from_numpy(x)))
    torch.manual_seed(11)
    noise = torch.randn(2, 4, 8, 8)
    torch.manual_seed(11)
    z = DiagonalGaussianDistribution(mom).sample() * 0.18215
    rec = dict(x=x, moments=mom.numpy(), noise=noise.numpy(), z=z.numpy(), state_dict_signature=synthetic.state_dict_signature(shapes))
    path = os.path.join(ROOT, "tests", "golden", "vae_encoder_narrow.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; moments absmax", float(np.abs(rec["moments"]).max()),
          "logvar range", float(mom[:, 4:].min()), float(mom[:, 4:].max()))


def main_decoder():
    """tests/golden/vae_decoder_narrow.npz: the reference's Decoder behind post_quant_conv (autoencoder.py:490-506)."""
    import_reference()
    from sgm.modules.diffusionmodules.model import Decoder
    torch.set_grad_enabled(False)
    dec = Decoder(**VAE_NARROW).eval()
    pq = torch.nn.Conv2d(4, 4, 1)
    shapes = {"decoder." + k: tuple(v.shape) for k, v in dec.state_dict().items()}
    shapes.update({"post_quant_conv." + k: tuple(v.shape) for k, v in pq.state_dict().items()})
    sd = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1357, gain=1.0).items()}
    # biases are zero in the synthetic filler; give post_quant_conv one so the padding interaction of the folded conv is exercised
    sd["post_quant_conv.bias"] = torch.tensor([0.3, -0.2, 0.1, 0.25])
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")})
    pq.load_state_dict({k[len("post_quant_conv."):]: v for k, v in sd.items() if k.startswith("post_quant_conv.")})
    g = np.random.Generator(np.random.PCG64(78))
    z = (g.standard_normal((2, 4, 8, 8)).astype(np.float32) * 0.18215 * 4)
    out = dec(pq(torch.from_numpy(z) / 0.18215))
    rec = dict(z=z, out=out.numpy(), pq_bias=sd["post_quant_conv.bias"].numpy(), state_dict_signature=synthetic.state_dict_signature(shapes))
    path = os.path.join(ROOT, "tests", "golden", "vae_decoder_narrow.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; out absmax", float(np.abs(rec["out"]).max()), "rms", float(out.pow(2).mean().sqrt()))


def main_video_decoder():
    """tests/golden/vae_video_decoder_narrow.npz: the reference's temporal_ae.VideoDecoder (svd.yaml:119-133, narrow), T = 3."""
    import_reference()
    from sgm.modules.autoencoding.temporal_ae import VideoDecoder
    torch.set_grad_enabled(False)
    dec = VideoDecoder(**dict(VAE_NARROW, attn_type="vanilla", video_kernel_size=[3, 1, 1])).eval()
    shapes = {"decoder." + k: tuple(v.shape) for k, v in dec.state_dict().items()}
    sdn = synthetic.fill_state_dict(shapes, seed=9753, gain=1.0)
    g = np.random.Generator(np.random.PCG64(79))
    for k in sdn:                                        # the filler leaves 1-element and bias tensors trivial: make the time path count
        if k.endswith("mix_factor"):
            sdn[k] = g.uniform(-1.0, 1.0, sdn[k].shape).astype(np.float32)
        if "time_mix_conv" in k or (k.endswith(".bias") and "time_stack" in k and "layers.0" not in k):
            sdn[k] = (g.standard_normal(sdn[k].shape) * (0.3 if k.endswith("weight") else 0.05)).astype(np.float32)
    sd = {k: torch.from_numpy(v) for k, v in sdn.items()}
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items()})
    T = 3
    z = (g.standard_normal((2 * T, 4, 8, 8)).astype(np.float32) * 0.18215 * 4)
    out = dec(torch.from_numpy(z) / 0.18215, timesteps=T)
    extra = {k: v for k, v in sdn.items() if k.endswith("mix_factor") or "time_mix_conv" in k or (k.endswith(".bias") and "time_stack" in k and "layers.0" not in k)}
    rec = dict(z=z, out=out.numpy(), T=T, state_dict_signature=synthetic.state_dict_signature(shapes),
               extra_keys=np.array(sorted(extra)), **{"extra_%d" % i: extra[k] for i, k in enumerate(sorted(extra))})
    path = os.path.join(ROOT, "tests", "golden", "vae_video_decoder_narrow.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; out absmax", float(np.abs(rec["out"]).max()), "rms", float(out.pow(2).mean().sqrt()))


if __name__ == "__main__":
    if "--video-decoder" in sys.argv:
        main_video_decoder()
    elif "--decoder" in sys.argv:
        main_decoder()
    else:
        main()
