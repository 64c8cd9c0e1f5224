"""Generate tests/golden/unet_svd_narrow.npz from the REFERENCE's VideoUNet (read-only import) on a narrow-width
instance of the SVD topology, with vidseg_diffusion_amd.synthetic.fill_state_dict weights.  Build-container only."""
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
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    torch.set_grad_enabled(False)
    cfg = dict(synthetic.SVD_NARROW)
    net = VideoUNet(use_checkpoint=False, spatial_transformer_attn_type="softmax", **cfg).eval()   # xformers absent
    net = net.to("cpu")
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synthetic.fill_state_dict(shapes, seed=4321)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    T = 3
    g = np.random.Generator(np.random.PCG64(6))
    x = g.standard_normal((2 * T, 8, 16, 16)).astype(np.float32)
    t = np.full((2 * T,), -0.7, dtype=np.float32)                  # c_noise = 0.25 log sigma is a small real number for SVD
    ctx = np.repeat(g.standard_normal((2, 1, 64)).astype(np.float32), T, axis=0)
    y = g.standard_normal((2 * T, 64)).astype(np.float32)
    out = net(torch.from_numpy(x), timesteps=torch.from_numpy(t), context=torch.from_numpy(ctx), y=torch.from_numpy(y),
              num_video_frames=T, image_only_indicator=torch.zeros(2, T))
    rec = dict(state_dict_signature=synthetic.state_dict_signature(shapes), weight_seed=4321, T=T, fw_x=x, fw_t=t, fw_ctx=ctx, fw_y=y,
               fw_out=out.numpy())
    for i in (3, 7, 8, 11):
        blk = net.output_blocks[i]
        assert "SpatialVideoTransformer" in str(type(blk[1]))
        rec[f"fw_output_block_{i}_spatial_self_attn_q"] = blk[1].transformer_blocks[0].attn1.q.half().numpy()
        rec[f"fw_output_block_{i}_temporal_self_attn_q"] = blk[1].time_stack[0].attn1.q.half().numpy()
        rec[f"fw_output_block_{i}_temporal_self_attn_k"] = blk[1].time_stack[0].attn1.k.half().numpy()
        rec[f"fw_output_block_{i}_temporal_cross_attn_k"] = blk[1].time_stack[0].attn2.k.half().numpy()
    path = os.path.join(ROOT, "tests", "golden", "unet_svd_narrow.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", float(np.abs(out.numpy()).mean()), len(shapes), "tensors",
          sum(int(np.prod(s)) for s in shapes.values()) / 1e6, "M params")


if __name__ == "__main__":
    main()
