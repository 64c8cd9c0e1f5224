"""Steps 4-5 of one SD window alone (bench.step45_sd) -- `python tools/step45_timing.py [--decode-only]`; with --decode-only just a few
first-stage decodes of a 14-frame 512x512 window (for a rocprofv3 --kernel-trace --stats of the decoder)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    if "--decode-only" in sys.argv:
        from vidseg_diffusion_amd import synthetic
        from vidseg_diffusion_amd.vae import AutoencoderKL, decode_first_stage
        dd = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
                  attn_resolutions=[], dropout=0.0)
        vae = AutoencoderKL(embed_dim=4, ddconfig=dd)
        vshapes = {k: tuple(v.shape) for k, v in vae.state_dict().items()}
        vae.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(vshapes, seed=99).items()})
        z = torch.randn(14, 4, 64, 64, generator=torch.Generator().manual_seed(1)).to(dev) * 0.18215 * 4
        for _ in range(2):
            decode_first_stage(vae, z, 0.18215)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            decode_first_stage(vae, z, 0.18215)
        torch.cuda.synchronize()
        print(json.dumps({"decode_ms_per_window": round(1e3 * (time.perf_counter() - t0) / 4, 2)}))
        return
    eng, cfg, _sd, _n = bench.build(False, False, dev)
    print(json.dumps(bench.step45_sd(eng, cfg, dev, 20), indent=1))


if __name__ == "__main__":
    main()
