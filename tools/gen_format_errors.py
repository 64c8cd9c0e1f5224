"""tests/golden/format_errors.json: what the 16-bit STORAGE FORMAT alone costs on the two inversion trajectories of the narrow SD UNet --
the CPU oracle with every matmul operand and stored activation rounded to fp16 / bf16 (oracle.unet.UNetOracle(round_bf16=...)) against the
reference's fp32 goldens.  tests/test_gpu_unet.py::test_inversion_vs_reference / ::test_inversion_window_vs_reference hold the device
to 1.2-1.3 x these figures; computing them takes the oracle 49 + 24 network evaluations per format (a minute and a half of the GPU
suite), so they are generated once here.  No reference import: the goldens are already committed.     python tools/gen_format_errors.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.unet import UNetOracle, euler_inversion, euler_sample  # noqa: E402
from vidseg_diffusion_amd import synthetic  # noqa: E402
from vidseg_diffusion_amd.unet import UNetModel  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def nrms(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def main():
    torch.set_grad_enabled(False)
    net = UNetModel(**synthetic.SD21_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, 1234).items()}
    g = np.load(os.path.join(G, "unet_sd_narrow.npz"))
    w = np.load(os.path.join(G, "sd_inversion_window_narrow.npz"))
    out = {}
    for fmt in ("f16", "bf16"):
        o = UNetOracle(sd, round_bf16=fmt)
        cc = torch.from_numpy(g["sm_c"])
        xo, _ = euler_inversion(o, torch.from_numpy(g["sm_latent"]), cc, torch.zeros_like(cc))
        rec = {"inversion_final": nrms(xo.numpy(), g["inv_final"])}
        cw = torch.from_numpy(w["c"])
        inv, _ = euler_inversion(o, torch.from_numpy(w["latent"]), cw, torch.zeros_like(cw))
        fx, ft = {}, {}

        def cb(x, i, t):
            if i in (0, 12, 24):
                fx[i] = x.numpy().copy()
            if i == 24:
                for b in (6, 7, 8):
                    ft[b] = t[f"output_block_{b}_spatial_self_attn_q"].float().numpy().copy()

        euler_sample(o, inv, cw, torch.zeros_like(cw), t_start=0, noise=None, callback=cb)
        for i in (0, 12, 24):
            rec[f"window_x_step{i}"] = nrms(fx[i], w[f"x_step{i}"])
        for b in (6, 7, 8):
            rec[f"window_q{b}"] = nrms(ft[b], w[f"q{b}"].astype(np.float32))
        out[fmt] = rec
        print(fmt, rec, flush=True)
    with open(os.path.join(G, "format_errors.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
