"""Time of the first-stage encode for one 14-frame 512x512 window (reported beside, never inside, bench.py's metric:
SURVEY.md §8(d) "VAE/conditioner excluded and reported separately")."""
import json, sys, time
sys.path.insert(0, '.')
import torch
from vidseg_diffusion_amd import synthetic
from vidseg_diffusion_amd.vae import AutoencoderKL, encode_first_stage

DD = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
          attn_resolutions=[], dropout=0.0)
dev = torch.device("cuda:0")
net = AutoencoderKL(embed_dim=4, ddconfig=DD)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=99).items()})
F, H, W = 14, (int(sys.argv[1]) if len(sys.argv) > 1 else 512), (int(sys.argv[2]) if len(sys.argv) > 2 else 512)
x = (torch.rand(F, 3, H, W) * 2 - 1).to(dev)
noise = torch.randn(F, 4, H // 8, W // 8)
for _ in range(2):
    z = encode_first_stage(net, x, 0.18215, noise=noise)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    z = encode_first_stage(net, x, 0.18215, noise=noise)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / n
flops = 1.12e12 * F * (H * W) / (512 * 512)          # encoder 1.12 TFLOP/frame @512^2
print(json.dumps({"what": "first-stage encode, one window", "frames": F, "height": H, "width": W, "ms": round(ms, 2),
                  "frames_per_s": round(1e3 * F / ms, 1), "tflops": round(flops / ms / 1e9, 1), "finite": bool(torch.isfinite(z).all())}))
