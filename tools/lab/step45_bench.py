"""Steps 1-5 for one BASELINE configs[1] window at full size (SD 2.1 UNet + first stage, synthetic weights), everything in
HBM: feature pass + masks, then the 2*K modulated passes, 2*K decodes, difference maps and arg-max.  Reported beside the
headline metric, never inside it.  usage: python tools/lab/step45_bench.py [K]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vidseg_diffusion_amd import feature_extraction as FE  # noqa: E402
from vidseg_diffusion_amd import synthetic  # noqa: E402
from vidseg_diffusion_amd.pipeline import build_sd_engine, segment_window, segmentation_map_window  # noqa: E402
from vidseg_diffusion_amd.unet import UNetModel  # noqa: E402
from vidseg_diffusion_amd.vae import AutoencoderKL  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
F, LAT = 14, 64
cfg = dict(synthetic.SD21_FULL)
net = UNetModel(**cfg)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1234).items()})
net.pack(dev)
eng = build_sd_engine(net, num_steps=25, scale=5.0)
dd = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
          attn_resolutions=[], dropout=0.0)
vae = AutoencoderKL(embed_dim=4, ddconfig=dd)
vshapes = {k: tuple(v.shape) for k, v in vae.state_dict().items()}
vae.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(vshapes, seed=99).items()})
lat = torch.from_numpy(synthetic.latent_clip(F, LAT, LAT, seed=1)).to(dev)
cc, ucc = synthetic.sd_conditioning(F, context_dim=cfg["context_dim"])
c, uc = {"crossattn": torch.from_numpy(cc).to(dev)}, {"crossattn": torch.from_numpy(ucc).to(dev)}
noise = torch.from_numpy(np.random.Generator(np.random.PCG64(9)).standard_normal(tuple(lat.shape)).astype(np.float32)).to(dev)
base, exp = "/nonexistent/step45", "w"
for it in range(2):
    FE.FeatureStore.clear()
    FE.MaskStore.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    labels, _ = segment_window(eng, lat, c, uc, num_masks=K, t_start=22, is_aggre_attn=True, seed=17, noise=noise, feature_folder=base,
                               exp_name=exp, keep_all_steps=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    uniq = np.unique(labels)
    folder = os.path.join(base, exp, "match_gt_mask", f"output_block_8_output_block_7_output_block_6_spatial_self_attn_q_masks_{K}")
    seg, _ = segmentation_map_window(eng, vae, lat, c, uc, uniq, folder, t_start=22, feature_folder=base, exp_name=exp, noise=noise, seed=17)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"run {it}: steps 1-3 {1e3 * (t1 - t0):.0f} ms; steps 4-5 ({len(uniq)} labels: {2 * len(uniq)} modulated passes + decodes, "
          f"seg map {tuple(seg.shape)}) {t2 - t1:.2f} s; labels in seg map {len(torch.unique(seg))}", flush=True)
