"""Full-size check of pipeline.feature_pass(masks_only=True): taps of decoder blocks 6-8 (cond half) vs the full schedule."""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from vidseg_diffusion_amd import feature_extraction as FE, synthetic
from vidseg_diffusion_amd.pipeline import build_sd_engine, feature_pass
from vidseg_diffusion_amd.unet import UNetModel
dev = torch.device("cuda:0")
F = 14
cfg = dict(synthetic.SD21_FULL)
net = UNetModel(**cfg)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1234).items()})
net.pack(dev)
eng = build_sd_engine(net, num_steps=25, scale=5.0)
lat = torch.from_numpy(synthetic.latent_clip(F, 64, 64, seed=1)).to(dev)
cc, ucc = synthetic.sd_conditioning(F, context_dim=cfg["context_dim"])
c, uc = {"crossattn": torch.from_numpy(cc).to(dev)}, {"crossattn": torch.from_numpy(ucc).to(dev)}
noise = torch.from_numpy(np.random.Generator(np.random.PCG64(9)).standard_normal(tuple(lat.shape)).astype(np.float32)).to(dev)
taps = {}
for tag, kw in (("full", {}), ("full2", {}), ("pruned", {"masks_only": True})):
    FE.FeatureStore.clear()
    feature_pass(eng, lat, c, uc, t_start=22, seed=17, noise=noise, feature_folder="/nonexistent/moc", exp_name=tag, keep_all_steps=False, **kw)
    torch.cuda.synchronize()
    st = FE.FeatureStore.folder("/nonexistent/moc", tag)
    taps[tag] = {b: st[f"output_block_{b}_spatial_self_attn_q_time_24"][F:].float() for b in (6, 7, 8)}
for b in (6, 7, 8):
    a, p, a2 = taps["full"][b], taps["pruned"][b], taps["full2"][b]
    print(f"block {b}: run-to-run identical {torch.equal(a, a2)}; pruned vs full nrms {((p - a).norm() / a.norm()).item():.3e}, "
          f"differing fp16 values {(p != a).float().mean().item():.4f}, max |diff| / max |tap| {((p - a).abs().max() / a.abs().max()).item():.3e}")
