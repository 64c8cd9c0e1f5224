import sys; sys.path.insert(0, '.')
import numpy as np, torch
from vidseg_diffusion_amd import synthetic, ops
from vidseg_diffusion_amd.unet import UNetModel
from vidseg_diffusion_amd.pipeline import build_sd_engine
from oracle.unet import UNetOracle, legacy_ddpm_sigmas, discrete_sigma_table, sigma_to_idx
dev = torch.device('cuda:0')
g = np.load('tests/golden/unet_sd_narrow.npz')
net = UNetModel(**synthetic.SD21_NARROW)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
sd = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, 1234).items()}
net.load_state_dict(sd)
def nrms(a, b): return float((a - b).norm() / b.norm())
eng = build_sd_engine(net)
o = UNetOracle(sd)
sig = legacy_ddpm_sigmas(25); table = discrete_sigma_table()
x = torch.from_numpy(g['sm_noised']) * torch.sqrt(1 + sig[0] ** 2)
c = torch.from_numpy(g['sm_c']); uc = torch.zeros_like(c)
s2 = torch.full((4,), float(sig[22])); idx = sigma_to_idx(s2, table); sq = table[idx]
c_in = 1 / (sq ** 2 + 1) ** 0.5
xin = torch.cat([x, x]) * c_in[:, None, None, None]
ctx = torch.cat([uc, c])
net_o = o.forward(xin, sigma_to_idx(sq, table).float(), ctx)
# gpu pieces
print("table eq", np.abs(eng.denoiser.sigmas.numpy() - table.numpy()).max(), "idx", idx, eng.denoiser.sigma_to_idx(s2))
net_g = net(xin.to(dev), timesteps=sigma_to_idx(sq, table).float().to(dev), context=ctx.to(dev)).cpu()
print("net direct", nrms(net_g, net_o))
den_g = eng.denoiser(eng.model, torch.cat([x, x]).to(dev), s2, {"crossattn": ctx.to(dev)}).cpu()
den_o = net_o * (-sq)[:, None, None, None] + torch.cat([x, x])
print("denoised", nrms(den_g, den_o))
xs = eng.sampler.sampler_step(torch.ones(2) * sig[22], torch.ones(2) * sig[23], lambda i, s, cc, **k: eng.denoiser(eng.model, i, s, cc),
                              x.to(dev), {"crossattn": c.to(dev)}, {"crossattn": uc.to(dev)}).cpu()
print("step", nrms(xs, torch.from_numpy(g['sm_x_steps'][0])))
