import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from vidseg_diffusion_amd import synthetic, ops
from vidseg_diffusion_amd.unet import UNetModel
from oracle.unet import UNetOracle
dev = torch.device('cuda:0')
g = np.load('tests/golden/unet_sd_narrow.npz')
net = UNetModel(**synthetic.SD21_NARROW)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
sd = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, 1234).items()}
net.load_state_dict(sd)
x, t, ctx = (torch.from_numpy(g[k]) for k in ('fw_x', 'fw_t', 'fw_ctx'))
out = net(x.to(dev), timesteps=t.to(dev), context=ctx.to(dev)).cpu()
ref = torch.from_numpy(g['fw_out'])
o32 = UNetOracle(sd); r32 = o32.forward(x, t, ctx)
obf = UNetOracle(sd, round_bf16=True); rbf = obf.forward(x, t, ctx)
def nerr(a, b): return float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm())
print("gpu vs reference(fp32) max/rms", nerr(out, ref))
print("gpu vs oracle bf16-matched   ", nerr(out, rbf))
print("oracle bf16 vs fp32          ", nerr(rbf, r32))
for b in (3, 6, 7, 8, 11):
    tb = net.output_blocks[b][1].transformer_blocks[0]
    for nm, a in (("self", tb.attn1), ("cross", tb.attn2)):
        k = f"output_block_{b}_spatial_{nm}_attn_q"
        print(k, "vs ref", nerr(a.q.float().cpu(), torch.from_numpy(g['fw_' + k]).float()), "vs bf16 oracle", nerr(a.q.float().cpu(), obf.taps[k].float()))
# full-size timing
if len(sys.argv) > 1:
    F = 14
    net = UNetModel(**synthetic.SD21_FULL)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    t0 = time.time(); sd = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, 1234).items()}; print("fill", time.time() - t0)
    net.load_state_dict(sd); t0 = time.time(); net.pack(dev); torch.cuda.synchronize(); print("pack", time.time() - t0)
    x = torch.randn(2 * F, 64, 64, 4, device=dev); tt = torch.full((2 * F,), 500.0, device=dev); ctx = torch.randn(2 * F, 77, 1024, device=dev).to(ops.act_dtype())
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        o = net.forward_nhwc(x, tt, ctx); torch.cuda.synchronize(); print("full fwd ms", (time.time() - t0) * 1e3, float(o.abs().mean()))
