"""smoke(): one narrow-width SD UNet forward on cuda:0 in both precision modes, checked against the oracle.  Test infrastructure (it imports `oracle`):
lives under tests/ and is called by __graft_entry__.smoke() only."""
import numpy as np
import torch


def run(dev):
    from oracle.unet import UNetOracle            # noqa: checker only
    from vidseg_diffusion_amd import synthetic
    from vidseg_diffusion_amd.unet import UNetModel
    net = UNetModel(**synthetic.SD21_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, 1234).items()}
    net.load_state_dict(sd)
    g = np.random.Generator(np.random.PCG64(2))
    x = torch.from_numpy(g.standard_normal((2, 4, 16, 16)).astype(np.float32))
    t = torch.tensor([958.0, 958.0])
    ctx = torch.from_numpy(g.standard_normal((2, 7, 64)).astype(np.float32))
    out = net(x.to(dev), timesteps=t.to(dev), context=ctx.to(dev)).cpu()
    ref = UNetOracle(sd).forward(x, t, ctx)
    err = float((out - ref).norm() / ref.norm())
    assert err < 3e-2, f"smoke: UNet forward deviates {err:.3g} (normalised rms) from the fp32 oracle"
    from vidseg_diffusion_amd import ops
    if ops.act_dtype() == torch.float16:            # the precision mode bench.py quotes `value` on (exact.py; fp16 build only)
        net.set_precision("exact")
        out = net(x.to(dev), timesteps=t.to(dev), context=ctx.to(dev)).cpu()
        err = float((out - ref).norm() / ref.norm())
        assert err < 1e-4, f"smoke: exact-mode UNet forward deviates {err:.3g} (normalised rms) from the fp32 oracle"
