"""CPU enqueue time vs GPU time of one full-size SD UNet forward (batch 28)."""
import sys, time; sys.path.insert(0, '.')
import torch
from vidseg_diffusion_amd import synthetic, ops
from vidseg_diffusion_amd.unet import UNetModel
dev = torch.device('cuda:0')
cfg = dict(synthetic.SD21_FULL)
net = UNetModel(**cfg)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1234).items()})
net.pack(dev)
x = torch.randn(28, 64, 64, 4, device=dev)
t = torch.full((28,), 958.0, device=dev)
ctx = torch.randn(28, 77, 1024, device=dev).to(ops.act_dtype())
for _ in range(2): net.forward_nhwc(x, t, ctx)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    net.forward_nhwc(x, t, ctx)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
# back-to-back x4
t0 = time.perf_counter()
for _ in range(4): net.forward_nhwc(x, t, ctx)
torch.cuda.synchronize()
print(f"4 forwards: {1e3*(time.perf_counter()-t0)/4:.1f} ms each")
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out = net.forward_nhwc(x, t, ctx)
    torch.cuda.synchronize()
    for _ in range(2): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4): g.replay()
    torch.cuda.synchronize()
    print(f"graph replay: {1e3*(time.perf_counter()-t0)/4:.1f} ms each")
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
