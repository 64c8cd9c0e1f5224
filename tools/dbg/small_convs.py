import sys; sys.path.insert(0, '.')
import torch
from vidseg_diffusion_amd import ops
dev = torch.device('cuda:0')
def bench(fn, name, nbytes, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"{name:44s} {ms*1e3:8.1f} us  {nbytes/ms/1e6:7.1f} GB/s")
for (B, H, W, Cin, Cout) in [(28, 64, 64, 4, 320), (28, 72, 128, 8, 320), (14, 512, 512, 3, 128)]:
    x = torch.randn(B, H, W, Cin, device=dev); w = ops.pack_conv_in(torch.randn(Cout, Cin, 3, 3) * 0.05, dev); b = torch.zeros(Cout, device=dev)
    bench(lambda: ops.conv_in(x, w, b), f"conv_in B{B} {H}x{W} {Cin}->{Cout}", B * H * W * (Cin * 4 + Cout * 2))
for (B, H, W, Cin) in [(28, 64, 64, 320), (28, 72, 128, 320)]:
    x = torch.randn(B, H, W, Cin, device=dev).to(ops.act_dtype()); w = ops.pack_conv_out(torch.randn(4, Cin, 3, 3) * 0.05, dev); b = torch.zeros(4, device=dev)
    bench(lambda: ops.conv_out4(x, w, b), f"conv_out4 B{B} {H}x{W} {Cin}->4", B * H * W * (Cin * 2 + 16))
