"""Kernels that still carry packed fp32 VALU ops (profiles/KERNEL_NOTES.md, round 6), each run many times while full-size exact-mode UNet
evaluations run on a second stream: bit-stability of k_mean_normalize* (the analysis stream shares the chip with the next window's UNet
in every pipelined run), the 16-bit attention kernels and the 16-bit GEMM epilogues."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from vidseg_diffusion_amd import analysis as A
    from vidseg_diffusion_amd import ops, synthetic
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    eng, cfg, _sd, _n = bench.build(False, False, dev)
    net = eng.model.diffusion_model
    net.set_precision("exact")
    net.tap_mode = "none"
    net._set_taps()
    g = torch.Generator().manual_seed(3)
    ux = torch.randn(28, 4, 64, 64, generator=g).to(dev)
    ut = torch.full((28,), 958.0, device=dev)
    uctx = torch.randn(28, 77, 1024, generator=g).to(dev)
    net(ux, timesteps=ut, context=uctx)
    torch.cuda.synchronize()
    ad = ops.act_dtype()
    cases = []
    for (F, fh, fw) in ((14, 32, 32), (14, 36, 64)):
        blocks, _ = synthetic.attention_q_dumps(F, fh, fw, 640, num_blocks=3, seed=1)
        dumps = [torch.from_numpy(b).to(dev) for b in blocks]
        n = F * fh * fw
        cases.append((f"k_mean_normalize {F}x{fh}x{fw}x640", 6000, lambda dumps=dumps, n=n: A.mean_normalize(dumps, n, n)[1]))
    q, k, v = (torch.randn(28, 4096, 320, generator=g).to(dev).to(ad) for _ in range(3))
    cases.append(("16-bit attention N4096 H5", 600, lambda: ops.attention(q, k, v, 5)))
    q2, k2, v2 = (torch.randn(28, 1024, 640, generator=g).to(dev).to(ad) for _ in range(3))
    cases.append(("16-bit attention N1024 H10", 1500, lambda: ops.attention(q2, k2, v2, 10)))
    a = torch.randn(114688, 320, generator=g).to(dev).to(ad)
    w = ops.pack_linear((torch.randn(320, 320, generator=g) * 0.02), dev)
    bias = torch.randn(320, generator=g).to(dev)
    r = torch.randn(114688, 320, generator=g).to(dev).to(ad)
    cases.append(("16-bit linear 114688x320x320 + residual (k_gemm_ws)", 1500, lambda: ops.linear(a, w, bias, residual=r)))
    a2 = torch.randn(28672, 640, generator=g).to(dev).to(ad)
    w2 = ops.pack_linear((torch.randn(1920, 640, generator=g) * 0.02), dev)
    cases.append(("16-bit linear 28672x1920x640", 1500, lambda: ops.linear(a2, w2, None)))
    side = torch.cuda.Stream()
    keep = []
    total = 0
    for name, iters, fn in cases:
        ref = fn().clone()
        torch.cuda.synchronize()
        bad = 0
        for it in range(iters):
            if it % 40 == 0:
                with torch.cuda.stream(side):
                    keep.append(net(ux, timesteps=ut, context=uctx))
                    del keep[:-2]
            if not torch.equal(fn(), ref):
                bad += 1
        torch.cuda.synchronize()
        total += bad
        print(f"{name:60s} {iters} runs beside UNet evaluations: {'OK' if bad == 0 else f'{bad} DIFFER'}", flush=True)
    print("unstable results:", total)


if __name__ == "__main__":
    main()
