"""The small-M split-operand launches of the pruned last step (14 samples) under the two selection rules: VIDSEG_GEMM="xsmall=0" (the
16-bit thresholds of the big tile: these shapes stay on k_gemm_dma's 3K walk) against "xsmall=1" (k_gemm_p7x from half a round of tiles
and K' / S >= 960 on).  Results compared, time per launch.        usage: python tools/xsmall_bench.py"""
import os
import subprocess
import sys

sys.path.insert(0, ".")
LINS = [  # M, K, N, residual  (K = the layer's width; the launch sees K' = 3K)
    (3584, 1280, 1280, True), (14336, 640, 640, True), (3584, 1280, 3840, False), (1792, 1280, 1280, True), (1792, 2560, 1280, True), (896, 1280, 1280, True),
    (896, 2560, 1280, True), (896, 5120, 1280, True), (7168, 640, 1280, False), (3584, 640, 1280, False), (14336, 320, 640, False), (2156, 1024, 1280, False),
    (2156, 1024, 2560, False), (1078, 1024, 2560, False), (28672, 640, 640, True), (7168, 1280, 1280, True)]
CONVS = [  # B, H, W, Cin, Cout, stride, residual
    (14, 8, 8, 1280, 1280, 1, True), (14, 8, 8, 2560, 1280, 1, False), (14, 16, 16, 1280, 1280, 2, False), (14, 16, 16, 1280, 1280, 1, True),
    (14, 16, 16, 2560, 1280, 1, False), (28, 8, 8, 1280, 1280, 1, True), (28, 8, 8, 2560, 1280, 1, False)]


def run(tag):
    import torch
    from vidseg_diffusion_amd import exact as X
    dev = torch.device("cuda:0")
    outs = []

    def bench(name, f):
        o = f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30):
            f()
        e.record()
        torch.cuda.synchronize()
        print(f"{tag} {name}: {s.elapsed_time(e) / 30 * 1e3:8.1f} us", flush=True)
        outs.append(o.float().cpu())

    for (M, K, N, res) in LINS:
        g = torch.Generator(device="cpu").manual_seed(M + K + N)
        a3 = X.split3(torch.randn((M, K), generator=g).to(dev))
        w3 = X.pack_linear_x(torch.randn((N, K), generator=g) * 0.03, dev)
        b = torch.randn(N, generator=g).to(dev)
        r = torch.randn((M, N), generator=g).to(dev) if res else None
        bench(f"linear {M}x{N}x{3 * K} res{int(res)}", lambda: X.linear_x(a3, w3, b, residual=r))
    for (B, H, W, Ci, Co, st, res) in CONVS:
        g = torch.Generator(device="cpu").manual_seed(B * H + Ci + Co)
        x3 = X.split3(torch.randn((B, H, W, Ci), generator=g).to(dev))
        w3 = X.pack_conv3x3_x(torch.randn((Co, Ci, 3, 3), generator=g) * 0.03, dev)
        b = torch.randn(Co, generator=g).to(dev)
        Ho, Wo = (H + 2 - 3) // st + 1, (W + 2 - 3) // st + 1
        r = torch.randn((B, Ho, Wo, Co), generator=g).to(dev) if res else None
        bench(f"conv B{B} {H}x{W} {Ci}->{Co} s{st} res{int(res)} ({B * Ho * Wo}x{Co}x{27 * Ci})", lambda: X.conv3x3_x(x3, w3, b, stride=st, residual=r))
    torch.save(outs, f"/tmp/xsmall_{tag}.pt")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        import torch
        for tag in ("xsmall0", "xsmall1"):
            subprocess.run([sys.executable, __file__, tag], env={**os.environ, "VIDSEG_GEMM": f"xsmall={tag[-1]}"}, check=True, timeout=900)
        a, b = torch.load("/tmp/xsmall_xsmall0.pt"), torch.load("/tmp/xsmall_xsmall1.pt")
        worst = max(float((x.double() - y.double()).norm() / x.double().norm()) for x, y in zip(a, b))
        print("worst nrms between the two selections", worst)
        assert worst < 4e-6
