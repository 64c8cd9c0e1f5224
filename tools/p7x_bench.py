"""k_gemm_p7x (each plane of the split operands staged once) against k_gemm_p7 on the 3K axis: results (same products, another fp32
summation order: ~1e-7), run-to-run bit stability and time per launch on the exact mode's shapes.  usage: python tools/p7x_bench.py"""
import os
import subprocess
import sys

sys.path.insert(0, ".")
CONVS = [  # B, H, W, Cin, Cout, stride, up, residual
    (28, 64, 64, 320, 320, 1, 1, True), (28, 64, 64, 960, 320, 1, 1, False), (28, 32, 32, 640, 640, 1, 1, True), (28, 32, 32, 1920, 640, 1, 1, False),
    (28, 16, 16, 1280, 1280, 1, 1, True), (28, 16, 16, 2560, 1280, 1, 1, False), (28, 8, 8, 1280, 1280, 1, 1, True), (28, 64, 64, 320, 320, 2, 1, False),
    (28, 16, 16, 1280, 1280, 1, 2, False), (14, 32, 32, 640, 640, 1, 1, True), (3, 37, 29, 192, 320, 1, 1, False)]
GEGLUS = [(114688, 320, 2560), (28672, 640, 5120), (7168, 1280, 10240), (57344, 320, 2560), (3584, 1280, 10240)]   # M, K, 2 * inner
LINS = [  # M, K, N, residual
    (114688, 320, 320, True), (114688, 1280, 320, True), (28672, 640, 640, True), (28672, 2560, 640, True), (7168, 1280, 1280, True),
    (7168, 5120, 1280, True), (114688, 320, 960, False), (28672, 640, 1920, False), (7168, 1280, 3840, False), (2156, 1024, 1280, False), (999, 320, 640, False)]


def run(tag):
    import torch
    from vidseg_diffusion_amd import exact as X
    from vidseg_diffusion_amd import ops
    dev = torch.device("cuda:0")
    outs = []

    def bench(name, f, flops):
        o = f()
        same = all(torch.equal(o, f()) for _ in range(5))
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            f()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        print(f"{tag} {name}: {ms * 1e3:8.1f} us {flops / ms / 1e9:7.1f} TF/s (3 products counted) stable={same}", flush=True)
        outs.append(o.float().cpu())

    for (B, H, W, Ci, Co, st, up, res) in CONVS:
        g = torch.Generator(device="cpu").manual_seed(B * H + Ci + Co)
        x = torch.randn((B, H, W, Ci), generator=g).to(dev)
        w3 = X.pack_conv3x3_x(torch.randn((Co, Ci, 3, 3), generator=g) * 0.03, dev)
        b = torch.randn(Co, generator=g).to(dev)
        x3 = X.split3(x)
        Ho, Wo = (H * up + 2 - 3) // st + 1, (W * up + 2 - 3) // st + 1
        r = torch.randn((B, Ho, Wo, Co), generator=g).to(dev) if res else None
        bench(f"conv B{B} {H}x{W} {Ci}->{Co} s{st} up{up} res{int(res)}", lambda: X.conv3x3_x(x3, w3, b, stride=st, up=up, residual=r),
              2.0 * B * Ho * Wo * Co * 27 * Ci)
    for (M, K, N, res) in LINS:
        g = torch.Generator(device="cpu").manual_seed(M + K + N)
        a = torch.randn((M, K), generator=g).to(dev)
        w3 = X.pack_linear_x(torch.randn((N, K), generator=g) * 0.03, dev)
        b = torch.randn(N, generator=g).to(dev)
        a3 = X.split3(a)
        r = torch.randn((M, N), generator=g).to(dev) if res else None
        bench(f"linear {M}x{N}x{K} res{int(res)}", lambda: X.linear_x(a3, w3, b, residual=r), 2.0 * M * N * 3 * K)
        if tag == "p7x" and M <= 30000:
            ref = (a.double() @ (w3[:, :K].double() + w3[:, 2 * K:].double()).t().to(dev) + b.double() + (r.double() if res else 0)).float().cpu()
            e = float((outs[-1].double() - ref.double()).norm() / ref.double().norm())
            print(f"    vs float64 of the split operands: nrms {e:.2e}", flush=True)
            assert e < 2e-6, e
    for (M, K, N) in GEGLUS:
        g = torch.Generator(device="cpu").manual_seed(M + K + N)
        a3 = X.split3(torch.randn((M, K), generator=g).to(dev))
        w3g, bg, grp = X.pack_geglu_x(torch.randn((N, K), generator=g) * 0.03, torch.randn(N, generator=g), dev)
        bench(f"GEGLU {M}x{N}x{K} (groups of {grp})", lambda: X.geglu_linear_x(a3, w3g, bg, grp), 2.0 * M * N * 3 * K)
    torch.save(outs, f"/tmp/p7x_{tag}.pt")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        import torch
        modes = (("p7", {"VIDSEG_GEMM": "p7x=0"}), ("p7x", {"VIDSEG_GEMM": "p7x=1"}))
        for tag, env in modes:
            subprocess.run([sys.executable, __file__, tag], env={**os.environ, **env}, check=True, timeout=900)
        a, b = torch.load(f"/tmp/p7x_{modes[0][0]}.pt"), torch.load(f"/tmp/p7x_{modes[1][0]}.pt")
        worst = 0.0
        for i, (x, y) in enumerate(zip(a, b)):
            e = float((x.double() - y.double()).norm() / x.double().norm())
            worst = max(worst, e)
            print(f"case {i}: p7x vs p7 nrms {e:.2e} max|d| {float((x - y).abs().max()):.2e}")
        print("worst", worst)
        assert worst < 4e-6                                          # two fp32 summation orders of the same products, each ~5e-7 from float64
