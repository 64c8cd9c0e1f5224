"""k_gemm_phx (256 x 320 tile, each plane of the split operands staged once) against k_gemm_ph<5> on the 3K axis, on the exact SVD window's
shapes whose M fills whole rounds of 256-row tiles: results, run-to-run bit stability, time per launch.   usage: python tools/phx_bench.py"""
import os
import subprocess
import sys

sys.path.insert(0, ".")
CONVS = [(28, 36, 64, 640, 640, 1, 1, True), (28, 36, 64, 1920, 640, 1, 1, False), (28, 18, 32, 1280, 1280, 1, 1, True), (28, 18, 32, 2560, 1280, 1, 1, False),
         (28, 18, 32, 1280, 1280, 1, 2, False), (28, 72, 128, 320, 320, 2, 1, False)]   # B, H, W, Cin, Cout, stride, up, residual
LINS = [(64512, 640, 640, True), (64512, 2560, 640, True), (16128, 1280, 1280, True), (16128, 5120, 1280, True), (64512, 640, 1920, False),
        (16128, 1280, 3840, False), (1000, 640, 640, False)]   # M, K, N, residual


def run(tag):
    import torch
    from vidseg_diffusion_amd import exact as X
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
        x3 = X.split3(torch.randn((B, H, W, Ci), generator=g).to(dev))
        w3 = X.pack_conv3x3_x(torch.randn((Co, Ci, 3, 3), generator=g) * 0.03, dev)
        b = torch.randn(Co, generator=g).to(dev)
        Ho, Wo = (H * up + 2 - 3) // st + 1, (W * up + 2 - 3) // st + 1
        r = torch.randn((B, Ho, Wo, Co), generator=g).to(dev) if res else None
        bench(f"conv B{B} {H}x{W} {Ci}->{Co} s{st} up{up} res{int(res)}", lambda: X.conv3x3_x(x3, w3, b, stride=st, up=up, residual=r), 2.0 * B * Ho * Wo * Co * 27 * Ci)
    for (M, K, N, res) in LINS:
        g = torch.Generator(device="cpu").manual_seed(M + K + N)
        a = torch.randn((M, K), generator=g)
        w = torch.randn((N, K), generator=g) * 0.03
        a3, w3 = X.split3(a.to(dev)), X.pack_linear_x(w, dev)
        b = torch.randn(N, generator=g).to(dev)
        r = torch.randn((M, N), generator=g).to(dev) if res else None
        bench(f"linear {M}x{N}x{K} res{int(res)}", lambda: X.linear_x(a3, w3, b, residual=r), 2.0 * M * N * 3 * K)
        if tag == "phx" and M <= 20000:
            ref = a.double() @ w.double().t() + b.cpu().double() + (r.cpu().double() if res else 0)
            e = float((outs[-1].double() - ref).abs().max() / ref.abs().max())
            print(f"    vs float64: max err {e:.2e}", flush=True)
            assert e < 5e-6, e
    torch.save(outs, f"/tmp/phx_{tag}.pt")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        import torch
        for tag, env in (("ph", {"VIDSEG_GEMM": "phx=0"}), ("phx", {"VIDSEG_GEMM": "phx=1"})):
            subprocess.run([sys.executable, __file__, tag], env={**os.environ, **env}, check=True, timeout=900)
        a, b = torch.load("/tmp/phx_ph.pt"), torch.load("/tmp/phx_phx.pt")
        worst = 0.0
        for i, (x, y) in enumerate(zip(a, b)):
            e = float((x.double() - y.double()).norm() / x.double().norm())
            worst = max(worst, e)
            print(f"case {i}: phx vs ph nrms {e:.2e} max|d| {float((x - y).abs().max()):.2e}")
        print("worst", worst)
        assert worst < 4e-6
