"""Panel width of the tile order (map_tile: the tiles resident together on an XCD form a (32 / gn) x gn block) against time per launch on the
exact mode's shapes: VIDSEG_GEMM="gg=<n>" (GEGLU tile) / "gn=<n>" (every other tile) per child process.   python tools/panel_sweep.py"""
import os
import subprocess
import sys

sys.path.insert(0, ".")
GEGLUS = [(114688, 320, 2560), (28672, 640, 5120), (7168, 1280, 10240), (258048, 320, 2560), (64512, 640, 5120), (16128, 1280, 10240)]
LINS = [(114688, 320, 320, True), (114688, 320, 960, False), (114688, 1280, 320, True), (28672, 640, 640, True), (28672, 2560, 640, True),
        (28672, 640, 1920, False), (7168, 1280, 3840, False)]


def run():
    import torch
    from vidseg_diffusion_amd import exact as X
    from vidseg_diffusion_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)

    def t(f, reps=10):
        for _ in range(3):
            f()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            f()
        e.record()
        torch.cuda.synchronize()
        return 1e3 * s.elapsed_time(e) / reps
    out = []
    if os.environ.get("SWEEP_KIND") == "geglu":
        for (M, K, N) in GEGLUS:
            a3 = X.split3(torch.randn((M, K), generator=g).to(dev))
            w3g, bg, grp = X.pack_geglu_x(torch.randn((N, K), generator=g) * 0.03, torch.randn(N, generator=g), dev)
            out.append(t(lambda: X.geglu_linear_x(a3, w3g, bg, grp)))
            del a3
    else:
        for (M, K, N, res) in LINS:
            a3 = X.split3(torch.randn((M, K), generator=g).to(dev))
            w3 = X.pack_linear_x(torch.randn((N, K), generator=g) * 0.03, dev)
            r = torch.randn((M, N), generator=g).to(dev) if res else None
            b = ops.f32(torch.randn(N, generator=g), dev)
            out.append(t(lambda: X.linear_x(a3, w3, b, residual=r)))
    print("US " + " ".join(f"{u:.1f}" for u in out))


if __name__ == "__main__":
    if os.environ.get("SWEEP_CHILD"):
        run()
        sys.exit(0)
    for kind, key, shapes, widths in (("geglu", "gg", GEGLUS, (0, 1, 2, 3, 4, 5, 8, 10, 20, 40)), ("lin", "gn", LINS, (0, 1, 2, 3, 4, 6))):
        print(f"# {kind}: us per launch; columns = " + "  ".join("x".join(str(v) for v in sh[:3]) for sh in shapes) + "   (0 = the default rule)")
        for wv in widths:
            env = dict(os.environ, SWEEP_CHILD="1", SWEEP_KIND=kind)
            if wv:
                env["VIDSEG_GEMM"] = f"{key}={wv}"
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("US ")]
            print(f"{key}={wv:2d}: " + (line[0][3:] if line else "FAILED " + r.stderr[-300:]), flush=True)
