"""A/B of the phased big tile (VIDSEG_GEMM=ph=1) against k_gemm_tile: bit-equality of outputs (same MFMA order), run-to-run
stability (race screen) and time per launch.  usage: python tools/lab/ph_bench.py            (spawns both modes)"""
import os
import subprocess
import sys

sys.path.insert(0, ".")
SHAPES = [  # B, H, W, C0, C1, Cout, stride, up
    (28, 64, 64, 320, 0, 320, 1, 1), (28, 64, 64, 640, 320, 320, 1, 1), (28, 32, 32, 640, 0, 640, 1, 1), (28, 32, 32, 1280, 640, 640, 1, 1),
    (28, 16, 16, 1280, 0, 1280, 1, 1), (28, 16, 16, 1280, 1280, 1280, 1, 1), (28, 32, 32, 320, 0, 640, 1, 1), (28, 16, 16, 1280, 0, 1280, 1, 2),
    (28, 64, 64, 320, 0, 320, 2, 1), (3, 37, 29, 192, 128, 256, 1, 1), (28, 32, 32, 640, 0, 512, 1, 1)]


def run(tag):
    import torch
    from vidseg_diffusion_amd import ops
    dev = torch.device("cuda:0")
    outs = []
    for (B, H, W, C0, C1, Co, st, up) in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(B * H + C0 + Co)
        x0 = torch.randn((B, H, W, C0), generator=g).to(ops.act_dtype()).to(dev)
        x1 = torch.randn((B, H, W, C1), generator=g).to(ops.act_dtype()).to(dev) if C1 else None
        w = ops.pack_conv3x3(torch.randn((Co, C0 + C1, 3, 3), generator=g) * 0.03, dev)
        b = torch.zeros(Co, device=dev)
        f = lambda: ops.conv3x3(x0, w, b, x1=x1, stride=st, up=up)
        o = f()
        same = all(torch.equal(o, f()) for _ in range(6))
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            f()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        M = o.numel() // Co
        fl = 2 * M * Co * 9 * (C0 + C1)
        print(f"{tag} conv B{B} {H}x{W} C{C0}+{C1}->{Co} s{st} up{up}: {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF/s stable={same}", flush=True)
        outs.append(o.cpu())
    torch.save(outs, f"/tmp/ph_{tag}.pt")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        import torch
        for tag, env in (("base", {"VIDSEG_GEMM": "big=2,ph=0"}), ("ph", {"VIDSEG_GEMM": "big=2,ph=1"})):
            subprocess.run([sys.executable, __file__, tag], env={**os.environ, **env}, check=True, timeout=600)
        a, b = torch.load("/tmp/ph_base.pt"), torch.load("/tmp/ph_ph.pt")
        for sh, x, y in zip(SHAPES, a, b):
            print(sh, "bit-equal" if torch.equal(x, y) else f"DIFF max {(x.float()-y.float()).abs().max().item():.4g} frac {(x != y).float().mean().item():.4g}")
