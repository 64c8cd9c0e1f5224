import os, sys
import numpy as np, torch, torch.nn.functional as TF
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vidseg_diffusion_amd import exact as X  # noqa: E402
dev = torch.device("cuda:0")
def rnd(shape, seed, scale=1.0):
    return torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal(shape).astype(np.float32)) * scale
for (B, H, Nq, Nk) in ((1, 10, 128, 1024), (1, 2, 1024, 1024), (1, 10, 1024, 1024), (1, 10, 1024, 1024), (2, 5, 1024, 1024), (1, 10, 1024, 256)):
    C = H * 64
    q, k, vv = rnd((B, Nq, C), 19), rnd((B, Nk, C), 20), rnd((B, Nk, C), 21)
    kv = torch.cat([k, vv], -1).to(dev)
    out = X.attention_mfma(q.to(dev), kv, H, B, Nq, Nk).cpu()
    hd = lambda t, n: t.double().view(B, n, H, 64).transpose(1, 2)            # noqa: E731
    ref = TF.scaled_dot_product_attention(hd(q, Nq), hd(k, Nk), hd(vv, Nk)).transpose(1, 2).reshape(B, Nq, C)
    d = (out.double() - ref).abs().view(B, Nq, H, 64)
    rowerr = d.amax(-1) / ref.abs().max()
    bad = (rowerr > 2e-6).nonzero()
    print(f"B={B} H={H} Nq={Nq} Nk={Nk}: max {float(rowerr.max()):.2e}, bad (b, q, h) rows: {len(bad)}")
    S = (hd(q, Nq) @ hd(k, Nk).transpose(-1, -2)) / 8 * 1.4426950408889634   # [B,H,Nq,Nk] log2 units
    for (b, qi, h) in bad[:12].tolist():
        s = S[b, h, qi]
        tmax = s.view(-1, 64).amax(-1)
        run = torch.cummax(tmax, 0)[0]
        print(f"   b={b} q={qi} (block {qi // 128} wave {(qi % 128) // 32} lane {qi % 32}) h={h} err {float(rowerr[b, qi, h]):.2e} n_bad_d={int((d[b, qi, h] / ref.abs().max() > 2e-6).sum())}"
              f" rowmax {float(s.max()):.2f} at key {int(s.argmax())} tile maxima {[round(float(x), 1) for x in tmax[:16]]}")
