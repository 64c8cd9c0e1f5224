"""The fp32 CPU oracle against the REFERENCE's labels for full-size windows of the headline clip (tests/golden/c2_window*.npz):
UNet (865.9 M parameters, CFG batch 28 at 64x64 latents, three Euler steps) + aggregation + K-means K = 20 + 4-NN + dense tracking,
end to end.  Measured: identical masks (IoU 1.0000, 100 % of the tokens, Steps 3 and 3b) on all six windows
(profiles/r02_mask_rounding_study.txt) -- the oracle IS the reference at the benchmarked size.  ~80 s and ~25 GB per window on 8
cores, so the default CPU suite skips it: VIDSEG_SLOW_TESTS=1 runs window 0, VIDSEG_SLOW_TESTS=2 every fixture window."""
import glob
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
SLOW = int(os.environ.get("VIDSEG_SLOW_TESTS", "0") or 0)
FIX = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "c2_window*.npz")))


@pytest.mark.skipif(SLOW < 1, reason="full-size oracle run: VIDSEG_SLOW_TESTS=1 (window 0) / 2 (all windows)")
@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[:-4] for p in FIX])
def test_fp32_oracle_reproduces_reference_masks(path):
    from mask_rounding_study import F, K, LAT, Chunked
    from oracle import pipeline as OP
    from tools_metrics import matched_iou
    from vidseg_diffusion_amd import synthetic
    from vidseg_diffusion_amd.unet import UNetModel
    g = np.load(path)
    w = int(g["window_id"]) if "window_id" in g.files else 0
    if w > 0 and SLOW < 2:
        pytest.skip("VIDSEG_SLOW_TESTS=2 runs the further windows")
    torch.set_grad_enabled(False)
    cfg = dict(synthetic.SD21_FULL)
    shapes = {k: tuple(v.shape) for k, v in UNetModel(**cfg).state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1234, zero_gain=float(g["zero_gain"])).items()}
    c, uc = synthetic.sd_conditioning(F, context_dim=cfg["context_dim"], seq=77, seed=1)
    lat = synthetic.headline_latent(F, LAT, LAT, window_id=w)
    noise = torch.randn((F, 4, LAT, LAT), generator=torch.Generator().manual_seed(100 + w))
    assert synthetic.sha256_of(lat) == str(g["latent_sha256"]) and synthetic.sha256_of(noise.numpy()) == str(g["noise_sha256"])
    res = OP.segment_window(Chunked(sd), torch.from_numpy(lat), torch.from_numpy(c), torch.from_numpy(uc), noise, num_masks=K,
                            t_start=int(g["t_start"]), seed=int(g["seed"]), is_refine_mask=True)
    for name, key in (("match_labels", "match_labels"), ("labels", "corrected_labels")):
        iou, exact = matched_iou(res[name], g[key].astype(np.int64), K)
        assert iou >= 0.9999 and exact >= 0.9999, (w, name, iou, exact)
