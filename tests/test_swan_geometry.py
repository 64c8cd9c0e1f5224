"""The geometry of the reference's only shipped input: `input_video/swan`, 854x480 -> 832x448 (sd_pipeline_vspw.py:196-214),
latent 56x104, decoder grid 28x52 = the sampler's defaults (sgm/modules/diffusionmodules/sampling.py:239-242).  Tokens per level
5824 / 1456 / 364 / 91: none a multiple of 64 (the GEMM tiles' row granule), the lowest level 7x13 with odd sides.

tests/golden/sd_swan_narrow.npz = the REFERENCE's own run of one 14-frame window at that geometry (tools/gen_golden_swan.py, narrow-width
SD 2.1, K = 20, t_start 22, Steps 3 + 3b).

CPU: the oracle against it (one forward always; the whole window behind VIDSEG_SLOW_TESTS).
GPU: the HIP path against it -- narrow width in both precisions, masks = the reference's in the exact mode -- and the FULL-width UNet at
that geometry as a property test (finite, tap shapes, the analysis of the device's own taps = the oracle's, bit for bit:
tests/test_gpu_c2_window.py::test_full_width_window_at_the_swan_geometry, which shares that module's full-width network)."""
import os

import numpy as np
import pytest
import torch

from vidseg_diffusion_amd import synthetic

G = os.path.join(os.path.dirname(__file__), "golden", "sd_swan_narrow.npz")
F, LH, LW, K = 14, 56, 104, 20
N = (LH // 2) * (LW // 2)


def nrms(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.fixture(scope="module")
def gold():
    z = np.load(G)
    g = {k: z[k] for k in z.files}
    assert (int(g["F"]), int(g["lat_h"]), int(g["lat_w"]), int(g["K"])) == (F, LH, LW, K)
    return g


def swan_inputs(g):
    """The generator's inputs, rebuilt from their seeds and checked against the fixture's hashes."""
    lat = synthetic.region_clip(F, LH, LW, num_regions=K, seed=5, amp=2.0, noise=0.05)
    c = np.random.Generator(np.random.PCG64(12)).standard_normal((F, 77, synthetic.SD21_NARROW["context_dim"])).astype(np.float32)
    noise = torch.randn((F, 4, LH, LW), generator=torch.Generator().manual_seed(300))
    assert synthetic.sha256_of(lat) == str(g["latent_sha256"]) and synthetic.sha256_of(c) == str(g["c_sha256"])
    assert synthetic.sha256_of(noise.numpy()) == str(g["noise_sha256"]), "torch CPU generator stream changed"
    return lat, c, noise


def narrow_state_dict(g):
    from vidseg_diffusion_amd.unet import UNetModel
    net = UNetModel(**synthetic.SD21_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert synthetic.state_dict_signature(shapes) == str(g["state_dict_signature"])
    return net, {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=int(g["weight_seed"])).items()}


# ------------------------------------------------------------------------------------------------------------------ CPU: the oracle
def test_oracle_forward_at_the_swan_geometry(gold):
    from oracle.unet import UNetOracle
    torch.set_grad_enabled(False)
    _, sd = narrow_state_dict(gold)
    _, c, _ = swan_inputs(gold)
    o = UNetOracle(sd)
    ctx = torch.cat([torch.zeros(1, 77, c.shape[2]), torch.from_numpy(c[:1])])
    out = o.forward(torch.from_numpy(gold["fw_x"]), torch.from_numpy(gold["fw_t"]), ctx)
    assert out.shape == (2, 4, LH, LW)
    assert np.abs(out.numpy() - gold["fw_out"]).max() <= 2e-5 * np.abs(gold["fw_out"]).max()
    q7 = o.taps["output_block_7_spatial_self_attn_q"]
    assert tuple(q7.shape) == (2, N, 2 * synthetic.SD21_NARROW["model_channels"])
    ref = gold["fw_q7"].astype(np.float32)
    assert np.abs(q7.half().float().numpy()[:, ::4] - ref).max() <= 2.0 ** -10 * np.abs(ref).max()


@pytest.mark.skipif(os.environ.get("VIDSEG_SLOW_TESTS", "0") in ("", "0"), reason="a minute of oracle evaluations: VIDSEG_SLOW_TESTS=1")
def test_oracle_window_at_the_swan_geometry(gold):
    from oracle import analysis as OA
    from oracle.unet import UNetOracle, euler_sample
    from tools_metrics import matched_iou
    torch.set_grad_enabled(False)
    _, sd = narrow_state_dict(gold)
    lat, c, noise = swan_inputs(gold)
    o = UNetOracle(sd)
    taps = {}

    def cb(x, i, t):
        if i == 24:
            for b in (6, 7, 8):
                taps[b] = t[f"output_block_{b}_spatial_self_attn_q"].half().numpy().copy()

    ct = torch.from_numpy(c)
    final = euler_sample(o, torch.from_numpy(lat), ct, torch.zeros_like(ct), t_start=22, noise=noise, callback=cb)
    assert nrms(final.numpy(), gold["x_final"]) <= 1e-5
    for b in (6, 7, 8):
        ref = gold[f"q{b}_sub"].astype(np.float32)
        assert np.abs(taps[b][F:, ::8, ::2].astype(np.float32) - ref).max() <= 2.0 ** -10 * np.abs(ref).max(), b
    np.random.seed(17)
    _, lab, _ = OA.match_gt_mask(OA.aggregate_blocks([taps[8], taps[7], taps[6]]), K, np.random.mtrand._rand)
    iou, same = matched_iou(lab, gold["match_labels"].astype(np.int64), K)
    assert iou >= 0.99, (iou, same)                          # the oracle's fp16 taps differ from the reference's in a few last bits
    th, tw = OA.dense_tracking(taps[7], F, LH // 2, LW // 2)
    corr, _ = OA.correct_low_res_mask(lab.reshape(F, LH // 2, LW // 2), th, tw)
    iou2, _ = matched_iou(corr, gold["corrected_labels"].astype(np.int64), K)
    assert iou2 >= 0.99, iou2


# --------------------------------------------------------------------------------------------------------------- GPU: the HIP path
def _window_on_device(net, g, precision, refine=True, width_c=None):
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import build_sd_engine, segment_window
    dev = torch.device("cuda:0")
    lat, c, noise = swan_inputs(g)
    if width_c is not None:                                             # the full-width network reads a 1024-wide context
        c = np.random.Generator(np.random.PCG64(12)).standard_normal((F, 77, width_c)).astype(np.float32)
    if getattr(net, "_packed_on", None) is None:
        net.pack(dev)
    net.set_precision(precision)
    eng = build_sd_engine(net, num_steps=25, scale=5.0)
    cc = {"crossattn": torch.from_numpy(c).to(dev)}
    ucc = {"crossattn": torch.zeros_like(cc["crossattn"])}
    FE.FeatureStore.clear()
    FE.MaskStore.clear()
    labels, _ = segment_window(eng, torch.from_numpy(lat).to(dev), cc, ucc, num_masks=K, num_steps=25, t_start=22, seed=17, noise=noise.to(dev),
                               is_refine_mask=refine, feature_folder="/nonexistent/swan", exp_name="w", keep_all_steps=True)
    st = FE.FeatureStore.folder("/nonexistent/swan", "w")
    taps = {b: st[f"output_block_{b}_spatial_self_attn_q_time_24"].cpu().numpy() for b in (6, 7, 8)}
    x = st["xt_time_24"].cpu().numpy()
    shapes = {k: tuple(v.shape) for k, v in st.items() if k.endswith("_time_24")} if hasattr(st, "items") else {}
    FE.FeatureStore.clear()
    FE.MaskStore.clear()
    net.set_precision("fp16")
    return np.asarray(labels).reshape(F, N), taps, x, shapes


@pytest.mark.gpu
def test_narrow_window_at_the_swan_geometry_vs_reference(gold):
    """Narrow-width SD UNet, 14 frames at latent 56x104 (5824 / 1456 / 364 / 91 tokens), both precisions, against the reference's run."""
    from conftest import act_mode
    from tools_metrics import matched_iou
    net, sd = narrow_state_dict(gold)
    net.load_state_dict(sd)
    modes = ("fp16", "exact") if act_mode()[0] == "f16" else ("fp16",)
    for prec in modes:
        labels, taps, x, _ = _window_on_device(net, gold, prec)
        ex = nrms(x, gold["x_final"])
        et = {b: nrms(taps[b][F:, ::8, ::2].astype(np.float32), gold[f"q{b}_sub"].astype(np.float32)) for b in (6, 7, 8)}
        for b in (6, 7, 8):
            assert taps[b].shape == (2 * F, N, 2 * synthetic.SD21_NARROW["model_channels"]) and np.isfinite(taps[b].astype(np.float32)).all()
            assert abs(np.linalg.norm(taps[b][F:].astype(np.float64)) / float(gold[f"q{b}_norm"]) - 1) < 5e-3
        iou, same = matched_iou(labels.reshape(-1), gold["corrected_labels"].astype(np.int64).reshape(-1), K)
        print(f"swan geometry, narrow, {prec}: x nrms {ex:.2e}, taps nrms", {b: f"{e:.2e}" for b, e in et.items()}, f"masks IoU {iou:.4f} identical {same:.4f}")
        if prec == "exact":
            assert ex <= 5e-5 and max(et.values()) <= 1e-4, (ex, et)
            assert iou >= 0.99, (iou, same)
        else:
            assert ex <= act_mode()[1] and max(et.values()) <= act_mode()[1], (ex, et)
    net.release_exact()
