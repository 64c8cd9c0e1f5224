"""Pin oracle/unet.py (torch-fp32 restatement of the SD UNet + Euler/CFG sampler) against what the
REFERENCE's own modules produced (tools/gen_golden_unet.py), and check that the product's UNetModel
exposes exactly the reference's state-dict keys/shapes.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle.unet import UNetOracle, euler_sample, legacy_ddpm_sigmas
from vidseg_diffusion_amd import synthetic

G = os.path.join(os.path.dirname(__file__), "golden", "unet_sd_narrow.npz")


@pytest.fixture(scope="module")
def gold():
    z = np.load(G)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def narrow_sd():
    from vidseg_diffusion_amd.unet import UNetModel
    net = UNetModel(**synthetic.SD21_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    return shapes, {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1234).items()}


def test_state_dict_keys_match_reference(gold, narrow_sd):
    shapes, _ = narrow_sd
    assert synthetic.state_dict_signature(shapes) == str(gold["state_dict_signature"])


def test_full_size_key_count():
    from vidseg_diffusion_amd.unet import UNetModel
    net = UNetModel(**synthetic.SD21_FULL)
    sd = net.state_dict()
    assert len(sd) == 686
    assert sum(v.numel() for v in sd.values()) == 865_910_724


def test_unet_forward_matches_reference(gold, narrow_sd):
    _, sd = narrow_sd
    o = UNetOracle(sd)
    out = o.forward(torch.from_numpy(gold["fw_x"]), torch.from_numpy(gold["fw_t"]), torch.from_numpy(gold["fw_ctx"]))
    ref = gold["fw_out"]
    assert np.abs(out.numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
    for k, v in gold.items():
        if k.startswith("fw_output_block_"):
            got = o.taps[k[3:]].float().numpy()
            assert np.abs(got - v.astype(np.float32)).max() <= 2e-3 * np.abs(v.astype(np.float32)).max(), k   # fp16 taps
    from vidseg_diffusion_amd import synthetic
    for name in synthetic.RESBLOCK_FEATURE_PROBES:                       # openaimodel.py:349-350, 367-368
        for which, got in zip(("in", "out"), o.rb_feats[name]):
            v = gold[f"fw_rb_{name}_{which}"].astype(np.float32)
            assert got.shape == v.shape and np.abs(got.numpy() - v).max() <= 2e-3 * np.abs(v).max(), (name, which)


def test_sigmas_match_reference(gold):
    assert np.array_equal(legacy_ddpm_sigmas(25).numpy(), gold["sm_sigmas"])


def test_sampler_matches_reference(gold, narrow_sd):
    _, sd = narrow_sd
    o = UNetOracle(sd)
    xs, taps = [], {}

    def cb(x, i, t):
        xs.append(x.clone().numpy())
        if i == 24:
            for b in (6, 7, 8):
                taps[b] = t[f"output_block_{b}_spatial_self_attn_q"].float().numpy()

    c = torch.from_numpy(gold["sm_c"])
    final = euler_sample(o, torch.from_numpy(gold["sm_latent"]), c, torch.zeros_like(c), num_steps=25, t_start=22, scale=5.0,
                         noise=torch.from_numpy(gold["sm_noise"]), callback=cb)
    ref = gold["sm_x_steps"]
    assert len(xs) == 3
    assert np.abs(np.stack(xs) - ref).max() <= 5e-5 * np.abs(ref).max()
    assert np.abs(final.numpy() - gold["sm_final"]).max() <= 5e-5 * np.abs(ref).max()
    for b in (6, 7, 8):
        r = gold[f"sm_q_block_{b}_time_24"].astype(np.float32)
        assert np.abs(taps[b] - r).max() <= 2e-3 * np.abs(r).max()


def test_video_unet_forward_matches_reference():
    """SVD VideoUNet (VideoResBlock, SpatialVideoTransformer, temporal taps) restated by the oracle vs the reference."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_svd_narrow.npz"))
    g = {k: z[k] for k in z.files}
    from vidseg_diffusion_amd.video_unet import VideoUNet
    net = VideoUNet(**synthetic.SVD_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert synthetic.state_dict_signature(shapes) == str(g["state_dict_signature"])
    sd = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=4321).items()}
    o = UNetOracle(sd)
    out = o.forward(torch.from_numpy(g["fw_x"]), torch.from_numpy(g["fw_t"]), torch.from_numpy(g["fw_ctx"]),
                    y=torch.from_numpy(g["fw_y"]), num_video_frames=int(g["T"]))
    assert np.abs(out.numpy() - g["fw_out"]).max() <= 5e-5 * np.abs(g["fw_out"]).max()
    for k, v in g.items():
        if k.startswith("fw_output_block_"):
            got = o.taps[k[3:]].float().numpy()
            assert got.shape == v.shape, k
            assert np.abs(got - v.astype(np.float32)).max() <= 2e-3 * np.abs(v.astype(np.float32)).max(), k


def test_svd_sampler_matches_reference():
    from oracle.unet import edm_sigmas, euler_sample_svd
    from vidseg_diffusion_amd.video_unet import VideoUNet
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "svd_sampler_narrow.npz"))
    g = {k: z[k] for k in z.files}
    net = VideoUNet(**synthetic.SVD_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=4321).items()}
    np.testing.assert_allclose(edm_sigmas(25).numpy(), g["sm_sigmas"], rtol=1e-6)
    c = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("c_")}
    uc = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("uc_")}
    xs = []
    final = euler_sample_svd(UNetOracle(sd), torch.from_numpy(g["sm_latent"]), c, uc, noise=torch.from_numpy(g["sm_noise"]),
                             callback=lambda x, i, t: xs.append(x.numpy().copy()))
    assert len(xs) == 8
    assert np.abs(np.stack(xs) - g["sm_x_steps"]).max() <= 1e-4 * np.abs(g["sm_x_steps"]).max()
    assert np.abs(final.numpy() - g["sm_final"]).max() <= 1e-4 * np.abs(g["sm_final"]).max()


def _oracle_modulated(o, g, lam):
    lat, c, noise = (torch.from_numpy(g[k]) for k in ("latent", "c", "noise"))
    Fn = lat.shape[0]
    dumps, xts = {}, {}

    def dump_cb(x, i, taps):
        for k, v in taps.items():
            dumps[f"{k}_time_{i}"] = v.clone()
        xts[i] = x.clone()

    feat = euler_sample(o, lat, c, torch.zeros_like(c), noise=noise, callback=dump_cb)
    mod = dict(timesteps=[22], blocks=[7], attn_types=["cross_attn"], masks=torch.from_numpy(g["masks"]), modulate_uc=True,
               inject_types=["spatial_cross_attn_k", "spatial_cross_attn_q", "spatial_self_attn_k", "spatial_self_attn_q"],
               inject_blocks=list(range(1, 12)), dumps=dumps, xt=xts, blend=(22, 23), fh=8, fw=8)
    mod["lambda"] = lam
    xs = []
    final = euler_sample(o, lat, c, torch.zeros_like(c), noise=noise, callback=lambda x, i, t: xs.append(x.numpy().copy()), modulate=mod)
    return feat, np.stack(xs), final


def test_modulated_injected_pass_matches_reference(narrow_sd):
    """a17: injected q/k on decoder blocks, lambda*mask bias on block 7's cross-attention output at step 22, latent blending."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "sd_modulated_narrow.npz"))
    g = {k: z[k] for k in z.files}
    _, sd = narrow_sd
    o = UNetOracle(sd)
    for tag, lam in (("pos", 50.0), ("neg", -50.0)):
        feat, xs, final = _oracle_modulated(o, g, lam)
        assert np.abs(feat.numpy() - g["feat_final"]).max() <= 5e-5 * np.abs(g["feat_final"]).max()
        ref = g[f"mod_{tag}_x_steps"]
        assert xs.shape == ref.shape
        # injected q/k are the fp16 dumps here, fp32 tensors in the golden run -> 5e-3
        assert np.abs(xs - ref).max() <= 5e-3 * np.abs(ref).max(), tag
        assert np.abs(final.numpy() - g[f"mod_{tag}_final"]).max() <= 5e-3 * np.abs(ref).max()


def test_inversion_matches_reference(gold, narrow_sd):
    from oracle.unet import euler_inversion
    _, sd = narrow_sd
    c = torch.from_numpy(gold["sm_c"])
    x, lats = euler_inversion(UNetOracle(sd), torch.from_numpy(gold["sm_latent"]), c, torch.zeros_like(c))
    for got, ref in ((x, gold["inv_final"]), (lats[5], gold["inv_step5"]), (lats[24], gold["inv_step24"])):
        assert np.abs(got.numpy() - ref).max() <= 2e-4 * np.abs(ref).max()


def test_inversion_window_matches_reference(narrow_sd):
    """The inversion variant of the harness (sd_pipeline_vspw.py:233-236, 340-345): sampler.inversion, feature pass from
    t_start = 0, Steps 3 / 3b on the step-24 dumps -- oracle vs the window the reference ran that way (49 narrow evaluations)."""
    import os
    from oracle import analysis as OA
    from oracle.unet import euler_inversion, euler_sample
    _, sd = narrow_sd
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "sd_inversion_window_narrow.npz"))
    g = {k: z[k] for k in z.files}
    Fn, LAT, K = int(g["F"]), int(g["lat"]), int(g["K"])
    o = UNetOracle(sd)
    c = torch.from_numpy(g["c"])
    inv, _ = euler_inversion(o, torch.from_numpy(g["latent"]), c, torch.zeros_like(c))
    assert np.abs(inv.numpy() - g["inverted"]).max() <= 5e-4 * np.abs(g["inverted"]).max()
    xs, taps = {}, {}

    def cb(x, i, t):
        xs[i] = x.numpy().copy()
        if i == 24:
            for b in (6, 7, 8):
                taps[b] = t[f"output_block_{b}_spatial_self_attn_q"].numpy().copy()

    slow = os.environ.get("VIDSEG_SLOW_TESTS", "0") not in ("", "0")   # default: the first feature step only (the other 24 take a minute)
    if not slow:
        class _Stop(Exception):
            pass

        def cb1(x, i, t):
            xs[i] = x.numpy().copy()
            raise _Stop
        try:
            euler_sample(o, inv, c, torch.zeros_like(c), t_start=0, noise=None, callback=cb1)
        except _Stop:
            pass
        assert np.abs(xs[0] - g["x_step0"]).max() <= 2e-3 * np.abs(g["x_step0"]).max()
    else:
        final = euler_sample(o, inv, c, torch.zeros_like(c), t_start=0, noise=None, callback=cb)
        assert sorted(xs) == list(range(25))
        for i in (0, 12, 24):
            assert np.abs(xs[i] - g[f"x_step{i}"]).max() <= 2e-3 * np.abs(g[f"x_step{i}"]).max(), i
        assert np.abs(final.numpy() - g["x_final"]).max() <= 2e-3 * np.abs(g["x_final"]).max()
        for b in (6, 7, 8):
            ref = g[f"q{b}"].astype(np.float32)
            assert np.abs(taps[b].astype(np.float32) - ref).max() <= 4e-3 * np.abs(ref).max(), b
    np.random.seed(17)
    _, lab, _ = OA.match_gt_mask(OA.aggregate_blocks([g["q8"], g["q7"], g["q6"]]), K, np.random.mtrand._rand)
    assert np.array_equal(lab, g["match_labels"].astype(np.int64).reshape(-1))
    th, tw = OA.dense_tracking(g["q7"], Fn, LAT // 2, LAT // 2)
    corr, _ = OA.correct_low_res_mask(lab.reshape(Fn, LAT // 2, LAT // 2), th, tw)
    assert np.array_equal(corr, g["corrected_labels"].astype(np.int64).reshape(-1))


def _oracle_svd_modulated(o, g, lam, t_start=22):
    from oracle.unet import euler_sample_svd
    lat, noise = torch.from_numpy(g["latent"]), torch.from_numpy(g["noise"])
    c = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("c_")}
    uc = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("uc_")}
    dumps, xts = {}, {}

    def dump_cb(x, i, taps):
        for k, v in taps.items():
            dumps[f"{k}_time_{i}"] = v.clone()
        xts[i] = x.clone()

    feat = euler_sample_svd(o, lat, c, uc, t_start=t_start, noise=noise, callback=dump_cb)
    mod = dict(timesteps=[t_start], blocks=[8], attn_types=["self_attn"], layer_types=["spatial", "temporal"],
               masks=torch.from_numpy(g["masks"]), modulate_uc=True,
               inject_types=["temporal_cross_attn_k", "temporal_cross_attn_q", "temporal_self_attn_k", "temporal_self_attn_q"],
               inject_blocks=list(range(1, 12)), dumps=dumps, xt=xts, blend=(t_start, 25), fh=8, fw=8)
    mod["lambda"] = lam
    xs = []
    final = euler_sample_svd(o, lat, c, uc, t_start=t_start, noise=noise, callback=lambda x, i, t: xs.append(x.numpy().copy()),
                             modulate=mod)
    return feat, np.stack(xs), final


def test_svd_modulated_injected_pass_matches_reference():
    """a17 on the VideoUNet: lambda*mask on block 8's spatial AND temporal self-attention outputs at the first step, injected
    temporal q/k on every decoder block, latent blending (svd_pipeline_vspw.py:399-487, video_attention.py:166-216)."""
    from vidseg_diffusion_amd.video_unet import VideoUNet
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "svd_modulated_narrow.npz"))
    g = {k: z[k] for k in z.files}
    net = VideoUNet(**synthetic.SVD_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    o = UNetOracle({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=4321).items()})
    for tag in ("pos", "neg"):
        lam = float(g[f"lam_{tag}"])
        feat, xs, final = _oracle_svd_modulated(o, g, lam)
        assert np.abs(feat.numpy() - g["feat_final"]).max() <= 1e-4 * np.abs(g["feat_final"]).max()
        ref = g[f"mod_{tag}_x_steps"]
        assert xs.shape == ref.shape
        plain_gap = np.abs(ref[-1] - g["feat_final"]).max()
        err = np.abs(xs - ref).max()
        # injected q/k are the fp16 dumps here, fp32 tensors in the golden run
        assert err <= 5e-3 * np.abs(ref).max() and err < 0.2 * plain_gap, (tag, err, plain_gap)
        assert np.abs(final.numpy() - g[f"mod_{tag}_final"]).max() <= 5e-3 * np.abs(ref).max()
