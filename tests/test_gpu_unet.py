"""GPU parity of the SD UNet / sampler / one-window pipeline (HIP kernels behind the reference's plug-in API)
against the reference-produced golden vectors and the oracle.

Tolerance statement (floating point, 16-bit MFMA path; fp16 activations by default like the reference's CUDA autocast, bf16
with -DVIDSEG_ACT_BF16).  Each kernel alone is within 2^-8|ref| + 1e-3 max|ref| of an fp32 evaluation of the same 16-bit
inputs (tests/test_gpu_ops.py).  Through the whole ~60-layer network with random synthetic weights the storage *format*
itself moves the output relative to the fp32 reference (fp16: ~2e-3 normalised rms, bf16: ~1.5e-2) -- measured below by the
oracle's rounding mode -- so the network-level requirement is: the HIP path deviates from the fp32 reference by no more
than 1.5x what the format's rounding alone does (plus a small floor), and by less than `conftest.act_mode()`'s absolute bound
(4e-3 for fp16, 4e-2 for bf16, normalised rms).  Masks: IoU >= 0.99 vs the all-fp32 oracle in the fp16 build."""
import os

import numpy as np
import pytest
import torch

from vidseg_diffusion_amd import synthetic
from conftest import act_mode

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden", "unet_sd_narrow.npz")


def nrms(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available()
    from vidseg_diffusion_amd import _lib
    from vidseg_diffusion_amd.unet import UNetModel
    _lib.lib()
    dev = torch.device("cuda:0")
    z = np.load(G)
    gold = {k: z[k] for k in z.files}
    net = UNetModel(**synthetic.SD21_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, 1234).items()}
    net.load_state_dict(sd)
    return dev, gold, net, sd


def test_unet_forward_vs_reference(env):
    from oracle.unet import UNetOracle
    dev, g, net, sd = env
    x, t, ctx = (torch.from_numpy(g[k]) for k in ("fw_x", "fw_t", "fw_ctx"))
    plain = net(x.to(dev), timesteps=t.to(dev), context=ctx.to(dev)).cpu().numpy()
    net.stash_resblock_features(True)                                    # the reference's ResBlock stash, off by default
    try:
        out = net(x.to(dev), timesteps=t.to(dev), context=ctx.to(dev)).cpu().numpy()
        assert np.array_equal(out, plain), "the feature stash must not change the network's output"
        for name in synthetic.RESBLOCK_FEATURE_PROBES:                   # openaimodel.py:349-350 (before the emb add), 367-368 (before the skip)
            rb = net.get_submodule(name)
            for which, got in (("in", rb.in_layers_features), ("out", rb.out_layers_features)):
                ref = g[f"fw_rb_{name}_{which}"].astype(np.float32)
                assert got.dtype == torch.float16 and tuple(got.shape) == ref.shape, (name, which, got.shape, ref.shape)
                e = nrms(got.float().cpu().numpy(), ref)
                assert e < act_mode()[1], (name, which, e)
    finally:
        net.stash_resblock_features(False)
    assert net.get_submodule("middle_block.0").in_layers_features is None
    fmt = nrms(UNetOracle(sd, round_bf16=act_mode()[0]).forward(x, t, ctx).numpy(), g["fw_out"])      # what bf16 alone costs
    err = nrms(out, g["fw_out"])
    assert err < act_mode()[1] and err <= 1.5 * fmt + 1e-3, (err, fmt)
    for b in range(3, 12):                                                                     # reference dump protocol
        blk = net.output_blocks[b]
        assert len(blk) > 1 and "SpatialTransformer" in str(type(blk[1]))
        tb = blk[1].transformer_blocks[0]
        for nm, a in (("self", tb.attn1), ("cross", tb.attn2)):
            for w in ("q", "k"):
                ref = g[f"fw_output_block_{b}_spatial_{nm}_attn_{w}"].astype(np.float32)
                got = getattr(a, w)
                assert got.dtype == torch.float16 and tuple(got.shape) == ref.shape
                assert nrms(got.float().cpu().numpy(), ref) < act_mode()[1], (b, nm, w)
    for b in range(0, 3):
        assert len(net.output_blocks[b]) == 1 or "SpatialTransformer" not in str(type(net.output_blocks[b][1]))


def test_sampler_steps_vs_reference(env):
    from vidseg_diffusion_amd.pipeline import build_sd_engine
    dev, g, net, sd = env
    eng = build_sd_engine(net, num_steps=25, scale=5)
    # 1-ulp fp32 differences in `** 0.5` between host CPUs are expected (torch vectorised pow)
    np.testing.assert_allclose(eng.sampler.discretization(25).numpy(), g["sm_sigmas"], rtol=5e-7, atol=0)
    c = {"crossattn": torch.from_numpy(g["sm_c"]).to(dev)}
    uc = {"crossattn": torch.zeros_like(c["crossattn"])}
    lat = torch.from_numpy(g["sm_latent"]).to(dev)
    noised = eng.sampler.add_noise(lat, cond=c, uc=uc, num_steps=25, noise_level=22, noise=torch.from_numpy(g["sm_noise"]).to(dev))
    assert np.abs(noised.cpu().numpy() - g["sm_noised"]).max() <= 1e-6 * np.abs(g["sm_noised"]).max() + 1e-7
    xs, taps = [], {}

    def cb(xt, i):
        xs.append(xt.cpu().numpy())
        if i == 24:
            for b in (6, 7, 8):
                taps[b] = net.output_blocks[b][1].transformer_blocks[0].attn1.q.float().cpu().numpy()

    def denoiser(inp, sigma, cc, is_modulate_step=False, is_injected_step=False, modulate_params=None):
        return eng.denoiser(eng.model, inp, sigma, cc)

    final = eng.sampler(denoiser, noised.clone(), cond=c, uc=uc, img_callback=cb, t_start=22)
    assert len(xs) == 3
    for i in range(3):
        assert nrms(xs[i], g["sm_x_steps"][i]) < act_mode()[1], i
    assert nrms(final.cpu().numpy(), g["sm_final"]) < act_mode()[1]
    for b in (6, 7, 8):
        assert nrms(taps[b], g[f"sm_q_block_{b}_time_24"].astype(np.float32)) < act_mode()[1]


def test_window_pipeline_vs_oracle(env):
    """Steps 1-3b end to end on one 4-frame window; masks compared with the CPU oracle's (IoU), and -- the
    bit-exact part -- the analysis stage re-run by the oracle on the HIP path's own taps must give identical ids."""
    from oracle import analysis as OA
    from oracle import pipeline as OP
    from oracle.unet import UNetOracle
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import build_sd_engine, segment_window
    dev, g, net, sd = env
    Fn, K = 4, 5
    lat = synthetic.latent_clip(Fn, 16, 16, seed=3)
    c = np.random.Generator(np.random.PCG64(4)).standard_normal((Fn, 7, 64)).astype(np.float32)
    noise = torch.from_numpy(np.random.Generator(np.random.PCG64(5)).standard_normal(lat.shape).astype(np.float32))
    ref = OP.segment_window(UNetOracle(sd), torch.from_numpy(lat), torch.from_numpy(c), torch.zeros(Fn, 7, 64), noise,
                            num_masks=K, is_refine_mask=True, seed=17)
    eng = build_sd_engine(net)
    FE.FeatureStore.clear()
    FE.MaskStore.clear()
    cc = {"crossattn": torch.from_numpy(c).to(dev)}
    uc = {"crossattn": torch.zeros_like(cc["crossattn"])}
    labels, st = segment_window(eng, torch.from_numpy(lat).to(dev), cc, uc, num_masks=K, is_refine_mask=True, seed=17,
                                noise=noise.to(dev), feature_folder="/nonexistent/vs", exp_name="w")
    # bit-exact: oracle analysis on the device taps
    store = FE.FeatureStore.folder("/nonexistent/vs", "w")
    taps = {b: store[f"output_block_{b}_spatial_self_attn_q_time_24"].cpu().numpy() for b in (6, 7, 8)}
    np.random.seed(17)
    _, lab_o, _ = OA.match_gt_mask(OA.aggregate_blocks([taps[8], taps[7], taps[6]]), K, np.random.mtrand._rand)
    th, tw = OA.dense_tracking(taps[7], Fn, 8, 8)
    corr, _ = OA.correct_low_res_mask(lab_o.reshape(Fn, 8, 8), th, tw)
    assert np.array_equal(labels.reshape(-1), corr)
    # floating-point part: masks agree with the all-CPU oracle up to a label permutation
    from tools_metrics import matched_iou
    iou, exact = matched_iou(labels, ref["labels"], K)
    print("pipeline mask IoU vs fp32 oracle", iou, "exact", exact)
    # 256 tokens in 5 clusters: one borderline token moves a cluster's IoU by 1-2 %, and which tokens are borderline depends on the
    # fp32 summation order of the convs (tap-major K: 256/256 identical; chunk-major K: 253/256).  The >= 0.99 bar of the north star
    # is asserted on BASELINE configs[0] at full width (test_c1_full_width_window_vs_oracle, 4096 tokens).
    assert (iou >= 0.97 and exact >= 0.98) if act_mode()[0] == "f16" else iou >= 0.90


def test_video_unet_forward_vs_reference():
    """SVD VideoUNet on the GPU vs the reference-produced golden (narrow width, T=3): output, spatial and temporal taps."""
    from oracle.unet import UNetOracle
    from vidseg_diffusion_amd.video_unet import VideoUNet
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_svd_narrow.npz"))
    g = {k: z[k] for k in z.files}
    net = VideoUNet(**synthetic.SVD_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=4321).items()}
    net.load_state_dict(sd)
    T = int(g["T"])
    x, t, ctx, y = (torch.from_numpy(g[k]) for k in ("fw_x", "fw_t", "fw_ctx", "fw_y"))
    out = net(x.to(dev), timesteps=t.to(dev), context=ctx.to(dev), y=y.to(dev), num_video_frames=T,
              image_only_indicator=torch.zeros(2, T)).cpu().numpy()
    fmt = nrms(UNetOracle(sd, round_bf16=act_mode()[0]).forward(x, t, ctx, y=y, num_video_frames=T).numpy(), g["fw_out"])
    err = nrms(out, g["fw_out"])
    print("video unet nrms", err, "16-bit format", fmt)
    assert err < act_mode()[1] and err <= 1.5 * fmt + 5e-3, (err, fmt)
    for i in (3, 7, 8, 11):
        blk = net.output_blocks[i]
        assert "SpatialVideoTransformer" in str(type(blk[1]))
        pairs = [("spatial_self_attn_q", blk[1].transformer_blocks[0].attn1.q), ("temporal_self_attn_q", blk[1].time_stack[0].attn1.q),
                 ("temporal_self_attn_k", blk[1].time_stack[0].attn1.k), ("temporal_cross_attn_k", blk[1].time_stack[0].attn2.k)]
        for name, got in pairs:
            ref = g[f"fw_output_block_{i}_{name}"].astype(np.float32)
            assert tuple(got.shape) == ref.shape, (i, name, got.shape, ref.shape)
            assert nrms(got.float().cpu().numpy(), ref) < act_mode()[1], (i, name)


def test_svd_sampler_steps_vs_reference():
    """8 Euler steps (17..24) of the SVD engine (VideoUNet + Denoiser/VScalingWithEDMcNoise + LinearPredictionGuider +
    concat/vector conditioning) against the reference's own trajectory."""
    from vidseg_diffusion_amd.pipeline import build_svd_engine
    from vidseg_diffusion_amd.video_unet import VideoUNet
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "svd_sampler_narrow.npz"))
    g = {k: z[k] for k in z.files}
    net = VideoUNet(**synthetic.SVD_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=4321).items()})
    Fn = g["sm_latent"].shape[0]
    eng = build_svd_engine(net, num_frames=Fn)
    c = {k[2:]: torch.from_numpy(v).to(dev) for k, v in g.items() if k.startswith("c_")}
    uc = {k[3:]: torch.from_numpy(v).to(dev) for k, v in g.items() if k.startswith("uc_")}
    noised = eng.sampler.add_noise(torch.from_numpy(g["sm_latent"]).to(dev), cond=c, uc=uc, num_steps=25, noise_level=17,
                                   noise=torch.from_numpy(g["sm_noise"]).to(dev))
    assert np.abs(noised.cpu().numpy() - g["sm_noised"]).max() <= 1e-5 * np.abs(g["sm_noised"]).max() + 1e-7
    xs, taps = [], {}
    extra = {"image_only_indicator": torch.zeros(2, Fn), "num_video_frames": Fn}

    def cb(xt, i):
        xs.append(xt.cpu().numpy())
        if i == 24:
            taps["q8"] = net.output_blocks[8][1].transformer_blocks[0].attn1.q.float().cpu().numpy()
            taps["tq8"] = net.output_blocks[8][1].time_stack[0].attn1.q.float().cpu().numpy()

    final = eng.sampler(lambda inp, s, cc, **k: eng.denoiser(eng.model, inp, s, cc, **extra), noised.clone(), cond=c, uc=uc,
                        img_callback=cb, t_start=17)
    assert len(xs) == 8
    errs = [nrms(xs[i], g["sm_x_steps"][i]) for i in range(8)]
    print("svd step nrms", [round(e, 4) for e in errs])
    assert max(errs) < act_mode()[1]
    assert nrms(taps["q8"], g["sm_q8"].astype(np.float32)) < act_mode()[1]
    assert nrms(taps["tq8"], g["sm_tq8"].astype(np.float32)) < act_mode()[1]


def test_modulated_injected_pass_vs_reference(env):
    """a17 on the GPU: feature pass fills the in-HBM FeatureStore, then the reference-style modulated pass
    (is_modulate, injected q/k from the store, lambda*mask on block 7 cross-attn output, latent blending)."""
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import build_sd_engine, save_feature_maps
    dev, _, net, sd = env
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "sd_modulated_narrow.npz"))
    g = {k: z[k] for k in z.files}
    eng = build_sd_engine(net)
    Fn = g["latent"].shape[0]
    c = {"crossattn": torch.from_numpy(g["c"]).to(dev)}
    uc = {"crossattn": torch.zeros_like(c["crossattn"])}
    noised = eng.sampler.add_noise(torch.from_numpy(g["latent"]).to(dev), cond=c, uc=uc, num_steps=25, noise_level=22,
                                   noise=torch.from_numpy(g["noise"]).to(dev))
    FE.FeatureStore.clear()
    base, exp = "/nonexistent/vs_mod", "exp"

    def denoiser(inp, sigma, cc, is_modulate_step=False, is_injected_step=False, modulate_params=None):
        return eng.denoiser(eng.model, inp, sigma, cc, is_modulate_step=is_modulate_step, is_injected_step=is_injected_step,
                            modulate_params=modulate_params)

    feat = eng.sampler(denoiser, noised.clone(), cond=c, uc=uc, t_start=22,
                       img_callback=lambda xt, i: save_feature_maps(eng, base, exp, i, xt=xt))
    assert nrms(feat.cpu().numpy(), g["feat_final"]) < act_mode()[1]
    for tag, lam in (("pos", 50.0), ("neg", -50.0)):
        mp = {"feature_masks": [torch.from_numpy(m).to(dev) for m in g["masks"]], "modulate_block_idx": [7],
              "modulate_layer_type": ["spatial"], "modulate_attn_type": ["cross_attn"], "modulate_timestep": [22],
              "modulate_schedule": "constant", "modulate_lambda_start": lam, "modulate_lambda_end": lam, "num_frames": Fn,
              "modulate_uc": True, "is_injected_features": True,
              "injected_feature_types": ["spatial_cross_attn_k", "spatial_cross_attn_q", "spatial_self_attn_k", "spatial_self_attn_q"],
              "injected_block_types": ["output"], "input_block_indices": [3, 4, 5, 6, 7, 8, 9, 10, 11],
              "output_block_indices": list(range(1, 12)), "feature_folder": base, "exp_name": exp, "injected_features_group": {},
              "modulate_layer_frames": {}, "modulate_block_frames": {}, "modulate_timestep_frames": {}, "modulate_lambda_layers": {},
              "latent_mask_start": 22, "latent_mask_end": 23}
        xs = []
        final = eng.sampler(denoiser, noised.clone(), cond=c, uc=uc, img_callback=lambda xt, i: xs.append(xt.cpu().numpy()),
                            is_modulate=True, modulate_params=mp, t_start=22, is_latent_blending=True, feature_height=8, feature_width=8)
        ref = g[f"mod_{tag}_x_steps"]
        assert len(xs) == ref.shape[0]
        errs = [nrms(xs[i], ref[i]) for i in range(len(xs))]
        print("modulated", tag, "step nrms", [round(e, 4) for e in errs])
        assert max(errs) < act_mode()[1]
        # the modulation must actually have moved the sample the way the reference's did
        d_ref = g[f"mod_{tag}_final"] - g["feat_final"]
        d_got = final.cpu().numpy() - feat.cpu().numpy()
        print("modulated", tag, "delta nrms", round(nrms(d_got, d_ref), 4), "delta/ref", round(float(np.abs(d_ref).mean() / np.abs(g["feat_final"]).mean()), 4))
        assert nrms(d_got, d_ref) < (5e-2 if act_mode()[0] == "f16" else 0.35), nrms(d_got, d_ref)


def _format_errors():
    """What the 16-bit storage format alone costs on the inversion trajectories (the oracle in its rounding mode against the reference's
    fp32 goldens), per build format: tests/golden/format_errors.json, tools/gen_format_errors.py."""
    import json
    with open(os.path.join(os.path.dirname(G), "format_errors.json")) as fh:
        return json.load(fh)[act_mode()[0]]


def test_inversion_vs_reference(env):
    """a3b: --inversion_type inversion (sampling.py:264-296): 25 ascending Euler steps, first one skips the network."""
    from vidseg_diffusion_amd.pipeline import build_sd_engine
    dev, g, net, sd = env
    eng = build_sd_engine(net)
    c = {"crossattn": torch.from_numpy(g["sm_c"]).to(dev)}
    uc = {"crossattn": torch.zeros_like(c["crossattn"])}
    x, lats = eng.sampler.inversion(lambda inp, s, cc, **k: eng.denoiser(eng.model, inp, s, cc), torch.from_numpy(g["sm_latent"]).to(dev),
                                    cond=c, uc=uc, num_steps=25)
    assert len(lats) == 26
    assert lats[-1] is x                                             # SAM:294 rescales in place: the list's last entry IS the result
    assert nrms(lats[5].cpu().numpy(), g["inv_step5"]) < act_mode()[1]
    # 24 network steps up to sigma = 14.6 with random weights amplify rounding chaotically: the bf16 FORMAT alone (oracle
    # in bf16-rounding mode) ends 14 % away from the fp32 reference; the HIP path must not be worse than that.
    fmt = _format_errors()["inversion_final"]                        # the oracle in its rounding mode, tools/gen_format_errors.py
    err = nrms(x.cpu().numpy(), g["inv_final"])
    print("inversion nrms", err, "16-bit format", fmt)
    assert err <= 1.3 * fmt + 2e-3, (err, fmt)                       # measured (fp16 build): 2.33e-2 against 1.95e-2 for the format alone


def test_inversion_window_vs_reference(env):
    """`--inversion_type inversion` through the harness (sd_pipeline_vspw.py:233-236, 340-345, 357): sampler.inversion, then the
    feature pass from t_start = 0 with the dump callback at all 25 steps, Steps 3 / 3b on the step-24 dumps -- against ONE window the
    reference itself ran that way (tests/golden/sd_inversion_window_narrow.npz, tools/gen_golden_inversion_window.py).  49 network
    evaluations with random weights amplify rounding, so the latent / tap bars are relative to what the storage format alone costs
    (the oracle in its rounding mode); given the device's taps the masks are the oracle's bit for bit."""
    from oracle import analysis as OA
    from oracle.unet import UNetOracle, euler_inversion, euler_sample
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import build_sd_engine, segment_window
    dev, _, net, sd = env
    z = np.load(os.path.join(os.path.dirname(G), "sd_inversion_window_narrow.npz"))
    g = {k: z[k] for k in z.files}
    Fn, LAT, K = int(g["F"]), int(g["lat"]), int(g["K"])
    eng = build_sd_engine(net, num_steps=25, scale=5.0)
    c = {"crossattn": torch.from_numpy(g["c"]).to(dev)}
    uc = {"crossattn": torch.zeros_like(c["crossattn"])}
    FE.FeatureStore.clear()
    FE.MaskStore.clear()
    labels, _ = segment_window(eng, torch.from_numpy(g["latent"]).to(dev), c, uc, num_masks=K, num_steps=25, t_start=22, seed=17,
                               is_refine_mask=True, feature_folder="/nonexistent/inv", exp_name="w", keep_all_steps=True,
                               inversion_type="inversion")
    store = FE.FeatureStore.folder("/nonexistent/inv", "w")
    for i in range(25):                                                  # t_start = 0: the callback dumps at EVERY step (SDP:235-236, 103-105)
        assert f"xt_time_{i}" in store and f"output_block_7_spatial_self_attn_q_time_{i}" in store, i
    # the storage format's own cost on this 49-evaluation trajectory: the CPU oracle with every operand / activation rounded, computed
    # once by tools/gen_format_errors.py (tests/golden/format_errors.json; a minute of oracle evaluations per run before round 5)
    fe = _format_errors()
    for i in (0, 12, 24):
        err, fmt = nrms(store[f"xt_time_{i}"].cpu().numpy(), g[f"x_step{i}"]), fe[f"window_x_step{i}"]
        print(f"inversion window: x after step {i}: nrms {err:.3e} (16-bit format alone {fmt:.3e})")
        assert err <= 1.2 * fmt + 2e-3, (i, err, fmt)                 # measured: 0.95e-2 / 2.73e-2 / 2.85e-2, each just below the format's own
    taps = {}
    for b in (6, 7, 8):
        taps[b] = store[f"output_block_{b}_spatial_self_attn_q_time_24"].cpu().numpy()
        err, fmt = nrms(taps[b].astype(np.float32), g[f"q{b}"].astype(np.float32)), fe[f"window_q{b}"]
        print(f"inversion window: block {b} step-24 Q tap: nrms {err:.3e} (16-bit format alone {fmt:.3e})")
        assert err <= 1.2 * fmt + 2e-3, (b, err, fmt)                 # measured 3.0e-2 against 3.1e-2
    from tools_metrics import matched_iou
    iou = matched_iou(labels, g["corrected_labels"].astype(np.int64), K)
    print(f"inversion window: masks vs reference IoU {iou[0]:.4f} identical {iou[1]:.4f}")
    np.random.seed(17)                                                   # bit-exact part: the analysis of the device's own taps
    _, lab, _ = OA.match_gt_mask(OA.aggregate_blocks([taps[8], taps[7], taps[6]]), K, np.random.mtrand._rand)
    th, tw = OA.dense_tracking(taps[7], Fn, LAT // 2, LAT // 2)
    corr, _ = OA.correct_low_res_mask(lab.reshape(Fn, LAT // 2, LAT // 2), th, tw)
    assert np.array_equal(np.asarray(labels).reshape(-1), corr)
    FE.FeatureStore.clear()
    FE.MaskStore.clear()
    with pytest.raises(ValueError, match="Unknown inversion type"):      # SDP:345
        segment_window(eng, torch.from_numpy(g["latent"]).to(dev), c, uc, num_masks=K, inversion_type="ddim")


def test_smooth_latent_schedule(env):
    """is_smooth_latent (sampling.py:117-125, 199-212): at steps 23 / 24 the denoised latent is decoded, frames with
    (frame_id - {1, 2}) % 3 == 0 are replaced by the mean of their neighbours, and the result is encoded again.  Checked with a
    transparent first stage (decode = encode = identity on the latent) against a replay of the reference's loop on the
    unsmoothed run's denoised latents -- the schedule and the arithmetic, independent of any VAE."""
    from vidseg_diffusion_amd.pipeline import build_sd_engine
    dev, g, net, sd = env
    eng = build_sd_engine(net)
    Fn = 7
    lat = torch.from_numpy(synthetic.latent_clip(Fn, 16, 16, seed=4)).to(dev)
    cc, ucc = synthetic.sd_conditioning(Fn, context_dim=64, seq=7, seed=5)
    c, uc = {"crossattn": torch.from_numpy(cc).to(dev)}, {"crossattn": torch.from_numpy(ucc).to(dev)}
    noise = torch.randn(lat.shape, generator=torch.Generator().manual_seed(2)).to(dev)
    x0 = eng.sampler.add_noise(lat, cond=c, uc=uc, num_steps=25, noise_level=22, noise=noise)
    seen = []

    class Transparent:
        def decode_first_stage(self, z):
            seen.append(z.clone())
            return z.clone()

        def encode_first_stage(self, x):
            return x

    def denoiser(inp, sigma, cc_, **k):
        return eng.denoiser(eng.model, inp, sigma, cc_)

    with pytest.raises(AssertionError):
        eng.sampler(denoiser, x0.clone(), cond=c, uc=uc, t_start=22, is_smooth_latent=True, model=None)
    xs = []
    out = eng.sampler(denoiser, x0.clone(), cond=c, uc=uc, t_start=22, is_smooth_latent=True, model=Transparent(),
                      img_callback=lambda xt, i: xs.append(xt.clone()))
    assert len(seen) == 2 and len(xs) == 3                           # steps 23 and 24 only
    sig = eng.sampler.discretization(25)
    x = xs[0]                                                        # state after step 22 (no smoothing there)
    for k, (i, off) in enumerate(((23, 1), (24, 2))):
        den = seen[k].cpu().numpy().copy()
        for f in range(1, Fn - 1):
            if (f - off) % 3 == 0:
                den[f] = 0.5 * (den[f - 1] + den[f + 1])
        s0, s1 = float(sig[i]), float(sig[i + 1])
        xn = x.cpu().numpy()
        x = torch.from_numpy(xn + (xn - den) / s0 * (s1 - s0)).to(dev)   # to_d + euler_step (SAM:125-131)
        assert np.abs(x.cpu().numpy() - xs[k + 1].cpu().numpy()).max() <= 1e-5 * np.abs(xn).max() + 1e-6
    plain = eng.sampler(denoiser, x0.clone(), cond=c, uc=uc, t_start=22)
    assert not torch.equal(plain, out)


def test_svd_modulated_injected_pass_vs_reference():
    """a17 on the VideoUNet: feature pass fills the FeatureStore (spatial + temporal taps), then the reference-style modulated
    pass: lambda*mask on block 8's spatial and temporal self-attention outputs, injected temporal q/k, latent blending."""
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import build_svd_engine, save_feature_maps
    from vidseg_diffusion_amd.video_unet import VideoUNet
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "svd_modulated_narrow.npz"))
    g = {k: z[k] for k in z.files}
    net = VideoUNet(**synthetic.SVD_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=4321).items()})
    Fn = g["latent"].shape[0]
    eng = build_svd_engine(net, num_frames=Fn)
    c = {k[2:]: torch.from_numpy(v).to(dev) for k, v in g.items() if k.startswith("c_")}
    uc = {k[3:]: torch.from_numpy(v).to(dev) for k, v in g.items() if k.startswith("uc_")}
    T0 = 22
    noised = eng.sampler.add_noise(torch.from_numpy(g["latent"]).to(dev), cond=c, uc=uc, num_steps=25, noise_level=T0,
                                   noise=torch.from_numpy(g["noise"]).to(dev))
    FE.FeatureStore.clear()
    base, exp = "/nonexistent/vs_svdmod", "exp"
    extra = {"image_only_indicator": torch.zeros(2, Fn), "num_video_frames": Fn}

    def denoiser(inp, sigma, cc, is_modulate_step=False, is_injected_step=False, modulate_params=None):
        return eng.denoiser(eng.model, inp, sigma, cc, is_modulate_step=is_modulate_step, is_injected_step=is_injected_step,
                            modulate_params=modulate_params, **extra)

    feat = eng.sampler(denoiser, noised.clone(), cond=c, uc=uc, t_start=T0,
                       img_callback=lambda xt, i: save_feature_maps(eng, base, exp, i, xt=xt))
    assert nrms(feat.cpu().numpy(), g["feat_final"]) < act_mode()[1]
    for tag in [k[4:-6] for k in g if k.startswith("mod_") and k.endswith("_final")]:
        lam = float(g[f"lam_{tag}"])
        mp = {"feature_masks": [torch.from_numpy(m).to(dev) for m in g["masks"]], "modulate_block_idx": [8],
              "modulate_layer_type": ["spatial", "temporal"], "modulate_attn_type": ["self_attn"], "modulate_timestep": [T0],
              "modulate_schedule": "constant", "modulate_lambda_start": lam, "modulate_lambda_end": lam, "num_frames": Fn,
              "modulate_uc": True, "is_injected_features": True,
              "injected_feature_types": ["temporal_cross_attn_k", "temporal_cross_attn_q", "temporal_self_attn_k", "temporal_self_attn_q"],
              "injected_block_types": ["output"], "input_block_indices": [3, 4, 5, 6, 7, 8, 10, 11],
              "output_block_indices": list(range(1, 12)), "feature_folder": base, "exp_name": exp, "injected_features_group": {},
              "modulate_layer_frames": {}, "modulate_block_frames": {}, "modulate_timestep_frames": {}, "modulate_lambda_layers": {},
              "latent_mask_start": T0, "latent_mask_end": 25}
        xs = []
        final = eng.sampler(denoiser, noised.clone(), cond=c, uc=uc, img_callback=lambda xt, i: xs.append(xt.cpu().numpy()),
                            is_modulate=True, modulate_params=mp, t_start=T0, is_latent_blending=True, feature_height=8, feature_width=8,
                            model=None)
        ref = g[f"mod_{tag}_x_steps"]
        assert len(xs) == ref.shape[0]
        errs = [nrms(xs[i], ref[i]) for i in range(len(xs))]
        d_ref = g[f"mod_{tag}_final"] - g["feat_final"]
        d_got = final.cpu().numpy() - feat.cpu().numpy()
        print("svd modulated", tag, "lambda", lam, "step nrms", [round(e, 4) for e in errs], "delta nrms", round(nrms(d_got, d_ref), 4),
              "delta/ref", round(float(np.abs(d_ref).mean() / np.abs(g["feat_final"]).mean()), 4))
        assert max(errs) < act_mode()[1]
        # the modulation must actually have moved the sample the way the reference's did: the DIFFERENCE between the modulated
        # and the unmodulated final latent within 2 % of the reference's difference (measured 3.5e-3), for lambda = 50 and 2000 alike
        assert nrms(d_got, d_ref) < (2e-2 if act_mode()[0] == "f16" else 0.35), nrms(d_got, d_ref)


def test_modulation_sweep_step4(env):
    """Step 4 harness on the narrow SD engine: segment_window leaves dumps + masks, modulation_sweep runs 2*K modulated passes;
    one entry is re-run by hand through the (reference-parity-tested) sampler call and must be bit-identical, +/- lambda and
    different labels must give different latents, everything finite."""
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import build_sd_engine, load_feature_masks, make_denoiser, modulation_sweep, segment_window
    dev, g, net, sd = env
    eng = build_sd_engine(net)
    Fn = 3
    lat = torch.from_numpy(synthetic.latent_clip(Fn, 16, 16, seed=5)).to(dev)
    cc, ucc = synthetic.sd_conditioning(Fn, context_dim=64, seq=7, seed=2)
    c, uc = {"crossattn": torch.from_numpy(cc).to(dev)}, {"crossattn": torch.from_numpy(ucc).to(dev)}
    noise = torch.randn(lat.shape, generator=torch.Generator().manual_seed(3)).to(dev)
    FE.FeatureStore.clear(); FE.MaskStore.clear()
    base, exp = "/nonexistent/vs_sweep", "exp"
    labels, st = segment_window(eng, lat, c, uc, num_masks=3, t_start=22, is_aggre_attn=True, seed=17, noise=noise,
                                feature_folder=base, exp_name=exp, keep_all_steps=True)
    folder = os.path.join(base, exp, "match_gt_mask", "output_block_8_output_block_7_output_block_6_spatial_self_attn_q_masks_3")
    assert FE.MaskStore.get(folder) is not None
    uniq = np.unique(labels)
    m = load_feature_masks(folder, int(uniq[0]), num_frames=Fn, modulate_block_idx=7, base_height=2, base_width=2, device=dev)
    assert len(m) == Fn and m[0].shape == (64,) and m[0].dtype == torch.float64
    assert np.array_equal(m[1].cpu().numpy(), (labels[1] == uniq[0]).astype(np.float64))
    res = modulation_sweep(eng, lat, c, uc, uniq, folder, t_start=22, feature_folder=base, exp_name=exp, noise=noise, seed=17)
    assert set(res) == {(s, int(l)) for s in (1, -1) for l in uniq}
    for v in res.values():
        assert v.shape == lat.shape and torch.isfinite(v).all()
    a, b = res[(1, int(uniq[0]))], res[(-1, int(uniq[0]))]
    assert (a - b).abs().max() > 0
    if len(uniq) > 1:
        assert (a - res[(1, int(uniq[1]))]).abs().max() > 0
    # manual re-run of one entry
    mp = {"feature_masks": m, "modulate_block_idx": [7], "modulate_layer_type": ["spatial"], "modulate_attn_type": ["cross_attn"],
          "modulate_timestep": [22], "modulate_schedule": "constant", "modulate_lambda_start": 50.0, "modulate_lambda_end": 50.0,
          "num_frames": Fn, "modulate_uc": True, "is_injected_features": True,
          "injected_feature_types": ["spatial_cross_attn_k", "spatial_cross_attn_q", "spatial_self_attn_k", "spatial_self_attn_q"],
          "injected_block_types": ["output"], "input_block_indices": [3, 4, 5, 6, 7, 8, 9, 10, 11],
          "output_block_indices": list(range(1, 12)), "feature_folder": base, "exp_name": exp, "injected_features_group": {},
          "modulate_layer_frames": {}, "modulate_block_frames": {}, "modulate_timestep_frames": {}, "modulate_lambda_layers": {},
          "latent_mask_start": 22, "latent_mask_end": 23}
    x0 = eng.sampler.add_noise(lat, cond=c, uc=uc, num_steps=25, noise_level=22, noise=noise)
    ref = eng.sampler(make_denoiser(eng, Fn), x0.clone(), cond=c, uc=uc, is_modulate=True, modulate_params=mp, t_start=22,
                      is_latent_blending=True, feature_height=8, feature_width=8)
    assert torch.equal(ref, a)
    # Steps 4-5 in one call: sweep -> decode_first_stage -> difference maps -> arg-max, against the oracle's Step 5 on the device's
    # own decoded frames (narrow first stage with the golden's decoder weights)
    from oracle import process_output as OPO
    from tests.test_oracle_vae import narrow_decoder_state_dict
    from vidseg_diffusion_amd.pipeline import segmentation_map_window
    from vidseg_diffusion_amd.vae import decode_first_stage
    gz = np.load(os.path.join(os.path.dirname(__file__), "golden", "vae_decoder_narrow.npz"))
    vae, _, vsd = narrow_decoder_state_dict(gz["pq_bias"])
    vae.load_state_dict(vsd)
    seg, lat2 = segmentation_map_window(eng, vae, lat, c, uc, uniq, folder, t_start=22, feature_folder=base, exp_name=exp, noise=noise,
                                        seed=17)
    assert seg.shape == (Fn, 128, 128) and seg.dtype == torch.uint8 and all(torch.equal(lat2[k], res[k]) for k in res)
    pos = np.stack([decode_first_stage(vae, res[(1, int(l))], 0.18215).cpu().numpy() for l in uniq])
    neg = np.stack([decode_first_stage(vae, res[(-1, int(l))], 0.18215).cpu().numpy() for l in uniq])
    ref_seg, _ = OPO.seg_maps(pos, neg, [int(l) for l in uniq])
    assert np.array_equal(seg.cpu().numpy(), ref_seg)
    seg_f, _ = segmentation_map_window(eng, vae, lat, c, uc, uniq, folder, t_start=22, feature_folder=base, exp_name=exp, noise=noise,
                                       seed=17, filter_difference=True, label_maps=labels.reshape(Fn, 8, 8))
    assert seg_f.shape == seg.shape and set(np.unique(seg_f.cpu().numpy())) <= set(int(l) for l in uniq)


def test_c1_full_width_window_vs_oracle():
    """BASELINE configs[0] (the reference's CPU-runnable case): FULL-size SD 2.1 UNet, 4 frames at 256x256 (latent 32x32),
    K = 5, one step (t_start = 24), aggregated blocks.  HIP pipeline vs the all-fp32 CPU oracle: Q taps of blocks 6-8 within the
    bf16 tolerance, masks by IoU up to a label permutation; and the analysis of the device's own taps re-run by the oracle must
    give bit-identical masks."""
    from oracle import analysis as OA
    from oracle import pipeline as OP
    from oracle.unet import UNetOracle
    from tools_metrics import matched_iou
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import build_sd_engine, segment_window
    from vidseg_diffusion_amd.unet import UNetModel
    dev = torch.device("cuda:0")
    cfg = dict(synthetic.SD21_FULL)
    net = UNetModel(**cfg)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    # inputs as in the headline workload ("Conditioning of the headline workload"): near-init weights and K well-separated
    # regions, so the K-means optimum does not move with rounding-level differences of the taps (on a blob latent with plain
    # random weights the same taps, 1.5e-3 apart, gave IoU 0.93 after a change of summation order inside the attention kernel)
    sd = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1234, zero_gain=synthetic.HEADLINE["zero_gain"]).items()}
    net.load_state_dict(sd)
    Fn, K, T0 = 4, 5, 24
    lat = synthetic.region_clip(Fn, 32, 32, num_regions=K, seed=1, amp=synthetic.HEADLINE["amp"], noise=synthetic.HEADLINE["noise"],
                                protos=synthetic.HEADLINE["protos"])
    c, ucn = synthetic.sd_conditioning(Fn, context_dim=cfg["context_dim"], seq=77, seed=1)
    noise = torch.from_numpy(np.random.Generator(np.random.PCG64(9)).standard_normal(lat.shape).astype(np.float32))
    torch.set_grad_enabled(False)
    ref = OP.segment_window(UNetOracle(sd), torch.from_numpy(lat), torch.from_numpy(c), torch.from_numpy(ucn), noise, num_masks=K,
                            t_start=T0, seed=17)
    eng = build_sd_engine(net)
    FE.FeatureStore.clear(); FE.MaskStore.clear()
    cc, uc = {"crossattn": torch.from_numpy(c).to(dev)}, {"crossattn": torch.from_numpy(ucn).to(dev)}
    labels, _ = segment_window(eng, torch.from_numpy(lat).to(dev), cc, uc, num_masks=K, t_start=T0, seed=17, noise=noise.to(dev),
                               feature_folder="/nonexistent/c1", exp_name="w")
    store = FE.FeatureStore.folder("/nonexistent/c1", "w")
    taps = {b: store[f"output_block_{b}_spatial_self_attn_q_time_24"].cpu().numpy() for b in (6, 7, 8)}
    for b in (6, 7, 8):
        e = nrms(taps[b].astype(np.float32), ref["q_taps"][b].astype(np.float32))
        print("C1 tap", b, "nrms", round(e, 4))
        assert taps[b].shape == (2 * Fn, 256, 640) and e < act_mode()[1]
    np.random.seed(17)
    _, lab_o, _ = OA.match_gt_mask(OA.aggregate_blocks([taps[8], taps[7], taps[6]]), K, np.random.mtrand._rand)
    assert np.array_equal(labels.reshape(-1), lab_o)
    iou, exact = matched_iou(labels, ref["labels"], K)
    print("C1 full-width mask IoU vs fp32 oracle", iou, "exact", exact)
    assert iou >= (0.99 if act_mode()[0] == "f16" else 0.90)
    FE.FeatureStore.clear(); FE.MaskStore.clear()


def test_overlapped_clip_equals_sequential(env):
    """WindowPipeline (analysis of window w on a second stream while window w+1's feature pass runs) must give exactly the
    sequential loop's masks: 3 windows of 4 frames incl. the re-anchored last one, refinement on."""
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import build_sd_engine, segment_clip
    dev, g, net, sd = env
    eng = build_sd_engine(net)
    Ft = 10                                                        # windows [0,4) [4,8) [6,10)
    lat = torch.from_numpy(synthetic.latent_clip(Ft, 16, 16, seed=7)).to(dev)
    cc, ucc = synthetic.sd_conditioning(Ft, context_dim=64, seq=7, seed=3)
    cc, ucc = torch.from_numpy(cc).to(dev), torch.from_numpy(ucc).to(dev)
    noise_all = torch.randn(lat.shape, generator=torch.Generator().manual_seed(1)).to(dev)

    def run(overlap, tag):
        FE.FeatureStore.clear(); FE.MaskStore.clear()
        res = []
        # per-window noise must not depend on the execution order: pass it explicitly
        from vidseg_diffusion_amd.pipeline import WindowPipeline, WindowState, segment_window, window_slices
        slices = window_slices(Ft, 4)
        if overlap:
            pipe = WindowPipeline(eng, lanes=overlap, num_masks=4, is_refine_mask=True)
            for b, (s, e) in enumerate(slices):
                prev = pipe.push(lat[s:e].contiguous(), {"crossattn": cc[s:e]}, {"crossattn": ucc[s:e]}, t_start=22, seed=17,
                                 feature_folder="/nonexistent/ov" + tag, exp_name=f"w{b}", noise=noise_all[s:e].contiguous())
                if prev is not None:
                    res.append(prev)
            res += pipe.drain()
        else:
            st = WindowState()
            for b, (s, e) in enumerate(slices):
                lab, st = segment_window(eng, lat[s:e].contiguous(), {"crossattn": cc[s:e]}, {"crossattn": ucc[s:e]}, num_masks=4,
                                         is_refine_mask=True, t_start=22, seed=17, state=st, feature_folder="/nonexistent/ov" + tag,
                                         exp_name=f"w{b}", noise=noise_all[s:e].contiguous())
                res.append(lab)
        return res

    a, b, c2 = run(0, "s"), run(1, "p"), run(2, "q")              # sequential, one lane, two feature-pass lanes in flight
    assert len(a) == len(b) == len(c2) == 3
    for x, y in zip(a, c2):
        assert np.array_equal(x, y), "two-lane pipeline differs from the sequential loop"
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    out = segment_clip(eng, lat, lambda s, e: ({"crossattn": cc[s:e]}, {"crossattn": ucc[s:e]}), batch_size=4, num_masks=4, t_start=22,
                       seed=17, feature_folder="/nonexistent/ovc")
    assert [(s, e) for s, e, _ in out] == [(0, 4), (4, 8), (6, 10)]
    FE.FeatureStore.clear(); FE.MaskStore.clear()


def test_fp8_attention_path(env):
    """BASELINE configs[4]: every spatial attention of the UNet on the e4m3 kernel.  There is no reference for an fp8 path; the bar
    is closeness to the 16-bit path of the same build: UNet output and decoder Q taps within 5e-2 normalised rms (e4m3 keeps
    3 mantissa bits: 2^-4 relative per operand, averaged over 64-wide dot products and the softmax)."""
    from vidseg_diffusion_amd import ops
    dev, g, net, sd = env
    x, t, ctx = (torch.from_numpy(g[k]).to(dev) for k in ("fw_x", "fw_t", "fw_ctx"))
    base = net(x, timesteps=t, context=ctx).cpu().numpy()
    taps = {b: net.output_blocks[b][1].transformer_blocks[0].attn1.q.float().cpu().numpy() for b in (6, 7, 8)}
    prev = ops.set_attention_fp8(True, min_keys=0)
    try:
        out = net(x, timesteps=t, context=ctx).cpu().numpy()
        taps8 = {b: net.output_blocks[b][1].transformer_blocks[0].attn1.q.float().cpu().numpy() for b in (6, 7, 8)}
    finally:
        ops.set_attention_fp8(prev)
    assert not np.array_equal(out, base), "the fp8 switch did not change the attention kernel"
    assert nrms(out, base) < 5e-2, nrms(out, base)
    for b in taps:
        assert nrms(taps8[b], taps[b]) < 5e-2, (b, nrms(taps8[b], taps[b]))


def test_fp8_attention_path_video_unet_and_k50():
    """BASELINE configs[4] on the VideoUNet (SVD): the spatial self / cross attentions of every SpatialVideoTransformer on the
    e4m3 kernel (temporal attention over T frames stays 16-bit: its sequences are 14 keys long).  No reference exists for an
    fp8 path; bars: UNet output and the spatial / temporal Q taps of block 8 within 5e-2 normalised rms of the 16-bit path,
    and one window at K = 50 masks runs end to end with the refinement (the configs[4] mask count)."""
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd import ops
    from vidseg_diffusion_amd.pipeline import build_svd_engine, segment_window
    from vidseg_diffusion_amd.video_unet import VideoUNet
    from tools_metrics import matched_iou
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_svd_narrow.npz"))
    g = {k: z[k] for k in z.files}
    net = VideoUNet(**synthetic.SVD_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=4321).items()})
    T = int(g["T"])
    x, t, ctx, y = (torch.from_numpy(g[k]).to(dev) for k in ("fw_x", "fw_t", "fw_ctx", "fw_y"))

    def fwd():
        out = net(x, timesteps=t, context=ctx, y=y, num_video_frames=T, image_only_indicator=torch.zeros(2, T)).cpu().numpy()
        b8 = net.output_blocks[8][1]
        return out, b8.transformer_blocks[0].attn1.q.float().cpu().numpy(), b8.time_stack[0].attn1.q.float().cpu().numpy()

    base = fwd()
    prev = ops.set_attention_fp8(True, min_keys=0)
    try:
        got = fwd()
        # one window of the SVD engine at K = 50 on the fp8 path vs the 16-bit path (narrow width, 5 frames of 16x16 latents)
        Fn, K = 5, 50
        eng = build_svd_engine(net, num_frames=Fn)
        gen = torch.Generator().manual_seed(3)
        lat = torch.from_numpy(synthetic.latent_clip(Fn, 16, 16, seed=11)).to(dev)
        c = {"crossattn": torch.randn((1, 1, 64), generator=gen).repeat(Fn, 1, 1).to(dev), "concat": (lat[:1].repeat(Fn, 1, 1, 1) * 0.5),
             "vector": torch.randn((1, 64), generator=gen).repeat(Fn, 1).to(dev)}
        uc = {"crossattn": torch.zeros_like(c["crossattn"]), "concat": torch.zeros_like(c["concat"]), "vector": c["vector"].clone()}
        noise = torch.randn(lat.shape, generator=gen).to(dev)
        res = {}
        for tag, on in (("fp8", True), ("a16", False)):
            ops.set_attention_fp8(on, min_keys=0)
            FE.FeatureStore.clear(); FE.MaskStore.clear()
            res[tag], _ = segment_window(eng, lat, c, uc, num_masks=K, t_start=22, is_refine_mask=True, seed=17, noise=noise,
                                         feature_folder="/nonexistent/fp8k50", exp_name=tag, keep_all_steps=False)
    finally:
        ops.set_attention_fp8(prev)
        FE.FeatureStore.clear(); FE.MaskStore.clear()
    assert not np.array_equal(got[0], base[0]), "the fp8 switch did not change the attention kernel"
    for name, a, b in zip(("output", "spatial q8", "temporal q8"), got, base):
        e = nrms(a, b)
        print(f"video unet fp8 vs 16-bit {name}: nrms {e:.3e}")
        assert e < 5e-2, (name, e)
    for tag in res:
        assert res[tag].shape == (Fn, 64) and len(np.unique(res[tag])) <= K
    iou, same = matched_iou(res["fp8"], res["a16"], K)
    print(f"K=50 narrow SVD window, fp8 vs 16-bit masks: IoU {iou:.3f}, identical {same:.3f}")


def test_masks_only_feature_pass(env):
    """Opt-in pruning (pipeline.feature_pass(masks_only=True)): the last step on the conditional half only, stopped after output
    block 8.  Every UNet operator is per sample, so the conditional-half taps equal the full pass's up to fp32 summation order
    (the half batch changes split-K choices): another fp16 rounding of the same arithmetic, <= 2e-3 normalised rms here."""
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import build_sd_engine, segment_window
    dev, g, net, sd = env
    eng = build_sd_engine(net)
    Fn, K = 4, 5
    lat = torch.from_numpy(synthetic.latent_clip(Fn, 16, 16, seed=3)).to(dev)
    cc = {"crossattn": torch.from_numpy(np.random.Generator(np.random.PCG64(4)).standard_normal((Fn, 7, 64)).astype(np.float32)).to(dev)}
    uc = {"crossattn": torch.zeros_like(cc["crossattn"])}
    noise = torch.from_numpy(np.random.Generator(np.random.PCG64(5)).standard_normal(tuple(lat.shape)).astype(np.float32)).to(dev)
    out = {}
    for tag, kw in (("full", dict(keep_all_steps=False)), ("pruned", dict(masks_only=True))):
        FE.FeatureStore.clear(); FE.MaskStore.clear()
        labels, _ = segment_window(eng, lat, cc, uc, num_masks=K, is_refine_mask=True, seed=17, noise=noise, feature_folder="/nonexistent/mo",
                                   exp_name=tag, **kw)
        store = FE.FeatureStore.folder("/nonexistent/mo", tag)
        out[tag] = (labels, {b: store[f"output_block_{b}_spatial_self_attn_q_time_24"][Fn:].float().cpu().numpy() for b in (6, 7, 8)})
    from tools_metrics import matched_iou
    for b in (6, 7, 8):
        assert nrms(out["pruned"][1][b], out["full"][1][b]) < 2e-3, b
    iou, exact = matched_iou(out["pruned"][0], out["full"][0], K)
    assert iou >= 0.97 and exact >= 0.98, (iou, exact)
    with pytest.raises(ValueError):
        segment_window(eng, lat, cc, uc, num_masks=K, seed=17, noise=noise, feature_timestep="23", masks_only=True)


def test_masks_only_feature_pass_svd():
    """The same opt-in pruning on the video UNet (cond-only batch = one video, image_only_indicator [1, F]): the spatial taps of
    decoder blocks 6-8 and the temporal taps of block 8 at the last step vs the full schedule, and the same masks."""
    from tools_metrics import matched_iou
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import build_svd_engine, segment_window
    from vidseg_diffusion_amd.video_unet import VideoUNet
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "svd_sampler_narrow.npz"))
    g = {k: z[k] for k in z.files}
    net = VideoUNet(**synthetic.SVD_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=4321).items()})
    Fn = g["sm_latent"].shape[0]
    eng = build_svd_engine(net, num_frames=Fn)
    c = {k[2:]: torch.from_numpy(v).to(dev) for k, v in g.items() if k.startswith("c_")}
    uc = {k[3:]: torch.from_numpy(v).to(dev) for k, v in g.items() if k.startswith("uc_")}
    lat, noise = torch.from_numpy(g["sm_latent"]).to(dev), torch.from_numpy(g["sm_noise"]).to(dev)
    out = {}
    for tag, kw in (("full", dict(keep_all_steps=False)), ("pruned", dict(masks_only=True))):
        FE.FeatureStore.clear(); FE.MaskStore.clear()
        labels, _ = segment_window(eng, lat, c, uc, num_masks=4, t_start=17, is_refine_mask=True, seed=17, noise=noise,
                                   feature_folder="/nonexistent/mosvd", exp_name=tag, **kw)
        st = FE.FeatureStore.folder("/nonexistent/mosvd", tag)
        taps = {b: st[f"output_block_{b}_spatial_self_attn_q_time_24"][Fn:].float().cpu().numpy() for b in (6, 7, 8)}
        tq = st["output_block_8_temporal_self_attn_q_time_24"].float().cpu().numpy()
        taps["t8"] = tq[tq.shape[0] // 2:]
        out[tag] = (labels, taps)
    for k in out["full"][1]:
        assert nrms(out["pruned"][1][k], out["full"][1][k]) < 2e-3, k
    iou, exact = matched_iou(out["pruned"][0], out["full"][0], 4)
    assert iou >= 0.97 and exact >= 0.98, (iou, exact)


def test_engine_from_config_equals_hand_wired_engine(env):
    """`engine.DiffusionEngine` instantiated from a reference-schema config (every target a reference dotted path), weights loaded
    through checkpoint-style keys, drives Steps 1-3b exactly like the hand-wired `build_sd_engine`."""
    from tests.test_conditioner import _narrow_model_config
    from vidseg_diffusion_amd import feature_extraction as FE, util
    from vidseg_diffusion_amd.pipeline import build_sd_engine, segment_window
    dev, g, net, sd = env
    eng2 = util.instantiate_from_config(_narrow_model_config()["model"])
    missing, unexpected = eng2.load_state_dict({"model.diffusion_model." + k: v for k, v in sd.items()})
    assert not unexpected and not missing
    Fn, K = 4, 5
    lat = torch.from_numpy(synthetic.latent_clip(Fn, 16, 16, seed=3)).to(dev)
    cc = {"crossattn": torch.from_numpy(np.random.Generator(np.random.PCG64(4)).standard_normal((Fn, 7, 64)).astype(np.float32)).to(dev)}
    uc = {"crossattn": torch.zeros_like(cc["crossattn"])}
    noise = torch.from_numpy(np.random.Generator(np.random.PCG64(5)).standard_normal(tuple(lat.shape)).astype(np.float32)).to(dev)
    out = []
    for tag, e in (("hand", build_sd_engine(net)), ("config", eng2)):
        FE.FeatureStore.clear(); FE.MaskStore.clear()
        labels, _ = segment_window(e, lat, cc, uc, num_masks=K, is_refine_mask=True, seed=17, noise=noise, feature_folder="/nonexistent/ec",
                                   exp_name=tag, keep_all_steps=False)
        out.append(labels)
    assert np.array_equal(out[0], out[1])


def test_svd_steps_4_5_on_the_video_first_stage():
    """Steps 1-5 on the video engine: masks, the temporal modulation sweep, decodes through AutoencodingEngine + VideoDecoder
    (whole videos per call, `timesteps` = frames), difference maps and arg-max -- vs the oracle's Step 5 on the device's decodes."""
    from oracle import process_output as OPO
    from tests.test_oracle_vae import narrow_video_decoder_state_dict
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import build_svd_engine, segment_window, segmentation_map_window
    from vidseg_diffusion_amd.vae import decode_first_stage
    from vidseg_diffusion_amd.video_unet import VideoUNet
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "svd_sampler_narrow.npz"))
    g = {k: z[k] for k in z.files}
    net = VideoUNet(**synthetic.SVD_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=4321).items()})
    Fn = g["sm_latent"].shape[0]
    eng = build_svd_engine(net, num_frames=Fn)
    c = {k[2:]: torch.from_numpy(v).to(dev) for k, v in g.items() if k.startswith("c_")}
    uc = {k[3:]: torch.from_numpy(v).to(dev) for k, v in g.items() if k.startswith("uc_")}
    lat, noise = torch.from_numpy(g["sm_latent"]).to(dev), torch.from_numpy(g["sm_noise"]).to(dev)
    gv = np.load(os.path.join(os.path.dirname(__file__), "golden", "vae_video_decoder_narrow.npz"))
    vae, _, vsd = narrow_video_decoder_state_dict(gv)
    vae.load_state_dict(vsd)
    FE.FeatureStore.clear(); FE.MaskStore.clear()
    base, exp, K = "/nonexistent/svd45", "w", 3
    labels, _ = segment_window(eng, lat, c, uc, num_masks=K, t_start=22, seed=17, noise=noise, feature_folder=base, exp_name=exp,
                               keep_all_steps=True)
    uniq = np.unique(labels)
    folder = os.path.join(base, exp, "match_gt_mask", f"output_block_8_output_block_7_output_block_6_spatial_self_attn_q_masks_{K}")
    assert FE.MaskStore.get(folder) is not None
    seg, lats = segmentation_map_window(eng, vae, lat, c, uc, uniq, folder, t_start=22, feature_folder=base, exp_name=exp, noise=noise,
                                        seed=17, modulate_layer_type=("temporal",), modulate_attn_type=("self_attn",))
    h, w = lat.shape[-2:]
    assert seg.shape == (Fn, 8 * h, 8 * w) and set(np.unique(seg.cpu().numpy())) <= set(int(l) for l in uniq)
    pos = np.stack([decode_first_stage(vae, lats[(1, int(l))], 0.18215).cpu().numpy() for l in uniq])
    neg = np.stack([decode_first_stage(vae, lats[(-1, int(l))], 0.18215).cpu().numpy() for l in uniq])
    ref_seg, _ = OPO.seg_maps(pos, neg, [int(l) for l in uniq])
    assert np.array_equal(seg.cpu().numpy(), ref_seg)
