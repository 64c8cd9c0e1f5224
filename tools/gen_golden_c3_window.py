"""Generate tests/golden/c3_window.npz: ONE full-size window of BASELINE configs[2] through the REFERENCE.

SVD img2vid topology at full width (VideoUNet, model_channels 320, 1524.6 M parameters), T = 14 frames at 576x1024 (latent
14x4x72x128; decoder blocks 6-8 at 36x64 = 2304 tokens), CFG batch 28 = two videos (unconditional, conditional),
EulerEDMSampler + Denoiser(VScalingWithEDMcNoise) + LinearPredictionGuider(1.0 -> 2.5) + OpenAIWrapper (channel concat of the
frame-0 latent), `add_noise` to step 24 and ONE Euler step (t_start = 24: one CFG evaluation, 89.6 TFLOP in fp32 -- the 8-step
schedule of the bench would take an hour here and tests nothing more about the kernels), the spatial AND temporal Q/K taps of
decoder blocks 6/7/8, then the reference's own `feature_extraction_main("match_gt_mask")` with the 3-block aggregate, K = 20
(all ten K-means restarts observed) and `correct_low_res_mask` on block 7 -- svd_pipeline_vspw.py:324-395 on exactly the
inputs `bench.py --config svd` builds for window 0 (make_inputs(svd=True): synthetic.latent_clip(14, 72, 128, seed=1), generator
seed 100 for the noise / CLIP token / vector, weights fill_state_dict(seed=1234)).

Everything is imported read-only from /root/reference (fp32, CPU); build container only (~10 min, ~30 GB).  The two videos of
the CFG batch are evaluated one after the other: no operator of the VideoUNet crosses the video axis b (GroupNorm statistics are
per (b t) sample for the spatial layers and per b for the 3-D ResBlocks, temporal attention is per (b, location); video_model.py:
451-566), so the values are those of the batch-2 call; the split only bounds memory.  xformers is absent, so the transformer
blocks are built with spatial_transformer_attn_type="softmax" (SURVEY a11).

The fixture stores subsamples (conditional video, every 16th location, every 4th channel, fp16) of the six taps + full-tensor
norms, the Euler-step result, the reference's labels / corrected labels / ten restarts, and sha256 of every input.

    python tools/gen_golden_c3_window.py                                   # the one-step fixture c3_window.npz (t_start = 24, window 0)
    python tools/gen_golden_c3_window.py --t-start 17 --windows 0 1 2      # the REFERENCE's schedule (svd_pipeline_vspw.py:236-237, 598:
                                                                           # modulate_timestep 17 -> 8 Euler steps, 8 CFG evaluations,
                                                                           # ~75 min and ~30 GB per window here) -> c3_t17_w{0,1,2}.npz
The t_start = 17 fixtures hold coarser subsamples (every 32nd location, every 8th channel; x_final every 2nd latent row / column + its
norm and the per-step norms of x) so that three windows stay ~2 MB each.
"""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_golden_c2_window import RestartRecorder  # noqa: E402
from ref_import import REF, import_reference  # noqa: E402
from vidseg_diffusion_amd import synthetic  # noqa: E402

F, LH, LW, K, T_START, NUM_STEPS = 14, 72, 128, 20, 24, 25
BLOCKS = ["output_block_8", "output_block_7", "output_block_6"]


def svd_inputs(window_id=0):
    """bench.py make_inputs(svd=True) on the CPU (same generator order: noise, CLIP token, vector)."""
    cfg = synthetic.SVD_FULL
    g = torch.Generator().manual_seed(100 + window_id)
    lat = torch.from_numpy(synthetic.latent_clip(F, LH, LW, seed=1 + window_id))
    noise = torch.randn(lat.shape, generator=g)
    ctx = torch.randn((1, 1, cfg["context_dim"]), generator=g).repeat(F, 1, 1)
    cat = lat[:1].repeat(F, 1, 1, 1) / 0.18215 * 0.2
    vec = torch.randn((1, cfg["adm_in_channels"]), generator=g).repeat(F, 1)
    c = {"crossattn": ctx, "concat": cat, "vector": vec}
    uc = {"crossattn": torch.zeros_like(ctx), "concat": torch.zeros_like(cat), "vector": vec.clone()}
    return lat, c, uc, noise


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--t-start", type=int, default=24)
    ap.add_argument("--windows", type=int, nargs="*", default=[0])
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    global T_START
    T_START = args.t_start
    fe = import_reference()
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    torch.set_grad_enabled(False)
    t_all = time.time()
    net = VideoUNet(use_checkpoint=False, spatial_transformer_attn_type="softmax", **synthetic.SVD_FULL).eval().to("cpu")
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synthetic.fill_state_dict(shapes, seed=1234)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    del sd
    for wid in args.windows:
        one_window(fe, net, shapes, wid, t_all)


def one_window(fe, net, shapes, wid, t_all):
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    full = T_START == 24 and wid == 0                                  # the original one-step fixture keeps its finer subsamples
    LS, CS = (16, 4) if full else (32, 8)
    rec = dict(state_dict_signature=synthetic.state_dict_signature(shapes), weight_seed=1234, F=F, lat_h=LH, lat_w=LW, K=K,
               t_start=T_START, num_steps=NUM_STEPS, seed=17, window_id=wid, loc_stride=LS, ch_stride=CS)
    lat, c, uc, noise = svd_inputs(wid)
    rec.update(latent_sha256=synthetic.sha256_of(lat.numpy()), noise_sha256=synthetic.sha256_of(noise.numpy()),
               ctx_sha256=synthetic.sha256_of(c["crossattn"].numpy()), vector_sha256=synthetic.sha256_of(c["vector"].numpy()))

    mods = {}
    for b in (6, 7, 8):
        st = net.output_blocks[b][1]
        assert "SpatialVideoTransformer" in str(type(st))
        mods[f"s{b}"] = st.transformer_blocks[0].attn1               # spatial self-attention: q/k [(b t), s, c]
        mods[f"t{b}"] = st.time_stack[0].attn1                       # temporal self-attention: q/k [(b s), t, c]
    net.__dict__.pop("forward", None)                                  # a previous window's wrapper is dropped first
    orig_forward = net.forward
    got = {}

    def split_forward(x, timesteps=None, context=None, y=None, num_video_frames=None, image_only_indicator=None, **kw):
        outs, parts = [], {k: ([], []) for k in mods}
        for v in range(x.shape[0] // F):                              # one video (14 frames) at a time
            sl = slice(v * F, (v + 1) * F)
            outs.append(orig_forward(x[sl], timesteps=timesteps[sl], context=context[sl], y=y[sl], num_video_frames=num_video_frames,
                                     image_only_indicator=image_only_indicator[v:v + 1], **kw))
            for k, a in mods.items():
                parts[k][0].append(a.q)
                parts[k][1].append(a.k)
        for k, a in mods.items():
            a.q, a.k = torch.cat(parts[k][0], 0), torch.cat(parts[k][1], 0)
        print(f"  network evaluation done, {time.time() - t_all:.0f} s", flush=True)
        return torch.cat(outs, 0)

    net.forward = split_forward
    dd = "sgm.modules.diffusionmodules."
    den_m = Denoiser(scaling_config={"target": dd + "denoiser_scaling.VScalingWithEDMcNoise"})
    sampler = EulerEDMSampler(discretization_config={"target": dd + "discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
                              guider_config={"target": dd + "guiders.LinearPredictionGuider",
                                             "params": {"max_scale": 2.5, "min_scale": 1.0, "num_frames": F}},
                              num_steps=NUM_STEPS, s_churn=0, s_tmin=0, s_tmax=999, s_noise=1, device="cpu")
    model = OpenAIWrapper(net)
    extra = {"image_only_indicator": torch.zeros(2, F), "num_video_frames": F}

    def denoiser(inp, sigma, cc, is_modulate_step=False, is_injected_step=False, modulate_params=None):
        return den_m(model, inp, sigma, cc, is_modulate_step=is_modulate_step, is_injected_step=is_injected_step,
                     modulate_params=modulate_params, **extra)

    torch.manual_seed(100 + wid)                              # add_noise draws torch.randn_like(x): the generator-seeded draw above
    noised = sampler.add_noise(lat.clone(), cond=c, uc=uc, num_steps=NUM_STEPS, noise_level=T_START)
    sig = sampler.discretization(NUM_STEPS, device="cpu")
    chk = (lat + noise * sig[T_START]) / torch.sqrt(1.0 + sig[0] ** 2.0)
    assert torch.equal(noised, chk), "torch.randn_like under manual_seed != Generator draw"

    step_norms = []

    def cb(xt, i):
        step_norms.append(float(torch.linalg.norm(xt.double())))
        if i == NUM_STEPS - 1:
            for k, a in mods.items():
                got[k + "q"], got[k + "k"] = a.q.half(), a.k.half()

    final = sampler(denoiser, noised.clone(), cond=c, uc=uc, img_callback=cb, t_start=T_START)
    assert len(step_norms) == NUM_STEPS - T_START
    rec.update(x_noised_sub=noised.numpy()[:, :, ::4, ::4].astype(np.float32), x_step_norms=np.array(step_norms, dtype=np.float64))
    if full:
        rec["x_final"] = final.numpy().astype(np.float32)
    else:
        rec["x_final_sub"] = final.numpy()[:, :, ::2, ::2].astype(np.float32)
        rec["x_final_norm"] = np.float64(torch.linalg.norm(final.double()))
    S = (LH // 2) * (LW // 2)
    for b in (6, 7, 8):
        q = got[f"s{b}q"].numpy()                             # [2F, S, 640] fp16, unconditional video first
        assert q.shape == (2 * F, S, 640)
        rec[f"sq{b}_sub"] = q[F:, ::LS, ::CS]
        rec[f"sq{b}_norm"] = np.float64(np.linalg.norm(q[F:].astype(np.float64)))
        for w in ("q", "k"):
            t = got[f"t{b}{w}"].numpy()                       # [(b s), t, 640]: b-major, the conditional video is the second S rows
            assert t.shape == (2 * S, F, 640)
            rec[f"t{w}{b}_sub"] = t[S::LS, :, ::CS]
            rec[f"t{w}{b}_norm"] = np.float64(np.linalg.norm(t[S:].astype(np.float64)))

    base = tempfile.mkdtemp(prefix="vidseg_c3_")
    exp = "exp"
    fm = os.path.join(base, exp, "feature_maps")
    os.makedirs(fm)
    for name, b in zip(BLOCKS, (8, 7, 6)):
        torch.save(got[f"s{b}q"], os.path.join(fm, f"{name}_spatial_self_attn_q_time_24.pt"))
    names = [f"{i:05d}" for i in range(F)]
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        np.random.seed(17)
        with RestartRecorder() as recd:
            ul, ref_mask, ref_fm = fe.feature_extraction_main(
                "match_gt_mask", K, T_START, ",".join(BLOCKS), exp, exp, "spatial_self_attn_q", LH // 2, LW // 2, "24", frame_name_list=names,
                base_folder=base, num_frames=F, ref_mask=None, ref_feature_map=None, ref_unique_labels=None, gt_mask_path=None)
        rec["restart_labels"] = np.stack([r[0] for r in recd.runs])
        rec["restart_inertia"] = np.array([r[1] for r in recd.runs], dtype=np.float64)
        rec["restart_n_iter"] = np.array([r[2] for r in recd.runs], dtype=np.int32)
        rec["restart_best"] = np.int32(np.argmin(rec["restart_inertia"]))
        rec["unique_labels"] = np.asarray(ul)
        rec["match_labels"] = np.asarray(ref_mask).astype(np.int16)
        rec["ref_feature_sha256"] = synthetic.sha256_of(np.asarray(ref_fm))
        folder = os.path.join(base, exp, "match_gt_mask", "_".join(BLOCKS) + f"_spatial_self_attn_q_masks_{K}")
        _, ref_mask2, _ = fe.feature_extraction_main(
            "correct_low_res_mask", K, T_START, "output_block_7", exp, exp, "spatial_self_attn_q", LH // 2, LW // 2, "24",
            frame_name_list=names, base_folder=base, num_frames=F, ref_mask=ref_mask, ref_feature_map=ref_fm, ref_unique_labels=ul,
            gt_mask_path=None, mask_folder=folder)
        rec["corrected_labels"] = np.asarray(ref_mask2).astype(np.int16)
    finally:
        os.chdir(cwd)
        shutil.rmtree(base, ignore_errors=True)
    import sklearn
    rec["versions"] = np.array([f"torch {torch.__version__}", f"sklearn {sklearn.__version__}", f"numpy {np.__version__}"])
    if os.environ.get("C3_TAP_CACHE"):
        os.makedirs(os.environ["C3_TAP_CACHE"], exist_ok=True)
        np.savez(os.path.join(os.environ["C3_TAP_CACHE"], f"c3_taps_t{T_START}_w{wid}.npz"), **{k: v.numpy() for k, v in got.items()})
    out_path = os.path.join(ROOT, "tests", "golden", "c3_window.npz" if full else f"c3_t{T_START}_w{wid}.npz")
    np.savez_compressed(out_path, **rec)
    print("wrote", out_path, os.path.getsize(out_path) // 1024, "KiB in", f"{time.time() - t_all:.0f} s; labels", len(np.unique(rec["match_labels"])))


if __name__ == "__main__":
    main()
