#!/usr/bin/env python
"""Headline benchmark: segmented frames/s of the VidSeg hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one 14-frame 512x512 clip window per GPU (BASELINE configs[1]: SD 2.1, 20 masks,
is_aggre_attn on): add-noise -> 3 Euler steps (i = 22,23,24) of the full-size SD 2.1 UNet with classifier-free guidance
(batch 2x14) and the reference's Q/K taps on every decoder transformer block -> 3-block aggregation -> K-means (n_init 10)
-> 4-NN label propagation  => cluster-id masks [14, 32*32].

`value` is the PARITY mode (--precision parity, the default): the mode whose masks are the reference's (IoU >= 0.99 on >= 14 of the
16 fixture windows, `mask_iou_vs_reference`) -- fp32-accurate UNet on split fp16 operands (exact.py) with the dead work of the last
step pruned (pipeline.feature_pass(masks_only=True)).  The default run also reports `full_schedule` (same precision, nothing
pruned) and `fast_mode` (16-bit activations: 2.5-3x faster, masks at mean IoU 0.90), each with its own mask score.  Inputs are the synthetic headline workload of
vidseg_diffusion_amd/synthetic.py (`HEADLINE`: a clip of 20 drifting regions, near-initialisation weights), seeded and
already resident in HBM when the timed region starts (VAE + conditioner excluded); the same window was run
through the REFERENCE in fp32 (tools/gen_golden_c2_window.py -> tests/golden/c2_window.npz), and `mask_iou_vs_reference`
compares the masks of the timed run with the reference's.  N > 1: one window per GPU (weak scaling), see
vidseg_diffusion_amd/parallel.py for the exchange.

Prints ONE compact JSON line (rank 0, < 6 KB, LAST on stdout: `compact_line`) and writes every object of the run to bench_full.json
beside this file (and to gpurun_out/ when present).  Objects of the full record (the line carries their scalars):
  roofline      the single kernel with the most time in the timed region (normally k_gemm_ph<NJ>, the phased 256x320 /
                256x256 LDS-DMA implicit-GEMM tile): achieved = algorithmic FLOPs (2*M*N*K per launch) / HIP-event time of its
                launches, recorded on the launch stream inside the timed region; peak = 2500 TFLOP/s dense 16-bit MFMA
                (MI355X_MICROARCH.md); traffic = PMC bytes per launch of that kernel, measured IN THIS RUN by two rocprofv3
                --pmc passes over one window of the same workload (FETCH_SIZE, WRITE_SIZE; gfx950 correction of the guide),
                next to `algorithmic_bytes` per launch.  `family` = the same figures over ALL conv/linear launches.
  cpu_baseline  the oracle ("port") timed on this box's host cores on a bounded sample (see `sample`), CPU model stated.
  mask_iou_vs_reference  the metric's second half on THIS config: masks of the timed run vs the reference's fp32 masks.
  chained_window  throughput when the windows are chained like one long clip (windows > 0: 14336^2 4-NN instead of K-means).
  single_lane   throughput with ONE feature pass in flight (the headline runs two, --lanes 2); the pass the roofline's per-launch HIP
                events are recorded in, right after the timed region.
  roofline_post_unet  the post-UNet kernels alone at the headline's sizes: bytes / time against 8 TB/s (k_mean_normalize, one Lloyd
                E-step, 4-NN, dense tracking), the float64 contractions also against the f64 MFMA peak.
  step45        Steps 4-5 of one SD window in the parity precision (2K modulated passes + decodes + difference maps + arg-max), with
                and without the shared first-evaluation prefix; the SVD counterpart is `secondary.step4_latent_blending`.
  full_schedule / fast_mode   the other precision / schedule modes on the same windows (see above), with their mask scores.
  secondary     N = 1: BASELINE configs[2] (SVD 14x576x1024, t_start 17, refinement) measured after the headline, fewer steps, same
                precision mode (+ its own fast_mode).  N > 1: BASELINE configs[3] (SVD, 14 frames per GPU over the N ranks, RCCL
                all-gather for the cross-window correspondence), 1 warm-up + 2 timed steps, same timing rules as `value`.
"""
import argparse
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("VIDSEG_KEEP_LAST", "1")     # analysis.LAST_KMEANS: the ten restarts of every timed window (references only, read after the timing)

F_WIN, LAT, NUM_STEPS = 14, 64, 25
GOLDEN_C2 = os.path.join(ROOT, "tests", "golden", "c2_window.npz")


def make_inputs(dev, window_id, cfg, svd=False, lat_hw=(LAT, LAT)):
    from vidseg_diffusion_amd import synthetic
    g = torch.Generator().manual_seed(100 + window_id)
    if not svd:
        lat = torch.from_numpy(synthetic.headline_latent(F_WIN, lat_hw[0], lat_hw[1], window_id=window_id)).to(dev)
        noise = torch.randn(lat.shape, generator=g).to(dev)
        c, uc = synthetic.sd_conditioning(F_WIN, context_dim=cfg["context_dim"], seq=77, seed=1)
        return lat, {"crossattn": torch.from_numpy(c).to(dev)}, {"crossattn": torch.from_numpy(uc).to(dev)}, noise
    lat = torch.from_numpy(synthetic.latent_clip(F_WIN, lat_hw[0], lat_hw[1], seed=1 + window_id)).to(dev)
    noise = torch.randn(lat.shape, generator=g).to(dev)
    # SVD conditioning (svd_pipeline_vspw.py:300-311): CLIP-image token and frame-0 latent repeated over frames, fps/motion vector
    ctx = torch.randn((1, 1, cfg["context_dim"]), generator=g).repeat(F_WIN, 1, 1).to(dev)
    cat = lat[:1].repeat(F_WIN, 1, 1, 1) / 0.18215 * 0.2
    vec = torch.randn((1, cfg["adm_in_channels"]), generator=g).repeat(F_WIN, 1).to(dev)
    c = {"crossattn": ctx, "concat": cat, "vector": vec}
    uc = {"crossattn": torch.zeros_like(ctx), "concat": torch.zeros_like(cat), "vector": vec.clone()}
    return lat, c, uc, noise


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sd_cpu, cfg, k_masks):
    """Oracle on the host (rank 0, N = 1): one full-size UNet evaluation of FOUR frames (CFG pairs, batch 8) in fp32 and once
    more under bf16 autocast + the full-size analysis stage on synthetic dumps; scaled to a 14-frame window:
    t = 3 steps * (14 / 4) * t_unet_4frames + t_analysis."""
    from oracle import analysis as OA
    from oracle.unet import UNetOracle
    from vidseg_diffusion_amd import synthetic
    torch.set_grad_enabled(False)
    cores = torch.get_num_threads()
    o = UNetOracle(sd_cpu)
    nf = 4
    x = torch.from_numpy(synthetic.headline_latent(nf, LAT, LAT)).repeat(2, 1, 1, 1)
    c, uc = synthetic.sd_conditioning(nf, context_dim=cfg["context_dim"])
    ctx = torch.cat([torch.from_numpy(uc), torch.from_numpy(c)])
    t = torch.full((2 * nf,), 958.0)
    t0 = time.time()
    o.forward(x, t, ctx)
    t_unet = time.time() - t0
    t_bf16 = None
    try:
        t0 = time.time()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            o.forward(x, t, ctx)
        t_bf16 = time.time() - t0
    except Exception:                                                    # autocast coverage differs between torch builds
        t_bf16 = None
    blocks, _ = synthetic.attention_q_dumps(F_WIN, LAT // 2, LAT // 2, 640, num_blocks=3, seed=1)
    t0 = time.time()
    np.random.seed(17)
    OA.match_gt_mask(OA.aggregate_blocks(blocks), k_masks, np.random.mtrand._rand)
    t_an = time.time() - t0
    per = 3 * (F_WIN / nf)
    t_window = per * t_unet + t_an
    out = {"value": round(F_WIN / t_window, 5), "unit": "frames/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
           "sample": f"oracle fp32 on host: 1 full-size UNet eval of {nf} frames (CFG batch {2 * nf}, {t_unet:.1f}s) + full 14x32x32x640 "
                     f"K={k_masks} analysis ({t_an:.1f}s); window time = 3*(14/{nf})*t_unet + t_analysis = {t_window:.0f}s"}
    if t_bf16 is not None:
        out["value_bf16_autocast"] = round(F_WIN / (per * t_bf16 + t_an), 5)
        out["sample"] += f"; bf16 autocast: same UNet eval {t_bf16:.1f}s"
    return out


def timed_masks_vs_reference(timed, refine, k_masks, world=1, last_step_index=None, win_ids=None, svd=False):
    """The metric's second half on the work that was TIMED: every step's masks against the labels the REFERENCE produced for that
    window in fp32 (tests/golden/c2_window*.npz, tools/gen_golden_c2_window.py).  timed: [(window id, labels [F, N])] in step order.
    A window that was run more than once must have given identical masks (deterministic kernels) -- reported as `repeats_identical`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tools_metrics import matched_iou
    if k_masks != 20 or not timed:
        return None
    from tools_metrics import P0, binom_min_successes, wilson
    per, first, same = {}, {}, True
    for rec in timed:
        w, lab = rec[0], np.asarray(rec[1]).reshape(-1)
        runs = rec[2] if len(rec) > 2 else None
        if w in first:
            same = same and np.array_equal(first[w], lab)
            continue
        first[w] = lab
        path = os.path.join(ROOT, "tests", "golden", f"c3_t17_w{w}.npz" if svd else ("c2_window.npz" if w == 0 else f"c2_window_w{w}.npz"))
        if not os.path.exists(path):
            continue
        g = np.load(path)
        iou, exact = matched_iou(lab, g["corrected_labels" if refine else "match_labels"].astype(np.int64).reshape(-1), k_masks)
        per[w] = {"window": w, "iou": round(float(iou), 4), "identical_fraction": round(float(exact), 4)}
        if runs is not None and "restart_labels" in g.files:         # the ten K-means restarts against the reference's ten, index by index
            runs = runs.cpu().numpy().astype(np.int64)
            per[w]["restarts_in_place"] = int(sum(matched_iou(runs[i], g["restart_labels"][i].astype(np.int64), k_masks)[0] >= 0.99
                                                  for i in range(min(len(runs), len(g["restart_labels"])))))
    if not per:
        return None
    wins = [per[w] for w in sorted(per)]
    ious = np.array([x["iou"] for x in wins])
    n_ok, n = int((ious >= 0.99).sum()), len(wins)
    lo, hi = wilson(n_ok, n)
    rip = [x["restarts_in_place"] for x in wins if "restarts_in_place" in x]
    return {"iou": wins[0]["iou"], "identical_fraction": wins[0]["identical_fraction"], "windows": wins,
            "mean_iou": round(float(ious.mean()), 4), "median_iou": round(float(np.median(ious)), 4), "min_iou": round(float(ious.min()), 4),
            "windows_at_0.99": n_ok, "n_windows": n,
            "rate_at_0.99": round(n_ok / n, 4), "rate_95_interval": [round(lo, 4), round(hi, 4)],
            "reference_rate_under_1e-6_noise": round(P0, 4), "rejected_below": binom_min_successes(n),
            "mean_restarts_in_place": round(float(np.mean(rip)), 2) if rip else None,
            "mean_identical_fraction": round(float(np.mean([x["identical_fraction"] for x in wins])), 4),
            "repeats_identical": bool(same),
            "case": ("the masks of the TIMED steps (BASELINE configs[2] at full size: SVD 14x576x1024, K=20, t_start 17 = 8 CFG steps, full-width "
                     "VideoUNet) vs the reference's fp32 masks for the same windows (tests/golden/c3_t17_w*.npz, generated from /root/reference by "
                     "tools/gen_golden_c3_window.py --t-start 17)" if svd else
                     "the masks of the TIMED steps (BASELINE configs[1] at full size: 14x512x512, K=20, 3 CFG steps, full-width UNet; the steps "
                     "cycle over the fixture windows of the synthetic clip) vs the reference's fp32 masks for the same windows "
                     "(tests/golden/c2_window*.npz, generated from /root/reference by tools/gen_golden_c2_window.py)")
                    + ("; Step 3b (correct_low_res_mask) included" if refine else ""),
            "note": "best-of-10 K-means++ is chaotic in its input: a 1e-3 (fp16-level) change of the features re-rolls about half of the ten "
                    "restarts into other local optima (tools/mask_knee_study.py on the reference's own taps, tools/restart_study.py on the "
                    "device), so a window reads either ~1.0 or ~0.85; tests/test_gpu_c2_window.py asserts the restart-level equivalence"}


def dump_io_cost(F=F_WIN, lat=LAT):
    """What the reference's hand-off costs per sampler step on this host: `torch.save` of the Q/K dumps of decoder blocks 3-11
    (attn1.{k,q}, attn2.{k,q}: [2F, N, C] fp16; cross-attention k is [2F, 77, C]) plus x_t (sd_pipeline_vspw.py:103-139) into a
    temporary directory, then `torch.load` of the three tensors Step 3 reads (feature_extraction.py:646-668).  Host tensors: the
    device-to-host copy the reference also pays is not included.  The build keeps all of it in HBM (FeatureStore)."""
    tmp = tempfile.mkdtemp(prefix="vidseg_dump_", dir="/tmp")
    try:
        shapes = []
        for blocks, n, ch in (((3, 4, 5), (lat // 4) ** 2, 1280), ((6, 7, 8), (lat // 2) ** 2, 640), ((9, 10, 11), lat * lat, 320)):
            for b in blocks:
                for an, rows in (("self", n), ("cross", 77)):
                    shapes.append((f"output_block_{b}_spatial_{an}_attn_k_time_24.pt", (2 * F, rows, ch)))
                    shapes.append((f"output_block_{b}_spatial_{an}_attn_q_time_24.pt", (2 * F, n, ch)))
        tensors = [(nm, torch.zeros(sh, dtype=torch.float16).add_(0.5)) for nm, sh in shapes] + [("xt_time_24.pt", torch.zeros(F, 4, lat, lat))]
        nbytes = sum(t.numel() * t.element_size() for _, t in tensors)
        t0 = time.perf_counter()
        for nm, t in tensors:
            torch.save(t, os.path.join(tmp, nm))
        os.sync()
        t_save = time.perf_counter() - t0
        t0 = time.perf_counter()
        for b in (8, 7, 6):
            torch.load(os.path.join(tmp, f"output_block_{b}_spatial_self_attn_q_time_24.pt"))
        t_load = time.perf_counter() - t0
        return {"bytes_per_step": int(nbytes), "save_seconds_per_step": round(t_save, 3), "load_seconds_step3": round(t_load, 3),
                "steps_per_window": 3, "seconds_per_window": round(3 * t_save + t_load, 3),
                "note": "reference-style torch.save of one step's dumps (37 files) to a /tmp directory on this host + the three loads of Step 3; "
                        "host tensors (no device-to-host copy); not part of `cpu_baseline.value`, which hands the features over in memory"}
    except Exception as e:
        return {"error": repr(e)[:200]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def sd_flop_account(k_flops_per_step, evals, F=F_WIN, lat=LAT):
    """Reference-equivalent vs executed FLOPs per window (2*MAC).  Reference-equivalent: SURVEY.md section 8(d), counted on the
    reference's own modules with torch.utils.flop_counter: 22.52 TFLOP per CFG evaluation at config 2 (conv 52.0 %, linear 32.3 %,
    attention bmm 15.7 %).  Executed: the algorithmic 2*M*N*K of every conv / linear launch (summed inside the library) + the
    attention launches (4 * Nq * Nk * 64 per head).  The difference is the work the build legitimately does once instead of per
    evaluation: to_k / to_v of the step-constant context (16 cross-attentions x (evals - 1) evaluations), nothing else is skipped."""
    B = 2 * F
    attn = 0.0
    for n, heads, blocks in ((lat * lat, 5, 5), ((lat // 2) ** 2, 10, 5), ((lat // 4) ** 2, 20, 5), ((lat // 8) ** 2, 20, 1)):
        attn += blocks * (4.0 * n * n * 64 * heads * B + 4.0 * n * 77 * 64 * heads * B)
    ref_eval = 22.52e12 * (F / 14.0) * (lat / 64.0) ** 2
    kv = 0.0
    for ch, blocks in ((320, 5), (640, 5), (1280, 6)):
        kv += blocks * 2.0 * B * 77 * (2 * ch) * 1024
    return {"reference_equivalent_per_window": round(ref_eval * evals), "reference_equivalent_gemm_per_window": round(ref_eval * evals * 0.843),
            "executed_gemm_per_window": round(k_flops_per_step), "executed_attention_per_window": round(attn * evals),
            "not_re_executed_per_window": round(kv * (evals - 1)),
            "note": "reference-equivalent = SURVEY 8(d) count on the reference modules (conv + linear = 84.3 %); executed_gemm = sum of 2*M*N*K over "
                    "the launches of one window; not_re_executed = to_k / to_v of the constant cross-attention context, computed once per window "
                    "instead of once per evaluation (ops.window_cached); no other work is pruned or cached"}


def pmc_traffic(extra_args, timeout=240):
    """HBM-side bytes per launch of every GEMM kernel, measured now: two separate rocprofv3 --kernel-trace --pmc passes
    (FETCH_SIZE, then WRITE_SIZE) over one window of this workload in a child process.  rocprofv3 reports KB; on gfx950
    FETCH_SIZE counts a wide (16 B/lane) coalesced read stream at exactly half its bytes (MI355X_MICROARCH.md, HBM) and every
    operand load of these kernels is such a stream, so bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  Memory-side L2 counters
    (Infinity-Cache hits included): an upper bound on true HBM traffic.  Returns {kernel: (launches, bytes_per_launch)} or None."""
    import sqlite3
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="vidseg_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    sums = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--steps", "1", "--warmup", "0"] + extra_args
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
            if not dbs:
                return None
            db = sqlite3.connect(dbs[0])
            for name, val in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
                if "k_gemm" in name:
                    a = sums.setdefault(name.split("(")[0].replace("void ", ""), {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]})
                    a[counter][0] += 1
                    a[counter][1] += float(val)
        out = {}
        for k, a in sums.items():
            n = a["FETCH_SIZE"][0]
            if n and n == a["WRITE_SIZE"][0]:
                out[k] = (n, (2.0 * a["FETCH_SIZE"][1] + a["WRITE_SIZE"][1]) * 1024.0 / n, 2.0 * a["FETCH_SIZE"][1] * 1024.0 / n,
                          a["WRITE_SIZE"][1] * 1024.0 / n)
        return out or None
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def build(svd, narrow, dev):
    from vidseg_diffusion_amd import synthetic
    from vidseg_diffusion_amd.pipeline import build_sd_engine, build_svd_engine
    if svd:
        from vidseg_diffusion_amd.video_unet import VideoUNet
        cfg = dict(synthetic.SVD_NARROW if narrow else synthetic.SVD_FULL)
        net = VideoUNet(**cfg)
        zg = 1.0
    else:
        from vidseg_diffusion_amd.unet import UNetModel
        cfg = dict(synthetic.SD21_NARROW if narrow else synthetic.SD21_FULL)
        net = UNetModel(**cfg)
        zg = synthetic.HEADLINE["zero_gain"]
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd_cpu = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1234, zero_gain=zg).items()}
    net.load_state_dict(sd_cpu)
    net.pack(dev)
    eng = build_svd_engine(net, num_frames=F_WIN, num_steps=NUM_STEPS) if svd else build_sd_engine(net, num_steps=NUM_STEPS, scale=5.0)
    return eng, cfg, sd_cpu, sum(int(np.prod(s)) for s in shapes.values())


def fp8_attention_roofline(dev, B=28, H=5, N=9216, reps=10):
    """The e4m3 attention kernel of configs[4] alone, at the SVD window's largest spatial self-attention (72x128 = 9216 tokens, 5 heads,
    CFG batch 28), operands already quantised: HIP events on the launch stream, algorithmic FLOPs 4 * Nq * Nk * 64 per head against the
    dense MX-fp8 MFMA peak (5 PFLOP/s, MI355X_MICROARCH.md)."""
    from vidseg_diffusion_amd import ops
    from vidseg_diffusion_amd._lib import call, ptr, stream
    C = H * 64
    g = torch.Generator().manual_seed(5)
    q8, k8, v8 = (ops.quant_fp8(torch.randn((B, N, C), generator=g).to(dev).to(ops.act_dtype())) for _ in range(3))
    out = torch.empty((B, N, C), dtype=ops.act_dtype(), device=dev)

    def launch():
        call("vidseg_attention_fp8", ptr(q8), C, ptr(k8), C, ptr(v8), C, ptr(out), C, B, H, N, N, 64, stream())
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    fl = 4.0 * N * N * 64 * H * B
    return {"bound": "mfma", "kernel": "k_attention_mx8 (e4m3 q, k, v, P; block-scaled 32x32x64 MFMA)", "achieved": round(fl / us / 1e6, 1), "peak": 5000.0,
            "unit": "TFLOP/s", "frac": round(fl / us / 1e6 / 5000.0, 4), "avg_launch_us": round(us, 1), "shape": {"B": B, "heads": H, "Nq": N, "Nk": N},
            "algorithmic_bytes": int(3 * B * N * C + 2 * B * N * C),
            "note": "kernel alone, back-to-back launches outside the window (in the window the quantisation of q, k, v adds three small launches)"}


def post_unet_roofline(dev, F=F_WIN, fh=LAT // 2, fw=LAT // 2, C=640, K=20, reps=20):
    """The post-UNet kernels alone at the headline's sizes ("HBM fraction for the post-UNet kernels"), torch
    events on the launch stream, back-to-back launches on synthetic dumps: algorithmic bytes (every operand and result once) / time
    against the 8 TB/s HBM peak, and for the float64 contractions their FLOPs against the 78.6 TFLOP/s f64 MFMA peak
    (MI355X_MICROARCH.md) -- each kernel is priced against the roofline that bounds it (`bound`)."""
    from vidseg_diffusion_amd import analysis as A
    from vidseg_diffusion_amd import synthetic
    N, n = fh * fw, F * fh * fw
    blocks, _ = synthetic.attention_q_dumps(F, fh, fw, C, num_blocks=3, seed=1)
    dumps = [torch.from_numpy(b).to(dev) for b in blocks]                       # [2F, N, C] fp16 each
    _, feat = A.mean_normalize(dumps, n, n)
    centers = feat[torch.randperm(n, generator=torch.Generator().manual_seed(3))[:K].to(dev)].double().contiguous()
    ref_labels = torch.randint(0, K, (N,), generator=torch.Generator().manual_seed(4), dtype=torch.int32).to(dev)
    cond7 = dumps[1][F:].contiguous()

    def timeit(fn, r=reps):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(r):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / r                                      # microseconds per call

    rows = []

    def row(kernel, ref, bound, us, nbytes, flops=None):
        gbs = nbytes / us / 1e3
        d = {"kernel": kernel, "reference": ref, "bound": bound, "us": round(us, 1), "algorithmic_bytes": int(nbytes), "GB/s": round(gbs, 1),
             "frac_hbm": round(gbs / 8000.0, 4)}
        if flops is not None:
            d.update({"f64_flops": int(flops), "f64_TFLOP/s": round(flops / us / 1e6, 2), "frac_f64_mfma": round(flops / us / 1e6 / 78.6, 4)})
        rows.append(d)

    row("k_mean_normalize", "FE:739-748, 550-555 (3-block mean + max-abs normalise, conditional half)", "hbm",
        timeit(lambda: A.mean_normalize(dumps, n, n)), 3 * n * C * 2 + n * C * 2)
    row("k_lloyd_assign (one full E-step, one restart)", "sklearn _kmeans.py lloyd_iter via FE:562-572", "f64 mfma / latency",
        timeit(lambda: A.kmeans_predict(feat, centers)), n * C * 2 + K * C * 8 + n * 4, 2.0 * n * K * C)
    row("k_knn (+ 2 row-norm passes): 4-NN of the window's tokens in frame 0", "FE:603-613", "f64 mfma",
        timeit(lambda: A.knn_predict(feat[:N], ref_labels, feat), 5), (n + N) * C * 2 + n * 4, 2.0 * n * N * C)
    row("k_track_normalize + 13 x (k_track_cos, k_row_select): dense tracking of block 7", "FE:176-364", "f64 mfma",
        timeit(lambda: A.dense_tracking(cond7, F, fh, fw), 3), n * C * 2 * 2 + (F - 1) * (2 * N * C * 2 + N * N * 2), (F - 1) * 2.0 * 2.0 * N * N * C)
    return {"peak_hbm_GB/s": 8000.0, "peak_f64_mfma_TFLOP/s": 78.6, "kernels": rows,
            "note": "kernels alone, back-to-back on synthetic dumps of the headline's sizes (14 x 32 x 32 tokens, C = 640, K = 20); in the window they run "
                    "on the analysis stream under the next window's UNet.  k_mean_normalize is the one pure streaming kernel of the stage; the "
                    "others are float64 contractions (exactness is the point: bit-exact labels) priced against the f64 MFMA peak"}


def step45_sd(eng, cfg, dev, k_masks, ab_labels=3):
    """Steps 4-5 of one SD window (BASELINE configs[1] geometry) in the parity precision, outside `value`: after Steps 1-3 with every step's
    dumps kept (sd_pipeline_vspw.py:336-405), the 2K modulated sampler passes (SDP:416-515: lambda * mask on decoder block 7's
    cross-attention output at step 22, the feature pass's q / k dumps injected into decoder blocks 1-11, latents blended with the feature
    pass's x_t outside the mask at steps 22-23), each final latent through the first stage's decoder (SDP:150-152), the per-label
    difference maps and the arg-max over labels (process_output.py:8-167).  The sweep is timed twice: `ab_labels` labels with every pass
    in full with their Q/K taps written (share_prefix=False, keep_taps=True: the sweep as round 5 ran it) and all K labels as the sweep
    runs now: the first evaluation's prefix shared, no tap stores, two passes in flight (pipeline.modulation_sweep)."""
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd import process_output as PO
    from vidseg_diffusion_amd import synthetic
    from vidseg_diffusion_amd.pipeline import modulation_sweep, segment_window
    from vidseg_diffusion_amd.vae import AutoencoderKL, decode_first_stage
    net = eng.model.diffusion_model
    prec0 = net.precision
    dd = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
              attn_resolutions=[], dropout=0.0)
    vae = AutoencoderKL(embed_dim=4, ddconfig=dd)
    vshapes = {k: tuple(v.shape) for k, v in vae.state_dict().items()}
    vae.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(vshapes, seed=99).items()})
    sync = torch.cuda.synchronize
    try:
        net.set_precision("exact")
        FE.FeatureStore.clear()
        FE.MaskStore.clear()
        lat, c, uc, noise = make_inputs(dev, 0, cfg)
        base, exp = "/nonexistent/bench_step45", "w0"
        skw = dict(num_masks=k_masks, num_steps=NUM_STEPS, t_start=22, seed=17, noise=noise, feature_folder=base, exp_name=exp, keep_all_steps=True)
        segment_window(eng, lat, c, uc, **skw)
        sync()
        t0 = time.perf_counter()
        lab, _ = segment_window(eng, lat, c, uc, **skw)
        sync()
        t_feat = time.perf_counter() - t0
        folder = os.path.join(base, exp, "match_gt_mask", f"output_block_8_output_block_7_output_block_6_spatial_self_attn_q_masks_{k_masks}")
        labels = [int(v) for v in np.unique(lab)]
        kw = dict(t_start=22, num_steps=NUM_STEPS, feature_folder=base, exp_name=exp, noise=noise, seed=17)
        modulation_sweep(eng, lat, c, uc, labels[:1], folder, **kw)                          # warm-up (allocator, first-touch)
        decode_first_stage(vae, lat, 0.18215)
        sync()
        t0 = time.perf_counter()
        plain = modulation_sweep(eng, lat, c, uc, labels[:ab_labels], folder, share_prefix=False, keep_taps=True, lanes=1, **kw)   # as round 5 ran it
        sync()
        t_plain = (time.perf_counter() - t0) / (2 * ab_labels)
        t0 = time.perf_counter()
        res = modulation_sweep(eng, lat, c, uc, labels, folder, share_prefix=True, **kw)
        sync()
        t_sweep = time.perf_counter() - t0
        same = all(torch.equal(plain[k], res[k]) for k in plain)
        t0 = time.perf_counter()
        maps, maxima = [], []
        for lb in labels:
            m, mx = PO.difference_map(decode_first_stage(vae, res[(1, lb)], 0.18215), decode_first_stage(vae, res[(-1, lb)], 0.18215))
            maps.append(m)
            maxima.append(mx)
        sync()
        t_dec = time.perf_counter() - t0
        t0 = time.perf_counter()
        seg = PO.seg_map(torch.stack(maps), torch.stack(maxima), labels)
        sync()
        t_arg = time.perf_counter() - t0
        n = 2 * len(labels)
        total = t_sweep + t_dec + t_arg
        return {"config": "SD configs[1] geometry (14 x 512 x 512, K = 20), precision exact", "passes_per_window": n,
                "sweep_ms_per_window": round(1e3 * t_sweep, 1), "ms_per_pass": round(1e3 * t_sweep / n, 2),
                "ms_per_pass_every_pass_in_full": round(1e3 * t_plain, 2),
                "prefix_saving_frac": round(1.0 - (t_sweep / n) / t_plain, 4), "bit_identical_to_unshared": bool(same),
                "decode_and_difference_ms_per_window": round(1e3 * t_dec, 1), "argmax_ms_per_window": round(1e3 * t_arg, 2),
                "steps_4_5_ms_per_window": round(1e3 * total, 1), "steps_1_3_all_dumps_ms_per_window": round(1e3 * t_feat, 1),
                "frames_per_s_steps_1_to_5": round(F_WIN / (t_feat + total), 3), "labels_in_segmentation_map": int(len(torch.unique(seg))),
                "note": "2K modulated passes of 3 CFG evaluations each (block 7 cross-attention, lambda 50, q / k injection into decoder blocks "
                        "1-11, latent blending at steps 22-23) + 2K first-stage decodes (16-bit kernels) + difference maps + arg-max; the first "
                        "evaluation's prefix (encoder, middle, decoder blocks 0-6, block 7's ResBlock) is computed once and resumed by the other "
                        "2K - 1 passes, no pass writes Q/K taps (nothing reads them) and two passes are in flight on their own HIP streams; `ms_per_pass_every_pass_in_full` = the same passes as round 5 "
                        "ran them, every evaluation in full with its taps (timed on %d labels, bit-identical latents); outside `value`" % ab_labels}
    except Exception as e:
        return {"error": repr(e)[:300]}
    finally:
        net.set_precision(prec0)
        FE.FeatureStore.clear()
        FE.MaskStore.clear()


def run_config(args, svd, rank, world, dev, steps, warmup, secondary=False, prebuilt=None):
    """Time `steps` steps of one config; returns (out dict for rank 0 | None, sd_cpu, cfg, eng, labels)."""
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd import ops, parallel
    from vidseg_diffusion_amd.pipeline import WindowPipeline
    k_masks = args.masks or 20
    t_start = 17 if svd else 22
    refine = True if svd else args.refine
    eng, cfg, sd_cpu, n_params = prebuilt or build(svd, args.narrow, dev)
    eng.model.diffusion_model.set_precision("exact" if args.exact else "fp16")
    # The timed steps CYCLE over the windows of the synthetic clip (step i of rank r = window (i * world + r) mod n): K-means
    # iteration counts, restarts and tie replays are data dependent, so one repeated window would time -- and score -- a single
    # draw.  SD headline: the windows for which the reference's labels are committed (tests/golden/c2_window*.npz), so that
    # `value` and `mask_iou_vs_reference` describe the same work; SVD: windows 0..2.  All inputs are resident in HBM beforehand.
    if svd and not args.narrow:
        win_ids = sorted(int(np.load(p)["window_id"]) for p in glob.glob(os.path.join(ROOT, "tests", "golden", "c3_t17_w*.npz"))) or list(range(3))
    elif args.narrow:
        win_ids = list(range(3))
    else:
        win_ids = sorted(int(np.load(p)["window_id"]) for p in glob.glob(os.path.join(ROOT, "tests", "golden", "c2_window*.npz"))) or [0]
    if args.one_window or args.pmc_child:
        win_ids = win_ids[:1]
    inputs = {w: make_inputs(dev, w, cfg, svd=svd, lat_hw=(72, 128) if svd else (LAT, LAT)) for w in win_ids}
    c, uc = inputs[win_ids[0]][1], inputs[win_ids[0]][2]
    if not svd:                                                      # one conditioning for the clip (sd_conditioning(seed=1)): share the tensors
        inputs = {w: (v[0], c, uc, v[3]) for w, v in inputs.items()}
    torch.cuda.synchronize()
    lanes = 1 if (args.no_overlap or args.pmc_child) else max(1, args.lanes if args.lanes is not None else (1 if svd else 2))
    overlap = not args.no_overlap and not args.pmc_child
    fkw = dict(num_steps=NUM_STEPS, t_start=t_start, seed=17, masks_only=args.masks_only)
    if args.inversion:
        fkw["inversion_type"] = "inversion"
    step_no = [0]
    record = []                                                      # (window id, labels) of every step run since the last reset

    def last_restarts():
        """The ten restarts' labels of the K-means that just ran (a reference to the device tensor: nothing is copied inside the timing)."""
        from vidseg_diffusion_amd import analysis as A
        km, A.LAST_KMEANS = A.LAST_KMEANS, None                      # consumed: a chained (4-NN only) window reads None, never an earlier window's restarts
        return km.all_labels if km is not None else None

    def next_window():
        w = win_ids[(step_no[0] * world + rank) % len(win_ids)]
        step_no[0] += 1
        return w, f"r{rank}s{step_no[0] % 6}", inputs[w]

    if world == 1 and overlap:
        # windows run through pipeline.WindowPipeline: the analysis of step i (second HIP stream) overlaps the feature passes of the
        # next `lanes` steps; every step is still a complete window (K-means included) and all of them finish inside the timed region
        def run_steps(n, chain=False, nl=None):
            pipe = WindowPipeline(eng, chain=chain, lanes=nl or lanes, num_masks=k_masks, is_aggre_attn=True, is_refine_mask=refine)
            last, fifo = None, []
            for _ in range(n):
                FE.MaskStore.clear()
                w, nm, (lat, cw, ucw, noise) = next_window()
                fifo.append(w)
                got = pipe.push(lat, cw, ucw, keep_all_steps=False, exp_name=nm, noise=noise, **fkw)
                if got is not None:
                    record.append((fifo.pop(0), got, last_restarts()))
                    last = got
            for got in pipe.drain():                                 # (drain analyses one window after the other: the last K-means is the last window's)
                record.append((fifo.pop(0), got, None))
                last = got
            return last
    elif world > 1 and overlap:
        def run_steps(n, chain=False, nl=None):
            spipe = parallel.ShardedPipeline(eng, rank, world, lanes=nl or lanes, num_masks=k_masks, is_aggre_attn=True, is_refine_mask=refine)
            last = None
            for _ in range(n):
                FE.MaskStore.clear()
                w, nm, (lat, cw, ucw, noise) = next_window()
                got = spipe.push(lat, cw, ucw, exp_name=nm, noise=noise, **fkw)
                last = got if got is not None else last
            rest = spipe.drain()
            return rest[-1] if rest else last
    else:
        def run_steps(n, chain=False, nl=None):
            last = None
            for _ in range(n):
                FE.FeatureStore.clear()
                FE.MaskStore.clear()
                w, nm, (lat, cw, ucw, noise) = next_window()
                last = parallel.segment_windows_sharded(eng, lat, cw, ucw, noise=noise, num_masks=k_masks, num_steps=NUM_STEPS, t_start=t_start,
                                                        is_aggre_attn=True, is_refine_mask=refine, seed=17, rank=rank, world=world,
                                                        masks_only=args.masks_only, inversion_type=fkw.get("inversion_type", "add_noise"))
                if world == 1:
                    record.append((w, last, last_restarts()))
            return last

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    run_steps.record, run_steps.windows, run_steps.fkw = record, win_ids, fkw   # the secondary timings of main() read the masks they produced
    if warmup:
        run_steps(warmup)
    barrier()
    del record[:]
    prof_in_region = lanes == 1 and os.environ.get("VIDSEG_BENCH_NOPROF") != "1"   # one lane: a launch's HIP-event time is its own
    if prof_in_region:
        ops.gemm_profile_begin()
    t0 = time.perf_counter()
    labels = run_steps(steps)
    barrier()
    elapsed = time.perf_counter() - t0
    prof_steps, single_lane = steps, None
    if prof_in_region:
        k_ms, k_flops, k_launches = ops.gemm_profile_end()
        kinds = ops.gemm_profile_kinds()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timed = list(record)                                            # (window id, labels) of exactly the timed steps, in order
    if not prof_in_region and not args.pmc_child:
        # the roofline's per-launch HIP events: a single-lane pass over the same windows, outside the timed region (two lanes share
        # the chip, so inside it a launch's event time would include waiting for the other lane's kernels)
        prof_steps = max(2, min(steps, 8)) if not svd else max(1, min(steps, 2))
        run_steps(1, nl=1)
        barrier()
        ops.gemm_profile_begin()
        t1 = time.perf_counter()
        run_steps(prof_steps, nl=1)
        barrier()
        dt1 = time.perf_counter() - t1
        k_ms, k_flops, k_launches = ops.gemm_profile_end()
        kinds = ops.gemm_profile_kinds()
        single_lane = {"value": round(F_WIN * world * prof_steps / dt1, 3), "unit": "frames/s", "ms_per_step": round(1e3 * dt1 / prof_steps, 3),
                       "steps": prof_steps,
                       "note": "the same windows with ONE feature pass in flight (pipeline.WindowPipeline(lanes=1)) and the GEMM profiler's HIP "
                               "events on every conv / linear launch: the pass the `roofline` object is measured in"}
    elif args.pmc_child:
        k_ms, k_flops, k_launches, kinds = 0.0, 0.0, 0, []
    if rank != 0 or args.pmc_child:
        return None, sd_cpu, cfg, eng, labels, run_steps, timed

    frames = F_WIN * world * steps
    fam_tf = (k_flops / (k_ms * 1e-3)) / 1e12 if k_ms > 0 else 0.0
    dom = max(kinds, key=lambda r: r[1])                                      # the single kernel with the most time in the region
    achieved = (dom[2] / (dom[1] * 1e-3)) / 1e12 if dom[1] > 0 else 0.0
    evals = 8 if svd else 3
    if args.inversion:
        evals = (NUM_STEPS - 1) + NUM_STEPS                          # sampler.inversion skips the network on its first pair (SAM:110-111)
    if svd:
        metric = f"segmented frames/sec (14-frame 576x1024 clip, {k_masks} masks, SVD)"
        workload = (f"BASELINE configs[2]: SVD img2vid full-size VideoUNet ({n_params / 1e6:.1f}M params, random-init), 14-frame 576x1024 "
                    f"window per GPU (latent 14x4x72x128), t_start=17 (8 CFG UNet evals, batch 28), spatial+temporal taps, "
                    f"is_aggre_attn, K-means K={k_masks} + 4-NN, is_refine_mask (dense tracking + vote)")
    else:
        metric = f"segmented frames/sec (14-frame 512^2 clip, {k_masks} masks)"
        workload = (f"BASELINE configs[1]: SD 2.1 full-size UNet ({n_params / 1e6:.1f}M params, synthetic near-init weights), 14-frame "
                    f"512x512 window per GPU (latent 14x4x64x64, synthetic 20-region clip), 25-step schedule with t_start=22 (3 CFG UNet "
                    f"evals, batch 28), Q/K taps on decoder blocks 3-11, is_aggre_attn (blocks 6,7,8), K-means K={k_masks} n_init=10 + 4-NN"
                    + (", is_refine_mask" if refine else ""))
    out = {
        "metric": metric, "value": round(frames / elapsed, 3), "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(1e3 * elapsed / steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f16" if ops.act_dtype() == torch.float16 else "bf16") if not args.exact else
                 "f32 carried as split f16 pairs (22-bit operands on the f16 MFMA, f32 accumulation)", "data": "synthetic",
        "config": {"workload": workload, "frames_per_gpu": F_WIN, "num_masks": k_masks, "unet_evals_per_step": evals,
                   "parallelism": f"window-per-gpu x{world}",
                   "overlap": (f"{lanes} feature pass(es) in flight on their own HIP streams; the analysis of a finished window runs on another "
                               f"stream meanwhile; every step is a complete window and all finish inside the timed region") if overlap
                   else "none (each window's analysis follows its own feature pass)"},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": 2500.0, "unit": "TFLOP/s",
                     "frac": round(achieved / 2500.0, 4), "traffic": None, "kernel": dom[0],
                     "launches_per_step": dom[3] // max(prof_steps, 1),
                     "avg_launch_us": round(1e3 * dom[1] / max(dom[3], 1), 2),
                     "ms_per_step": round(dom[1] / prof_steps, 3),
                     "measured_in": ("the timed region (one lane)" if prof_in_region else
                                     f"a single-lane pass of {prof_steps} steps over the same windows right after the timed region (`single_lane`)"),
                     "algorithmic_bytes": int(dom[4] / max(dom[3], 1)),
                     # every conv / linear of the UNet runs on this kernel family; the family-wide figures:
                     "family": {"achieved": round(fam_tf, 2), "frac": round(fam_tf / 2500.0, 4),
                                "launches_per_step": k_launches // max(prof_steps, 1), "gemm_ms_per_step": round(k_ms / prof_steps, 3),
                                "by_kernel": {n: {"ms_per_step": round(ms / prof_steps, 3), "tflops": round(fl / ms / 1e9, 1) if ms > 0 else 0.0,
                                                  "launches_per_step": ln // max(prof_steps, 1),
                                                  "algorithmic_bytes_per_launch": int(_b / ln)} for (n, ms, fl, ln, _b) in kinds if ln}}},
        "unique_labels": int(len(np.unique(labels))),
    }
    if not svd and not args.narrow:
        out["flops"] = sd_flop_account(k_flops / max(prof_steps, 1), evals)
        if args.exact:
            out["flops"]["note"] += ("; precision exact: executed_gemm counts the three fp16 products per fp32-accurate product (K axis 3x)"
                                     + ("; masks_only: the last evaluation executes the conditional half up to decoder block 8 only, "
                                        "reference_equivalent still counts the three full CFG evaluations" if args.masks_only else ""))
    if single_lane is not None:
        out["single_lane"] = single_lane
    if args.masks_only:
        if not args.parity:
            out["metric"] += " [masks-only pruning]"
        out["config"]["workload"] += ("; DEAD-WORK PRUNING (masks_only): the last UNet evaluation (step 24, whose only consumers in Steps 3-3b "
                                      "are the conditional half's Q taps of decoder blocks 6-8) runs on the conditional half only and stops "
                                      "after decoder block 8; the unconditional half, blocks 9-11, the output conv, the CFG combine and the "
                                      "Euler update of that step feed nothing the masks are made of (SURVEY 8(d): executed vs "
                                      "reference-equivalent FLOPs are both reported in `flops`; `full_schedule` times the unpruned pass)")
        out["config"]["unet_evals_per_step"] = f"{evals - 1} full + 1 taps-only (cond half, blocks <= 8)"
    if args.fp8_attn:
        out["dtype"] += " + e4m3 attention (q, k, v, P; >= 1024 keys)"
        out["config"]["workload"] += "; BASELINE configs[4] attention path: OCP e4m3 MFMA on the spatial self-attentions"
    out["config"]["windows_cycled"] = win_ids
    if svd and world == 1 and not args.narrow and (not args.masks_only or args.parity) and not args.fp8_attn and not args.no_step4:
        # BASELINE configs[2] names "is_refine_mask + latent blending": the blending lives in Step 4's modulated sampler passes
        # (sampling.py:229-250; svd_pipeline_vspw.py:396-487 -- 2*K of them per window).  One label's +lambda / -lambda pair is run and
        # timed here with the SVD driver's defaults (block 8, spatial + temporal self-attention rows, modulate_timestep 17 = t_start,
        # latent blending on, no feature injection), after an untimed window that keeps the x_t of every step in HBM for it.
        net4 = eng.model.diffusion_model
        prec4 = net4.precision
        stage(f"step 4 (modulated passes, precision {prec4})")
        try:
            from vidseg_diffusion_amd.pipeline import modulation_sweep, segment_window
            FE.FeatureStore.clear()
            FE.MaskStore.clear()
            lat, cw, ucw, noise = inputs[win_ids[0]]
            base, exp = "/nonexistent/bench_step4", "w0"
            lab, _ = segment_window(eng, lat, cw, ucw, num_masks=k_masks, num_steps=NUM_STEPS, t_start=t_start, seed=17, noise=noise,
                                    is_refine_mask=refine, feature_folder=base, exp_name=exp, keep_all_steps=True)
            folder = os.path.join(base, exp, "match_gt_mask", f"output_block_8_output_block_7_output_block_6_spatial_self_attn_q_masks_{k_masks}"
                                  + ("_corrected" if refine else ""))
            if FE.MaskStore.get(folder) is None:
                folder = folder.replace("_corrected", "")
            label = int(np.unique(lab)[0])
            kw = dict(t_start=t_start, num_steps=NUM_STEPS, modulate_block_idx=(8,), modulate_layer_type=("spatial", "temporal"),
                      modulate_attn_type=("self_attn",), is_injected_features=False, is_latent_blending=True, feature_folder=base, exp_name=exp,
                      noise=noise, seed=17)
            two = [int(v) for v in np.unique(lab)[:2]]
            modulation_sweep(eng, lat, cw, ucw, [label], folder, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = modulation_sweep(eng, lat, cw, ucw, [label], folder, share_prefix=False, lanes=1, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            t0 = time.perf_counter()
            res2 = modulation_sweep(eng, lat, cw, ucw, two, folder, share_prefix=True, **kw)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            n2 = 2 * len(two)
            resumed = (dt2 - dt / 2) / max(n2 - 1, 1)                                # a pass that resumes the shared prefix
            out["step4_latent_blending"] = {"ms_per_modulated_pass": round(1e3 * dt / 2, 2), "passes_per_window": 2 * k_masks, "precision": prec4,
                                            "ms_per_pass_resuming_shared_prefix": round(1e3 * resumed, 2),
                                            "sweep_ms_per_window_extrapolated": round(1e3 * (dt / 2 + (2 * k_masks - 1) * resumed), 1),
                                            "bit_identical_to_unshared": bool(all(torch.equal(res[k], res2[k]) for k in res)),
                                            "finite": bool(all(torch.isfinite(v).all().item() for v in res2.values())),
                                            "note": "the +lambda / -lambda modulated sampler passes of a label (8 CFG evaluations each, lambda*mask added "
                                                    "to the spatial and temporal self-attention rows of decoder block 8 at step 17, latents blended "
                                                    "with the feature pass's x_t outside the mask at every step): the Step 4 unit of configs[2]; "
                                                    "outside `value` (Steps 1-3b), which the metric is quoted on; same precision mode as `value`; "
                                                    "ms_per_modulated_pass = every pass in full (one label), ms_per_pass_resuming_shared_prefix = a pass "
                                                    "that resumes the first evaluation's shared prefix (pipeline.modulation_sweep, timed on two labels); "
                                                    "the 2K-pass sweep extrapolated = 1 full + (2K - 1) resumed; tests/test_gpu_c3_window.py::"
                                                    "test_step4_latent_blending_full_size pins these passes against the reference's own run"}
        except Exception as e:
            out["step4_latent_blending"] = {"error": repr(e)[:300]}
        finally:
            net4.set_precision(prec4)
            FE.FeatureStore.clear()
            FE.MaskStore.clear()
    if args.exact:
        out["config"]["precision"] = "exact"
        out["config"]["workload"] += ("; PRECISION exact (exact.py): fp32 activations, every conv / linear ONE call of the 16-bit MFMA kernels over "
                                      "split (hi, lo) operands ([a_hi|a_lo|a_hi] x [w_hi|w_hi|w_lo], fp32 accumulation) -- 3x the MFMA work of the "
                                      "16-bit mode; the mode whose masks are the reference's (`mask_iou_vs_reference`)")
        out["roofline"]["note_flops"] = ("achieved / frac count the MFMA work the kernel executes (2*M*N*3K per launch: three fp16 products per "
                                         "fp32-accurate product); `fp32_equivalent` = the same launches counted as 2*M*N*K")
        out["roofline"]["fp32_equivalent"] = {"achieved": round(achieved / 3.0, 2), "frac_of_f16_mfma_peak": round(achieved / 3.0 / 2500.0, 4),
                                              # the chip's own fp32-input matrix rate (v_mfma_f32_32x32x2_f32: 157.3 TFLOP/s dense, MI355X_MICROARCH.md)
                                              # is what an fp32 GEMM of the reference's precision could reach without the split
                                              "vs_f32_mfma_peak_157.3": round(achieved / 3.0 / 157.3, 2)}
        out["roofline"]["frac_algorithmic"] = round(achieved / 3.0 / 2500.0, 4)     # reference-equivalent FLOPs (2*M*N*K) / time / peak
        out["roofline"]["family"]["frac_algorithmic"] = round(fam_tf / 3.0 / 2500.0, 4)
    else:
        out["config"]["precision"] = "fp16"
        out["roofline"]["frac_algorithmic"] = out["roofline"]["frac"]
        out["metric"] += " [precision=fp16: masks NOT at parity with the reference, see mask_iou_vs_reference]"
    if args.inversion:
        out["metric"] += " [inversion_type=inversion: 49 UNet evaluations per window]"
        out["config"]["workload"] += ("; INVERSION VARIANT (sd_pipeline_vspw.py:233-236, 340-345): EulerEDMSampler.inversion over 25 sigma pairs "
                                      "(24 evaluations) then the feature pass from t_start = 0 (25 evaluations, Q/K dumps kept for the step that "
                                      "is read)")
    return out, sd_cpu, cfg, eng, labels, run_steps, timed


def stage(msg):
    """Progress on stderr (stdout carries the one JSON line): which leg of the run a failure belongs to."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


LINE_LIMIT = 6144                                   # bytes of the final stdout line (the driver keeps a bounded tail of stdout)
FULL_RECORD = "bench_full.json"                     # every object of the run (per-window lists, notes, per-kernel tables), beside bench.py


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def compact_line(full):
    """The ONE line the driver parses: the contract's keys + `roofline` + `cpu_baseline` + the mask score + one-line summaries of
    the side measurements, always < LINE_LIMIT bytes and strict JSON (no NaN).  Everything else of the run lives in FULL_RECORD."""
    out = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data",
                       "rccl_ranks", "dist_backend", "unique_labels"))
    out["vs_baseline"] = full.get("vs_baseline")
    out["metric"] = _short(out.get("metric", ""), 160)
    cfg = full.get("config", {})
    out["config"] = _pick(cfg, ("frames_per_gpu", "num_masks", "unet_evals_per_step", "parallelism", "precision"))
    out["config"]["workload"] = _short(cfg.get("workload", ""), 420)
    if "windows_cycled" in cfg:
        out["config"]["windows_cycled"] = len(cfg["windows_cycled"])
    r = full.get("roofline", {})
    out["roofline"] = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "frac_algorithmic", "avg_launch_us", "launches_per_step", "ms_per_step",
                                "algorithmic_bytes"))
    out["roofline"]["traffic"] = r.get("traffic")
    out["roofline"]["kernel"] = _short(str(r.get("kernel", "")).split(" (")[0], 60)
    out["roofline"]["measured_in"] = _short(r.get("measured_in", ""), 110)
    fam = r.get("family")
    if fam:
        out["roofline"]["family"] = _pick(fam, ("achieved", "frac", "frac_algorithmic", "launches_per_step", "gemm_ms_per_step"))
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "cpu", "value_bf16_autocast"))
        out["cpu_baseline"]["sample"] = _short(cb.get("sample", ""), 260)
    mk = ("n_windows", "windows_at_0.99", "mean_iou", "min_iou", "median_iou", "mean_identical_fraction", "repeats_identical")
    if full.get("mask_iou_vs_reference"):
        out["mask_iou_vs_reference"] = _pick(full["mask_iou_vs_reference"], mk + ("rate_95_interval",))
    if full.get("flops"):
        out["flops"] = _pick(full["flops"], ("reference_equivalent_per_window", "executed_gemm_per_window", "executed_attention_per_window"))
    for key in ("full_schedule", "fast_mode", "exact_mode", "parity_mode", "single_lane", "chained_window", "two_lanes"):
        v = full.get(key)
        if isinstance(v, dict):
            out[key] = _pick(v, ("value", "ms_per_step", "steps", "precision", "masks_only", "error"))
            if isinstance(v.get("mask_iou_vs_reference"), dict):
                out[key].update(_pick(v["mask_iou_vs_reference"], ("n_windows", "windows_at_0.99", "mean_iou", "min_iou")))
    for key in ("masks_vs_full_schedule", "fp8_vs_16bit_masks", "first_stage"):
        if isinstance(full.get(key), dict):
            out[key] = {k: v for k, v in full[key].items() if not isinstance(v, str)}
    for key in ("secondary", "secondary_fp8"):
        v = full.get(key)
        if not isinstance(v, dict):
            continue
        o = _pick(v, ("value", "unit", "steps", "warmup", "ms_per_step", "n_gpus", "rccl_ranks", "scaling", "error"))
        if "metric" in v:
            o["metric"] = _short(v["metric"], 110)
        if isinstance(v.get("config"), dict):
            o["precision"] = v["config"].get("precision")
            o["parallelism"] = v["config"].get("parallelism")
        if isinstance(v.get("roofline"), dict):
            o["roofline"] = _pick(v["roofline"], ("achieved", "peak", "frac", "frac_algorithmic", "avg_launch_us"))
            o["roofline"]["kernel"] = _short(str(v["roofline"].get("kernel", "")).split(" (")[0], 40)
        if isinstance(v.get("mask_iou_vs_reference"), dict):
            o["mask_iou_vs_reference"] = _pick(v["mask_iou_vs_reference"], ("n_windows", "windows_at_0.99", "mean_iou", "min_iou"))
        if isinstance(v.get("fast_mode"), dict):
            o["fast_mode"] = _pick(v["fast_mode"], ("value", "ms_per_step"))
        if isinstance(v.get("step4_latent_blending"), dict):
            o["step4_latent_blending"] = _pick(v["step4_latent_blending"], ("ms_per_modulated_pass", "ms_per_pass_resuming_shared_prefix",
                                                                             "sweep_ms_per_window_extrapolated", "passes_per_window", "precision", "error"))
        out[key] = o
    if isinstance(full.get("step45"), dict):
        out["step45"] = {k: v for k, v in full["step45"].items() if not isinstance(v, (str, dict, list))}
    post = full.get("roofline_post_unet")
    if isinstance(post, dict) and isinstance(post.get("kernels"), list):
        out["roofline_post_unet"] = [{"kernel": _short(k["kernel"].split(" ")[0], 24), "us": k["us"], "frac_hbm": k["frac_hbm"],
                                      **_pick(k, ("frac_f64_mfma",))} for k in post["kernels"]]
    out["full_record"] = FULL_RECORD
    line = json.dumps(out, allow_nan=False)
    for drop in ("roofline_post_unet", "two_lanes", "chained_window", "single_lane", "first_stage", "flops", "fast_mode", "secondary_fp8"):
        if len(line) < LINE_LIMIT:
            break
        out.pop(drop, None)
        line = json.dumps(out, allow_nan=False)
    assert len(line) < LINE_LIMIT, len(line)
    return line


def _finite(o):
    """NaN / inf -> None (strict JSON for the driver's parser)."""
    if isinstance(o, float):
        return o if np.isfinite(o) else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    if isinstance(o, np.generic):
        return _finite(o.item())
    return o


def emit(full):
    """Write the full record beside bench.py (and under gpurun_out/ when that exists, so it travels back from the GPU box), then print
    the compact line LAST on stdout."""
    full = _finite(full)
    text = json.dumps(full, allow_nan=False)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, FULL_RECORD), "w") as fh:
                    fh.write(text + "\n")
        except OSError:
            pass
    sys.stdout.flush()
    print(compact_line(full), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)       # (5 / 2 until round 6: a single 150 ms hiccup in a 0.8 s timed region read as -18 %, profiles/r06_c)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--refine", action="store_true", help="also run Step 3b (correct_low_res_mask)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--narrow", action="store_true", help="debug: narrow-width UNet (NOT the benchmark config)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run each window's analysis after its own feature pass instead of concurrently with the next ones'")
    ap.add_argument("--lanes", type=int, default=None,
                    help="feature passes in flight at once, each on its own HIP stream (pipeline.WindowPipeline / parallel.ShardedPipeline). "
                         "Default: 2 for --config sd (same launches, same masks, ~2 %% more throughput: a second window's kernels take the CUs a "
                         "launch leaves idle -- 28 samples fill 7/8 of a round of tiles), 1 for --config svd (its launches fill the chip; two "
                         "lanes measured 7 %% SLOWER there, profiles/r05_b).  With more than one lane the per-launch HIP events of the `roofline` "
                         "object are recorded in a SINGLE-LANE pass over the same windows right after the timed region (`single_lane`), where a "
                         "launch has the chip to itself")
    ap.add_argument("--vae", action="store_true", help="also time the first-stage encode of one window (reported beside the metric)")
    ap.add_argument("--config", default="sd", choices=["sd", "svd"],
                    help="sd = BASELINE configs[1] (headline); svd = configs[2]: SVD 14x576x1024, t_start 17, is_refine_mask")
    ap.add_argument("--fp8-attn", action="store_true",
                    help="BASELINE configs[4]: e4m3 attention (q, k, v, P) on the long spatial self-attentions; reports the masks' agreement "
                         "with the 16-bit path of the same build")
    ap.add_argument("--masks-only", action="store_true",
                    help="opt-in pruning, NOT the reference's schedule and not the headline: the last step runs on the conditional half only and "
                         "stops after decoder block 8 (pipeline.feature_pass(masks_only=True)); taps equal up to fp32 summation order")
    ap.add_argument("--masks", type=int, default=None, help="number of masks K (default 20; configs[4] uses 50)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the chained-window figure, the in-run PMC passes and the SVD secondary")
    ap.add_argument("--no-step4", action="store_true", help="--config svd: skip the Step-4 pair timed after the headline (kernel-trace runs of the window alone)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)      # one window under rocprofv3 --pmc (see pmc_traffic)
    ap.add_argument("--precision", default=None, choices=["parity", "exact", "fp16"],
                    help="parity (default with the fp16 build of the library, the headline; the bf16 build has no exact mode and defaults to fp16): "
                         "the cheapest mode whose masks are the reference's on >= 14 of the 16 fixture windows "
                         "(IoU >= 0.99) = UNetModel.set_precision('exact') -- fp32 activations, every conv / linear on the MFMA kernels over split "
                         "(hi, lo) operands, 3x the MFMA work -- with the dead work of the last step pruned (masks_only).  exact: the same precision, "
                         "every step in full (reported as `full_schedule` by the default run).  fp16: 16-bit activations, the reference's "
                         "CUDA-autocast dtype; 2.5-3x faster, masks at mean IoU 0.90 (reported as `fast_mode` by the default run)")
    ap.add_argument("--one-window", action="store_true", help="every step runs window 0 (the pre-round-3 behaviour) instead of cycling the fixture windows")
    ap.add_argument("--inversion", action="store_true",
                    help="the `--inversion_type inversion` variant of the drivers (sd_pipeline_vspw.py:233-236, 340-345): sampler.inversion "
                         "(24 network evaluations) then the feature pass from t_start = 0 (25 more, dumps at every step) -- 49 CFG evaluations per "
                         "window instead of 3; not the headline")
    ap.add_argument("--launch-dry-run", action="store_true",
                    help="start the N ranks, form the process group, all-reduce a one per rank and print {n_gpus, rccl_ranks} -- no GPU "
                         "work (CPU test of the launcher: VIDSEG_DIST_BACKEND=gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` called directly: become the launcher -- one rank per GPU through torch.distributed.run, exactly
        # the command the driver's contract names; rank 0 of the children prints the JSON line on the inherited stdout.
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("VIDSEG_DIST_BACKEND", "nccl")       # "gloo" + VIDSEG_ONE_GPU=1: dry-run of the N > 1 path on a 1-GPU box
    if os.environ.get("VIDSEG_ONE_GPU") == "1":
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree")
    if args.launch_dry_run:
        return launch_dry_run(rank, world, backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if os.environ.get("VIDSEG_ONE_GPU") != "1" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} HIP device(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
        ones = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones)                                        # every rank really is in the group: the sum is the group's size
        rccl_ranks = int(ones.item())
        assert rccl_ranks == dist.get_world_size() == args.gpus, (rccl_ranks, dist.get_world_size(), args.gpus)

    from vidseg_diffusion_amd import ops
    if args.precision is None:
        args.precision = "parity" if ops.act_dtype() == torch.float16 and not (args.fp8_attn or args.inversion) else "fp16"
    args.parity = args.precision == "parity"
    args.exact = args.precision in ("parity", "exact")
    if args.parity:
        args.masks_only = True
    svd = args.config == "svd"
    k_masks = args.masks or 20
    refine = True if svd else args.refine
    ops.set_attention_fp8(args.fp8_attn)
    if args.exact and ops.act_dtype() != torch.float16:
        raise SystemExit("bench.py: --precision parity / exact needs the fp16 build of libvidseg_hip.so; use --precision fp16 with the bf16 build")
    if args.exact and (args.fp8_attn or args.inversion):
        raise SystemExit("bench.py: --fp8-attn / --inversion are variants of the 16-bit mode: add --precision fp16")
    stage(f"headline: config {args.config}, precision {args.precision}, masks_only {args.masks_only}, {args.warmup} + {args.steps} steps")
    out, sd_cpu, cfg, eng, labels, run_steps, timed = run_config(args, svd, rank, world, dev, args.steps, args.warmup)
    stage("headline done")

    sec_n = None
    if world > 1 and not svd and not args.no_secondary and not args.narrow and args.parity and not args.no_overlap:
        # N > 1: BASELINE configs[3] (SVD, 14 frames per GPU, 14 * N frames in all) beside the SD weak-scaling value -- the config the
        # north star's ">= 6x at 8 GPUs" is quoted on.  EVERY rank takes part (the all-gathers of parallel.resolve_windows); one
        # warm-up + two timed steps, one 14-frame window per rank, parity mode, same barrier / max-over-ranks timing as the headline.
        del eng, sd_cpu, run_steps
        torch.cuda.empty_cache()
        stage(f"secondary: SVD configs[3] over {world} ranks")
        try:
            a3 = argparse.Namespace(**vars(args))
            a3.inversion, a3.fp8_attn = False, False
            sec_n, _sd3, _cfg3, _eng3, lab3, _rs3, _t3 = run_config(a3, True, rank, world, dev, steps=2, warmup=1, secondary=True)
            if sec_n is not None:
                sec_n["labels_last_step"] = np.asarray(lab3)
        except Exception as e:
            if out is not None:
                out["secondary"] = {"error": repr(e)[:300]}
            sec_n = None
        eng = sd_cpu = run_steps = None

    if out is not None and sec_n is not None:
        lab3 = sec_n.pop("labels_last_step")
        out["secondary"] = {k: sec_n[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "unique_labels",
                                                  "n_gpus", "scaling") if k in sec_n}
        out["secondary"]["rccl_ranks"] = rccl_ranks
        out["secondary"]["roofline"] = {k: sec_n["roofline"][k] for k in ("achieved", "peak", "frac", "frac_algorithmic", "avg_launch_us", "kernel",
                                                                          "family") if k in sec_n["roofline"]}
        out["secondary"]["config"]["workload"] = out["secondary"]["config"]["workload"].replace("BASELINE configs[2]", f"BASELINE configs[3] ({F_WIN * world} frames, 14 per GPU)")
        wins3 = sec_n["config"]["windows_cycled"]
        m3 = timed_masks_vs_reference([(wins3[((1 + 2 - 1) * world) % len(wins3)], lab3[0], None)], True, k_masks, svd=True) if lab3.ndim == 3 else None
        if m3 is not None:
            out["secondary"]["mask_iou_vs_reference"] = {k: m3[k] for k in ("mean_iou", "min_iou", "windows_at_0.99", "n_windows", "windows", "case")}
            out["secondary"]["mask_iou_vs_reference"]["note"] = ("rank 0's window of the last step (its own K-means); the other ranks' windows are chained to it by 4-NN "
                                                                 "propagation like windows > 0 of a clip, the fixtures hold each window's own K-means")

    if out is not None:
        out["rccl_ranks"] = rccl_ranks
        out["dist_backend"] = (backend + (" (RCCL)" if backend == "nccl" else "")) if world > 1 else None
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from tools_metrics import matched_iou
        if svd and not args.narrow and world == 1:
            m = timed_masks_vs_reference(timed, refine, k_masks, svd=True)
            if m is not None:
                out["mask_iou_vs_reference"] = m
        if not svd and not args.narrow:
            if world == 1:
                m = timed_masks_vs_reference(timed, refine, k_masks)
            else:                                                    # rank 0 holds the last step's label stack [world, F, N]
                wins = out["config"]["windows_cycled"]
                last = (args.warmup + args.steps - 1)
                # only rank 0's window is scored: the windows of ranks > 0 are CHAINED to it (4-NN label propagation, like windows > 0 of
                # a clip), while every fixture holds its window's own K-means
                m = timed_masks_vs_reference([(wins[(last * world) % len(wins)], np.asarray(labels)[0], None)] if not args.no_overlap else [],
                                             refine, k_masks)
                if m is not None:
                    m["note"] += ("; N > 1: rank 0's window of the last step only -- the other ranks' windows are chained to it by 4-NN "
                                  "propagation, the fixtures hold each window's own K-means")
            if m is not None:
                out["mask_iou_vs_reference"] = m
        if world == 1 and ((args.masks_only and not args.parity) or args.fp8_attn):   # outside the timed region: the same window on the plain path
            saved = (args.masks_only, args.fp8_attn)
            args.masks_only = False
            ops.set_attention_fp8(False)
            from vidseg_diffusion_amd import feature_extraction as FE
            from vidseg_diffusion_amd import parallel
            FE.FeatureStore.clear()
            lat, c, uc, noise = make_inputs(dev, timed[-1][0] if timed else 0, cfg, svd=svd, lat_hw=(72, 128) if svd else (LAT, LAT))
            ref_labels = parallel.segment_windows_sharded(eng, lat, c, uc, noise=noise, num_masks=k_masks, num_steps=NUM_STEPS,
                                                          t_start=17 if svd else 22, is_aggre_attn=True, is_refine_mask=refine, seed=17,
                                                          rank=0, world=1)
            args.masks_only = saved[0]
            ops.set_attention_fp8(saved[1])
            iou, exact = matched_iou(np.asarray(labels).reshape(-1), np.asarray(ref_labels).reshape(-1), k_masks)
            key = "masks_vs_full_schedule" if saved[0] else "fp8_vs_16bit_masks"
            out[key] = {"iou": round(float(iou), 4), "identical_fraction": round(float(exact), 4)}
        plain = world == 1 and not args.no_secondary and not args.narrow and (not args.masks_only or args.parity) and not args.fp8_attn
        if plain and not args.no_overlap:                            # windows chained like one long clip (outside the headline timing)
            stage("chained windows")
            n = max(3, min(args.steps, 8))
            run_steps(2, chain=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_steps(n, chain=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out["chained_window"] = {"value": round(F_WIN * n / dt, 3), "unit": "frames/s", "ms_per_step": round(1e3 * dt / n, 3), "steps": n,
                                     "note": "the same windows chained as one clip (sd_pipeline_vspw.py:381-401): window 0 of the chain runs "
                                             "K-means, every later one the 14336 x 14336 x 640 float64 4-NN against its predecessor "
                                             "(feature_extraction.py:603-613)"}
        if plain and not args.no_overlap and args.lanes == 1 and not svd:   # two feature passes in flight (outside the headline timing)
            stage("two lanes")
            n = max(4, min(args.steps, 10))
            run_steps(3, nl=2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_steps(n, nl=2)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out["two_lanes"] = {"value": round(F_WIN * n / dt, 3), "unit": "frames/s", "ms_per_step": round(1e3 * dt / n, 3), "steps": n,
                                "note": "pipeline.WindowPipeline(lanes=2): the feature passes of two consecutive windows in flight on their own "
                                        "HIP streams (same launches, same masks: tests/test_gpu_unet.py::test_overlapped_clip_equals_sequential); "
                                        "a second window's kernels take the CUs a launch leaves idle (28 = 4*7 samples fill 7/8 of a round of "
                                        "256 CUs with power-of-two tiles)"}
        if plain and not svd and ops.act_dtype() == torch.float16 and os.environ.get("VIDSEG_BENCH_MODES", "1") != "0":
            # the other precision / schedule modes on the same windows, same pipeline, outside the headline timing
            net = eng.model.diffusion_model
            prec0, mo0 = net.precision, run_steps.fkw["masks_only"]

            def time_mode(prec, mo, note):
                stage(f"mode precision={prec} masks_only={mo}")
                try:
                    net.set_precision(prec)
                    run_steps.fkw["masks_only"] = mo
                    run_steps(1)
                    torch.cuda.synchronize()
                    del run_steps.record[:]
                    n = len(run_steps.windows)
                    t0 = time.perf_counter()
                    run_steps(n)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    mm = timed_masks_vs_reference(list(run_steps.record), refine, k_masks)
                    return {"value": round(F_WIN * n / dt, 3), "unit": "frames/s", "ms_per_step": round(1e3 * dt / n, 3), "steps": n,
                            "precision": prec, "masks_only": mo,
                            "mask_iou_vs_reference": {k: mm[k] for k in ("mean_iou", "median_iou", "min_iou", "windows_at_0.99", "n_windows",
                                                                         "rate_95_interval", "mean_restarts_in_place", "mean_identical_fraction")} if mm else None,
                            "windows": [{k: x[k] for k in ("window", "iou", "restarts_in_place") if k in x} for x in mm["windows"]] if mm else None,
                            "note": note}
                except Exception as e:
                    return {"error": repr(e)[:300]}
                finally:
                    net.set_precision(prec0)
                    run_steps.fkw["masks_only"] = mo0

            x_note = ("UNetModel.set_precision('exact'): every activation in fp32, every conv / linear ONE call of the 16-bit MFMA kernels over a "
                      "three-fold K axis ([a_hi|a_lo|a_hi] x [w_hi|w_hi|w_lo], fp32 accumulation), fp32 GroupNorm / LayerNorm / GEGLU / attention "
                      "(csrc/exact_ops.hip); taps 4e-5 from the fp32 reference, so K-means++ draws the reference's seeds.  ")
            f_note = ("16-bit activations (the reference's CUDA-autocast dtype), every step in full: taps 1.2e-3 from the fp32 reference, which "
                      "re-rolls about half of the ten K-means++ restarts (DESIGN.md, mask parity) -- fast, NOT at mask parity")
            if args.parity:
                out["full_schedule"] = time_mode("exact", False, x_note + "Every step in full (the reference's schedule, nothing pruned)")
                out["fast_mode"] = time_mode("fp16", False, f_note)
            elif args.precision == "exact":
                out["fast_mode"] = time_mode("fp16", False, f_note)
            else:
                out["exact_mode"] = time_mode("exact", False, x_note + "Every step in full")
                out["parity_mode"] = time_mode("exact", True, x_note + "Last step pruned to the conditional half / decoder blocks <= 8 (masks_only)")
        if plain and not svd and args.exact and os.environ.get("VIDSEG_BENCH_STEP45", "1") != "0":
            stage("Steps 4-5 of one SD window (2K modulated passes + decodes + segmentation map)")
            out["step45"] = step45_sd(eng, cfg, dev, k_masks)
        if plain and not svd:
            stage("post-UNet kernels alone")
            try:
                out["roofline_post_unet"] = post_unet_roofline(dev)
            except Exception as e:                                   # never lose the headline line to a side measurement
                out["roofline_post_unet"] = {"error": repr(e)[:300]}
        if args.vae:                                                 # outside the timed region, never part of `value`
            out["first_stage"] = first_stage_timing(dev, svd)
        if plain and not svd:
            stage("PMC traffic passes (two rocprofv3 children)")
            tr = pmc_traffic((["--refine"] if args.refine else []) + ["--precision", args.precision]) \
                if os.environ.get("VIDSEG_BENCH_PMC", "1") != "0" else None
            dom = out["roofline"]["kernel"]
            key = dom.split(" (")[0]                                  # the kind's name starts with the kernel's own (k_gemm_p7x<5, false>, ...)
            if tr and key not in tr:                                 # kinds that cover several template instances: k_gemm_p7 -> k_gemm_p7<5>
                base = "k_gemm_ph<5" if key.startswith("k_gemm_ph") else ("k_gemm_dma<2" if key.startswith("k_gemm_dma") else key)
                key = next((k for k in sorted(tr) if k.startswith(base) and (base != "k_gemm_p7" or not k.startswith("k_gemm_p7x"))), key)
            if tr and key in tr:
                out["roofline"]["traffic"] = int(tr[key][1])
                out["roofline"]["traffic_how"] = (f"this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) over one window of "
                                                  f"the same workload, {key}: {tr[key][0]} launches, bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 / launches")
                out["roofline"]["traffic_by_kernel"] = {k: {"launches": v[0], "bytes_per_launch": int(v[1]), "read_bytes_per_launch": int(v[2]),
                                                            "write_bytes_per_launch": int(v[3])} for k, v in sorted(tr.items())}
            else:
                static = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
                if static:
                    with open(static[-1]) as fh:
                        tj = json.load(fh)
                    rec = tj.get("per_kernel", {}).get("void " + key) or tj.get("per_kernel", {}).get(key)
                    if rec:
                        out["roofline"]["traffic_static"] = {"bytes_per_launch": rec["traffic_bytes_per_launch"],
                                                             "source": os.path.relpath(static[-1], ROOT) + " (an earlier PMC pass, not this run)"}
        if not args.no_cpu_baseline and not args.narrow and not svd and world == 1:     # host-side leg: rank 0 at N = 1 only
            stage("cpu baseline (oracle on the host)")
            out["cpu_baseline"] = cpu_baseline(sd_cpu, cfg, k_masks)
            out["cpu_baseline"]["reference_style_dump_io"] = dump_io_cost()
        if plain and not svd:                                        # BASELINE configs[2] as a secondary record
            del eng, sd_cpu, run_steps
            torch.cuda.empty_cache()
            stage("secondary: SVD configs[2]")
            try:
                args.inversion = False
                # 1 warm-up + 8 timed steps: the timed steps cycle over fixture windows 1..8 -- window 6, the one the parity mode is known
                # to miss (tests/test_gpu_c3_window.py), is among the windows `mask_iou_vs_reference` scores
                sec, _sd2, _cfg2, eng2, _lab2, rs2, _t2 = run_config(args, True, rank, world, dev, steps=8, warmup=1, secondary=True)
                out["secondary"] = {k: sec[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "unique_labels",
                                                       "step4_latent_blending") if k in sec}
                out["secondary"]["roofline"] = {k: sec["roofline"][k] for k in ("achieved", "frac", "kernel", "family")}
                m2 = timed_masks_vs_reference(_t2, True, k_masks, svd=True)
                if m2 is not None:
                    out["secondary"]["mask_iou_vs_reference"] = {k: m2[k] for k in ("mean_iou", "min_iou", "windows_at_0.99", "n_windows", "rate_95_interval",
                                                                                    "mean_restarts_in_place", "mean_identical_fraction", "windows", "case")}
                if args.exact:                                       # the 16-bit mode of the same config beside it
                    stage("secondary: 16-bit mode")
                    net2 = eng2.model.diffusion_model
                    net2.set_precision("fp16")
                    rs2.fkw["masks_only"] = False
                    rs2(1)
                    torch.cuda.synchronize()
                    del rs2.record[:]
                    t0 = time.perf_counter()
                    rs2(3)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    m3 = timed_masks_vs_reference(list(rs2.record), True, k_masks, svd=True)
                    out["secondary"]["fast_mode"] = {"value": round(F_WIN * 3 / dt, 3), "unit": "frames/s", "ms_per_step": round(1e3 * dt / 3, 3),
                                                     "steps": 3, "precision": "fp16", "masks_only": False,
                                                     "mask_iou_vs_reference": {k: m3[k] for k in ("mean_iou", "min_iou", "windows_at_0.99",
                                                                                                  "n_windows")} if m3 else None,
                                                     "note": "16-bit activations, every step in full; NOT at mask parity "
                                                             "(tests/test_gpu_c3_window.py: exact mode 0.9994, 16-bit mode 0.72 on the fixture window)"}
                # BASELINE configs[4]: the same SVD window with the e4m3 attention path and K = 50 masks (16-bit mode elsewhere), same engine
                stage("secondary: configs[4] (SVD, fp8 attention, K = 50)")
                a4 = argparse.Namespace(**vars(args))
                a4.precision, a4.exact, a4.parity, a4.masks_only, a4.fp8_attn, a4.masks = "fp16", False, False, False, True, 50
                prev8 = ops.set_attention_fp8(True)
                try:
                    n2 = sum(int(p_.numel()) for p_ in eng2.model.diffusion_model.parameters())
                    sec8, *_r8 = run_config(a4, True, rank, world, dev, steps=3, warmup=1, secondary=True, prebuilt=(eng2, _cfg2, _sd2, n2))
                    out["secondary_fp8"] = {k: sec8[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "unique_labels")
                                            if k in sec8}
                    out["secondary_fp8"]["roofline"] = fp8_attention_roofline(dev)
                    out["secondary_fp8"]["gemm_family"] = {k: sec8["roofline"]["family"][k] for k in ("achieved", "frac", "gemm_ms_per_step")}
                finally:
                    ops.set_attention_fp8(prev8)
            except Exception as e:                                   # never lose the headline line to the secondary
                out.setdefault("secondary", {})["error"] = repr(e)[:300]
        emit(out)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def self_launch(n):
    """Re-exec this script as N ranks of one node: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port <free> bench.py <same flags>.  Returns the launcher's exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")                # dmabuf IPC (RCCL needs it on this host driver)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def launch_dry_run(rank, world, backend):
    """The launcher and the process group without any GPU work (tests/test_parallel_gloo.py::test_bench_self_launch)."""
    ranks = 1
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend)
        ones = torch.ones(1)
        dist.all_reduce(ones)
        ranks = int(ones.item())
        assert ranks == dist.get_world_size()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"n_gpus": world, "rccl_ranks": ranks, "dist_backend": backend, "dry_run": True}))


def first_stage_timing(dev, svd):
    from vidseg_diffusion_amd import synthetic
    from vidseg_diffusion_amd.vae import AutoencoderKL, decode_first_stage, encode_first_stage
    dd = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    vae = AutoencoderKL(embed_dim=4, ddconfig=dd)
    vshapes = {k: tuple(v.shape) for k, v in vae.state_dict().items()}
    vae.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(vshapes, seed=99).items()})
    ih, iw = (576, 1024) if svd else (512, 512)
    frames = (torch.rand(F_WIN, 3, ih, iw) * 2 - 1).to(dev)
    nz = torch.randn(F_WIN, 4, ih // 8, iw // 8)
    for _ in range(2):
        encode_first_stage(vae, frames, 0.18215, noise=nz)
    torch.cuda.synchronize()
    tv = time.perf_counter()
    for _ in range(3):
        encode_first_stage(vae, frames, 0.18215, noise=nz)
    torch.cuda.synchronize()
    enc_ms = 1e3 * (time.perf_counter() - tv) / 3
    zz = encode_first_stage(vae, frames, 0.18215, noise=nz)
    for _ in range(2):
        decode_first_stage(vae, zz, 0.18215)
    torch.cuda.synchronize()
    tv = time.perf_counter()
    for _ in range(3):
        decode_first_stage(vae, zz, 0.18215)
    torch.cuda.synchronize()
    return {"encode_ms_per_window": round(enc_ms, 2), "decode_ms_per_window": round(1e3 * (time.perf_counter() - tv) / 3, 2),
            "frames": F_WIN, "image": [ih, iw],
            "note": "AutoencoderKL.encode / .decode (SD image decoder), synthetic weights; excluded from `value`"}


if __name__ == "__main__":
    main()
