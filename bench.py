#!/usr/bin/env python
"""Headline benchmark: segmented frames/s of the VidSeg hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one 14-frame 512x512 clip window per GPU (BASELINE config 2:
SD 2.1, 20 masks, is_aggre_attn on): add-noise -> 3 Euler steps (i = 22,23,24) of the full-size SD 2.1 UNet with
classifier-free guidance (batch 2x14) and the reference's Q/K taps on every decoder transformer block -> 3-block
aggregation -> K-means (n_init 10) -> 4-NN label propagation  => cluster-id masks [14, 32*32].
Latents / conditioning / weights are synthetic, seeded and already resident in HBM when the timed region
starts (VAE + conditioner excluded).  N > 1: one window per GPU (weak scaling), see
vidseg_diffusion_amd/parallel.py for the exchange.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      the single kernel with the most time in the timed region (normally k_gemm_ph<NJ>, the phased 256x320 /
                256x256 LDS-DMA implicit-GEMM tile): achieved = algorithmic FLOPs (2*M*N*K per launch) / HIP-event time of its
                launches, recorded on the launch stream inside the timed region; peak = 2500 TFLOP/s dense bf16
                (MI355X_MICROARCH.md); traffic = PMC bytes per launch of that kernel (profiles/r01_traffic.json).
                `family` = the same figures over ALL conv/linear launches (three kernels share the work).
  cpu_baseline  the oracle ("port") timed on this box's host cores on a bounded sample (see `sample`).
  mask_iou_vs_oracle  the metric's second half: HIP masks vs the all-fp32 oracle's on BASELINE configs[0] at full UNet width
                (the case the CPU finishes in seconds), IoU up to a label permutation.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_WIN, LAT, K_MASKS, T_START, NUM_STEPS = 14, 64, 20, 22, 25


def make_inputs(dev, window_id, cfg, svd=False, lat_hw=(LAT, LAT)):
    from vidseg_diffusion_amd import synthetic
    lat = torch.from_numpy(synthetic.latent_clip(F_WIN, lat_hw[0], lat_hw[1], seed=1 + window_id)).to(dev)
    g = torch.Generator().manual_seed(100 + window_id)
    noise = torch.randn(lat.shape, generator=g).to(dev)
    if not svd:
        c, uc = synthetic.sd_conditioning(F_WIN, context_dim=cfg["context_dim"], seq=77, seed=1)
        return lat, {"crossattn": torch.from_numpy(c).to(dev)}, {"crossattn": torch.from_numpy(uc).to(dev)}, noise
    # SVD conditioning (svd_pipeline_vspw.py:300-311): CLIP-image token and frame-0 latent repeated over frames, fps/motion vector
    ctx = torch.randn((1, 1, cfg["context_dim"]), generator=g).repeat(F_WIN, 1, 1).to(dev)
    cat = lat[:1].repeat(F_WIN, 1, 1, 1) / 0.18215 * 0.2
    vec = torch.randn((1, cfg["adm_in_channels"]), generator=g).repeat(F_WIN, 1).to(dev)
    c = {"crossattn": ctx, "concat": cat, "vector": vec}
    uc = {"crossattn": torch.zeros_like(ctx), "concat": torch.zeros_like(cat), "vector": vec.clone()}
    return lat, c, uc, noise


def cpu_baseline(sd_cpu, cfg):
    """Oracle on the host: one full-size UNet evaluation for ONE frame (CFG pair, batch 2) + the full-size analysis
    stage on synthetic dumps; scaled to a 14-frame window: t = 3 steps * 14 frames * t_unet_frame + t_analysis."""
    from oracle import analysis as OA
    from oracle.unet import UNetOracle
    from vidseg_diffusion_amd import synthetic
    torch.set_grad_enabled(False)
    cores = torch.get_num_threads()
    o = UNetOracle(sd_cpu)
    x = torch.from_numpy(synthetic.latent_clip(1, LAT, LAT, seed=1)).repeat(2, 1, 1, 1)
    c, uc = synthetic.sd_conditioning(1, context_dim=cfg["context_dim"])
    ctx = torch.cat([torch.from_numpy(uc), torch.from_numpy(c)])
    t0 = time.time()
    o.forward(x, torch.tensor([958.0, 958.0]), ctx)
    t_unet = time.time() - t0
    blocks, _ = synthetic.attention_q_dumps(F_WIN, LAT // 2, LAT // 2, 640, num_blocks=3, seed=1)
    t0 = time.time()
    np.random.seed(17)
    OA.match_gt_mask(OA.aggregate_blocks(blocks), K_MASKS, np.random.mtrand._rand)
    t_an = time.time() - t0
    t_window = 3 * F_WIN * t_unet + t_an
    return {"value": round(F_WIN / t_window, 5), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 on host: 1 full-size UNet eval of 1 frame (CFG batch 2, {t_unet:.1f}s) + full 14x32x32x640 K=20 "
                      f"analysis ({t_an:.1f}s); window time = 3*14*t_unet + t_analysis = {t_window:.0f}s"}


def mask_iou_check(eng, sd_cpu, cfg, dev):
    """The metric's second half, on a case the CPU can finish in seconds: BASELINE configs[0] (4 frames at 256x256, K = 5, one
    step) at FULL UNet width -- HIP pipeline vs the all-fp32 oracle, IoU up to a label permutation (tests/tools_metrics.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import pipeline as OP
    from oracle.unet import UNetOracle
    from tools_metrics import matched_iou
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd import synthetic
    from vidseg_diffusion_amd.pipeline import segment_window
    Fn, K, T0 = 4, 5, 24
    lat = synthetic.latent_clip(Fn, 32, 32, seed=1)
    c, ucn = synthetic.sd_conditioning(Fn, context_dim=cfg["context_dim"], seq=77, seed=1)
    noise = torch.from_numpy(np.random.Generator(np.random.PCG64(9)).standard_normal(lat.shape).astype(np.float32))
    t0 = time.time()
    ref = OP.segment_window(UNetOracle(sd_cpu), torch.from_numpy(lat), torch.from_numpy(c), torch.from_numpy(ucn), noise, num_masks=K,
                            t_start=T0, seed=17)
    t_cpu = time.time() - t0
    FE.FeatureStore.clear()
    FE.MaskStore.clear()
    labels, _ = segment_window(eng, torch.from_numpy(lat).to(dev), {"crossattn": torch.from_numpy(c).to(dev)},
                               {"crossattn": torch.from_numpy(ucn).to(dev)}, num_masks=K, t_start=T0, seed=17, noise=noise.to(dev),
                               feature_folder="/nonexistent/bench_iou", exp_name="c1")
    FE.FeatureStore.clear()
    FE.MaskStore.clear()
    iou, exact = matched_iou(labels, ref["labels"], K)
    return {"iou": round(float(iou), 4), "identical_fraction": round(float(exact), 4),
            "case": f"BASELINE configs[0] at full width: 4x256x256, K=5, 1 step; fp32 oracle took {t_cpu:.1f}s on the host"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--refine", action="store_true", help="also run Step 3b (correct_low_res_mask)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--narrow", action="store_true", help="debug: narrow-width UNet (NOT the benchmark config)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run each window's analysis after its own feature pass instead of concurrently with the next one's")
    ap.add_argument("--overlap-multi", action="store_true",
                    help="N>1: also overlap (parallel.ShardedPipeline: collectives on a second stream). Verified with 2 gloo ranks on one "
                         "GPU; off by default until it has run over RCCL on a multi-GPU node")
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
    args = ap.parse_args()
    global K_MASKS
    if args.masks:
        K_MASKS = args.masks

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("VIDSEG_DIST_BACKEND", "nccl")       # "gloo" + VIDSEG_ONE_GPU=1: dry-run of the N > 1 path on a 1-GPU box
    if os.environ.get("VIDSEG_ONE_GPU") == "1":
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd import ops, parallel, synthetic
    from vidseg_diffusion_amd.pipeline import build_sd_engine
    from vidseg_diffusion_amd.unet import UNetModel

    svd = args.config == "svd"
    ops.set_attention_fp8(args.fp8_attn)
    global T_START
    if svd:
        from vidseg_diffusion_amd.video_unet import VideoUNet
        cfg = dict(synthetic.SVD_NARROW if args.narrow else synthetic.SVD_FULL)
        net = VideoUNet(**cfg)
        T_START = 17
        args.refine = True
    else:
        cfg = dict(synthetic.SD21_NARROW if args.narrow else synthetic.SD21_FULL)
        net = UNetModel(**cfg)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd_cpu = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=1234).items()}
    net.load_state_dict(sd_cpu)
    net.pack(dev)
    if svd:
        from vidseg_diffusion_amd.pipeline import build_svd_engine
        eng = build_svd_engine(net, num_frames=F_WIN, num_steps=NUM_STEPS)
        lat, c, uc, noise = make_inputs(dev, rank, cfg, svd=True, lat_hw=(72, 128))
    else:
        eng = build_sd_engine(net, num_steps=NUM_STEPS, scale=5.0)
        lat, c, uc, noise = make_inputs(dev, rank, cfg)
    torch.cuda.synchronize()

    def one_step():
        FE.FeatureStore.clear()
        FE.MaskStore.clear()
        return parallel.segment_windows_sharded(eng, lat, c, uc, noise=noise, num_masks=K_MASKS, num_steps=NUM_STEPS, t_start=T_START,
                                                is_aggre_attn=True, is_refine_mask=args.refine, seed=17, rank=rank, world=world,
                                                masks_only=args.masks_only)

    # N = 1: windows run through pipeline.WindowPipeline -- the analysis of step i (second HIP stream) overlaps the feature pass
    # of step i+1; every step is still a complete window (K-means included) and all of them finish inside the timed region.
    overlap = world == 1 and not args.no_overlap
    if overlap:
        from vidseg_diffusion_amd.pipeline import WindowPipeline
        pipe = WindowPipeline(eng, chain=False, num_masks=K_MASKS, is_aggre_attn=True, is_refine_mask=args.refine)
        step_no = [0]

        def run_steps(n):
            last = None
            for _ in range(n):
                step_no[0] += 1
                FE.MaskStore.clear()
                got = pipe.push(lat, c, uc, num_steps=NUM_STEPS, t_start=T_START, seed=17, noise=noise, keep_all_steps=False,
                                masks_only=args.masks_only,
                                exp_name=f"step{step_no[0] % 2}")
                last = got if got is not None else last
            got = pipe.flush()
            return got if got is not None else last
    elif world > 1 and args.overlap_multi and not args.no_overlap:
        spipe = parallel.ShardedPipeline(eng, rank, world, num_masks=K_MASKS, is_aggre_attn=True, is_refine_mask=args.refine)
        step_no = [0]

        def run_steps(n):
            last = None
            for _ in range(n):
                step_no[0] += 1
                FE.MaskStore.clear()
                got = spipe.push(lat, c, uc, noise=noise, num_steps=NUM_STEPS, t_start=T_START, seed=17, exp_name=f"r{rank}s{step_no[0] % 2}",
                                 masks_only=args.masks_only)
                last = got if got is not None else last
            got = spipe.flush()
            return got if got is not None else last
        overlap = True
    else:
        def run_steps(n):
            last = None
            for _ in range(n):
                last = one_step()
            return last

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup:
        run_steps(args.warmup)
    barrier()
    ops.gemm_profile_begin()
    t0 = time.perf_counter()
    labels = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    k_ms, k_flops, k_launches = ops.gemm_profile_end()
    kinds = ops.gemm_profile_kinds()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")           # PMC pass (separate rocprofv3 --pmc runs), see file
    if os.path.exists(tpath) and not svd and not args.narrow:
        with open(tpath) as fh:
            tj = json.load(fh)
            traffic = tj.get("traffic_bytes_per_launch")
    if rank == 0:
        frames = F_WIN * world * args.steps
        fam_tf = (k_flops / (k_ms * 1e-3)) / 1e12 if k_ms > 0 else 0.0
        dom = max(kinds, key=lambda r: r[1])                                  # the single kernel with the most time in the region
        if traffic is not None:                                               # PMC pass: per-launch bytes of that kernel if recorded
            for kname, rec in tj.get("per_kernel", {}).items():
                if ("k_gemm_ph<5>" in kname and dom[0].startswith("k_gemm_ph")) or \
                        ("k_gemm_dma<2>" in kname and dom[0].startswith("k_gemm_dma")):
                    traffic = rec["traffic_bytes_per_launch"]
                    break
        achieved = (dom[2] / (dom[1] * 1e-3)) / 1e12 if dom[1] > 0 else 0.0
        out = {
            "metric": "segmented frames/sec (14-frame 512^2 clip, 20 masks)",
            "value": round(frames / elapsed, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if ops.act_dtype() == torch.float16 else "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: SD 2.1 full-size UNet (865.9M params, random-init), 14-frame 512x512 window per GPU "
                                   "(latent 14x4x64x64), 25-step schedule with t_start=22 (3 CFG UNet evals, batch 28), Q/K taps on decoder "
                                   "blocks 3-11, is_aggre_attn (blocks 6,7,8), K-means K=20 n_init=10 + 4-NN"
                                   + (", is_refine_mask" if args.refine else ""),
                       "frames_per_gpu": F_WIN, "num_masks": K_MASKS, "unet_evals_per_step": 3, "parallelism": f"window-per-gpu x{world}",
                       "overlap": "analysis of step i on a second HIP stream, concurrent with the feature pass of step i+1" if overlap
                                  else "none (each window's analysis follows its own feature pass)"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(achieved / 2500.0, 4), "traffic": traffic, "kernel": dom[0],
                         "launches_per_step": dom[3] // max(args.steps, 1),
                         "avg_launch_us": round(1e3 * dom[1] / max(dom[3], 1), 2),
                         "ms_per_step": round(dom[1] / args.steps, 3),
                         # every conv / linear of the UNet runs on this kernel family; the family-wide figures:
                         "family": {"achieved": round(fam_tf, 2), "frac": round(fam_tf / 2500.0, 4),
                                    "launches_per_step": k_launches // max(args.steps, 1), "gemm_ms_per_step": round(k_ms / args.steps, 3),
                                    "by_kernel": {n: {"ms_per_step": round(ms / args.steps, 3), "tflops": round(fl / ms / 1e9, 1) if ms > 0 else 0.0,
                                                      "launches_per_step": ln // max(args.steps, 1)} for (n, ms, fl, ln) in kinds if ln}}},
            "unique_labels": int(len(np.unique(labels))),
        }
        if svd:
            out["metric"] = "segmented frames/sec (14-frame 576x1024 clip, 20 masks, SVD)"
            out["config"]["workload"] = ("BASELINE configs[2]: SVD img2vid full-size VideoUNet (1524.6M params, random-init), 14-frame 576x1024 "
                                         "window per GPU (latent 14x4x72x128), t_start=17 (8 CFG UNet evals, batch 28), spatial+temporal taps, "
                                         "is_aggre_attn, K-means K=20 + 4-NN, is_refine_mask (dense tracking + vote)")
            out["config"]["unet_evals_per_step"] = 8
        if args.masks_only:                                              # never the headline: the reference's schedule runs every step in full
            out["metric"] += " [masks-only pruning: NOT the reference schedule]"
            out["config"]["workload"] += ("; OPT-IN PRUNING (--masks-only): the last UNet evaluation runs on the conditional half only and "
                                          "stops after decoder block 8 (its other outputs are never read by Steps 3-3b)")
            out["config"]["unet_evals_per_step"] = "2 full + 1 taps-only (cond half, blocks <= 8)"
        if args.masks_only and world == 1:                               # outside the timed region: the same window on the full schedule
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from tools_metrics import matched_iou
            args.masks_only = False
            ref_labels = one_step()
            args.masks_only = True
            iou, exact = matched_iou(np.asarray(labels).reshape(-1), np.asarray(ref_labels).reshape(-1), K_MASKS)
            out["masks_vs_full_schedule"] = {"iou": round(float(iou), 4), "identical_fraction": round(float(exact), 4)}
        if args.masks:
            out["metric"] = out["metric"].replace("20 masks", f"{K_MASKS} masks")
            out["config"]["workload"] = out["config"]["workload"].replace("K=20", f"K={K_MASKS}")
        if args.fp8_attn:
            out["dtype"] += " + e4m3 attention (q, k, v, P; >= 1024 keys)"
            out["config"]["workload"] += "; BASELINE configs[4] attention path: OCP e4m3 MFMA on the spatial self-attentions"
        if args.fp8_attn and world == 1:                                 # outside the timed region: the same window on the 16-bit kernels
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from tools_metrics import matched_iou
            ops.set_attention_fp8(False)
            ref_labels = one_step()
            ops.set_attention_fp8(True)
            iou, exact = matched_iou(np.asarray(labels).reshape(-1), np.asarray(ref_labels).reshape(-1), K_MASKS)
            out["fp8_vs_16bit_masks"] = {"iou": round(float(iou), 4), "identical_fraction": round(float(exact), 4)}
        if args.vae:                                                     # outside the timed region, never part of `value`
            from vidseg_diffusion_amd.vae import AutoencoderKL, encode_first_stage
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
            from vidseg_diffusion_amd.vae import decode_first_stage
            zz = encode_first_stage(vae, frames, 0.18215, noise=nz)
            for _ in range(2):
                decode_first_stage(vae, zz, 0.18215)
            torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(3):
                decode_first_stage(vae, zz, 0.18215)
            torch.cuda.synchronize()
            out["first_stage"] = {"encode_ms_per_window": round(enc_ms, 2), "decode_ms_per_window": round(1e3 * (time.perf_counter() - tv) / 3, 2),
                                  "frames": F_WIN, "image": [ih, iw],
                                  "note": "AutoencoderKL.encode / .decode (SD image decoder), synthetic weights; excluded from `value`"}
        if not args.no_cpu_baseline and not args.narrow and not svd and world == 1:     # host-side legs: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(sd_cpu, cfg)
            out["mask_iou_vs_oracle"] = mask_iou_check(eng, sd_cpu, cfg, dev)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
