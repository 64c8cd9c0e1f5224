"""The window-sharded multi-rank path on REAL kernels: two ranks (both on cuda:0, gloo rendezvous with host-staged
collectives, since one box has one GPU) must reproduce the single-process sequential window loop of the reference
(segment_clip) bit for bit -- K-means on window 0, label chain by top-4 search + vote, refinement per window."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from vidseg_diffusion_amd import synthetic

pytestmark = pytest.mark.gpu
F, K = 3, 5


def _setup(dev, kind="sd", precision="fp16"):
    """kind "sd": the narrow SD UNet; "svd": the narrow VideoUNet -- the unit of BASELINE configs[3] (SVD windows sharded one per GPU):
    its temporal attention / convolution mix the frames of a window, which is why that config shards by WINDOW."""
    from vidseg_diffusion_amd.pipeline import build_sd_engine, build_svd_engine
    if kind == "svd":
        from vidseg_diffusion_amd.video_unet import VideoUNet
        net = VideoUNet(**synthetic.SVD_NARROW)
    else:
        from vidseg_diffusion_amd.unet import UNetModel
        net = UNetModel(**synthetic.SD21_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, 1234).items()})
    if precision != "fp16":
        net.pack(dev)
        net.set_precision(precision)
    return build_svd_engine(net, num_frames=F) if kind == "svd" else build_sd_engine(net)


def _inputs(win, dev, kind="sd"):
    lat = torch.from_numpy(synthetic.latent_clip(F, 16, 16, seed=50 + win)).to(dev)
    noise = torch.from_numpy(np.random.Generator(np.random.PCG64(90 + win)).standard_normal((F, 4, 16, 16)).astype(np.float32)).to(dev)
    if kind == "svd":                                                    # svd_pipeline_vspw.py:300-311: one CLIP-image token, frame-0 latent, fps / motion vector
        g = np.random.Generator(np.random.PCG64(7 + win))
        ctx = torch.from_numpy(g.standard_normal((1, 1, 64)).astype(np.float32)).repeat(F, 1, 1).to(dev)
        cat = lat[:1].repeat(F, 1, 1, 1) / 0.18215 * 0.2
        vec = torch.from_numpy(g.standard_normal((1, 64)).astype(np.float32)).repeat(F, 1).to(dev)
        c = {"crossattn": ctx, "concat": cat, "vector": vec}
        uc = {"crossattn": torch.zeros_like(ctx), "concat": torch.zeros_like(cat), "vector": vec.clone()}
        return lat, c, uc, noise
    c = torch.from_numpy(np.random.Generator(np.random.PCG64(7)).standard_normal((F, 7, 64)).astype(np.float32)).to(dev)
    return lat, {"crossattn": c}, {"crossattn": torch.zeros_like(c)}, noise


def _worker(rank, world, port, q, kind="sd", precision="fp16"):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    from vidseg_diffusion_amd import parallel
    eng = _setup(dev, kind, precision)
    lat, c, uc, noise = _inputs(rank, dev, kind)
    mo = precision == "exact"                                            # the parity mode of bench.py: exact + masks_only
    labels = parallel.segment_windows_sharded(eng, lat, c, uc, noise=noise, num_masks=K, is_refine_mask=True, seed=17, rank=rank,
                                              world=world, feature_folder="/nonexistent/par", exp_name=f"r{rank}", masks_only=mo)
    # the overlapped form: two steps through ShardedPipeline (step 2's feature pass queued before step 1's cross-window stage)
    fkw = dict(noise=noise, seed=17, feature_folder="/nonexistent/par", masks_only=mo)
    pipe = parallel.ShardedPipeline(eng, rank, world, num_masks=K, is_refine_mask=True)
    assert pipe.push(lat, c, uc, exp_name=f"p{rank}a", **fkw) is None
    first = pipe.push(lat, c, uc, exp_name=f"p{rank}b", **fkw)
    second = pipe.flush()
    assert np.array_equal(first, labels) and np.array_equal(second, labels), "overlapped sharded steps differ from the plain one"
    # two feature-pass lanes (two windows in flight on their own streams + scratch): same labels, delivered two pushes later
    pipe = parallel.ShardedPipeline(eng, rank, world, lanes=2, num_masks=K, is_refine_mask=True)
    got = [pipe.push(lat, c, uc, exp_name=f"l{rank}{i}", **fkw) for i in range(3)]
    assert got[0] is None and got[1] is None and np.array_equal(got[2], labels)
    rest = pipe.drain()
    assert len(rest) == 2 and all(np.array_equal(r, labels) for r in rest), "two-lane sharded steps differ from the plain one"
    q.put((rank, labels))
    dist.barrier()
    dist.destroy_process_group()


def _two_ranks_vs_sequential(kind, precision):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, kind, precision)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # sequential reference loop in this process
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import WindowState, segment_window
    dev = torch.device("cuda:0")
    eng = _setup(dev, kind, precision)
    FE.FeatureStore.clear()
    FE.MaskStore.clear()
    state, seq = WindowState(), []
    for win in range(2):
        lat, c, uc, noise = _inputs(win, dev, kind)
        labels, state = segment_window(eng, lat, c, uc, num_masks=K, is_refine_mask=True, seed=17, state=state, noise=noise,
                                       feature_folder="/nonexistent/seq", exp_name=f"w{win}", masks_only=(precision == "exact"))
        seq.append(labels)
    seq = np.stack(seq)
    for r in range(2):
        assert np.array_equal(res[r], seq), f"rank {r}: sharded labels differ from the sequential window loop ({kind}, {precision})"
    FE.FeatureStore.clear()
    FE.MaskStore.clear()


def test_two_ranks_equal_sequential_windows():
    _two_ranks_vs_sequential("sd", "fp16")


def test_two_ranks_equal_sequential_svd_windows():
    """BASELINE configs[3]'s unit: SVD windows, one per rank, through segment_windows_sharded / ShardedPipeline with the narrow
    VideoUNet (temporal attention / temporal convolutions / AlphaBlender inside every window, one-token context) in the 16-bit mode,
    one and two lanes; the gathered label chain must be the sequential window loop's bit for bit."""
    _two_ranks_vs_sequential("svd", "fp16")
    # the parity mode (exact + masks_only) of the same unit runs at FULL size in tests/test_gpu_c3_window.py::test_two_ranks_full_size_svd_windows


def _rccl_worker(port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)           # "nccl" IS RCCL on ROCm
    from vidseg_diffusion_amd import parallel
    from vidseg_diffusion_amd.pipeline import segment_window
    eng = _setup(dev)
    lat, c, uc, noise = _inputs(0, dev)
    ref, _ = segment_window(eng, lat, c, uc, num_masks=K, is_refine_mask=True, seed=17, noise=noise, feature_folder="/nonexistent/rc",
                            exp_name="ref")
    # the sharded path proper with a group of one: every collective of resolve_windows goes through RCCL on device buffers
    h = parallel.sharded_feature_pass(eng, lat, c, uc, noise=noise, seed=17, rank=0, feature_folder="/nonexistent/rc", exp_name="r0")
    got = parallel.sharded_resolve(eng, h, num_masks=K, is_refine_mask=True, rank=0, world=1)
    assert got.shape[0] == 1 and np.array_equal(got[0], ref), "sharded path over RCCL (world 1) differs from segment_window"
    os.environ["VIDSEG_CHECK_RANKS"] = "1"                                          # + the int64 checksum all-gather
    pipe = parallel.ShardedPipeline(eng, 0, 1, num_masks=K, is_refine_mask=True)    # collectives on the side stream
    assert pipe.push(lat, c, uc, noise=noise, seed=17, feature_folder="/nonexistent/rc", exp_name="p0a") is None
    first = pipe.push(lat, c, uc, noise=noise, seed=17, feature_folder="/nonexistent/rc", exp_name="p0b")
    second = pipe.flush()
    assert np.array_equal(first[0], ref) and np.array_equal(second[0], ref)
    t = torch.tensor([1.5], device=dev, dtype=torch.float64)                        # bench.py's max-over-ranks of the step time
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize()
    q.put(float(t.item()))
    dist.destroy_process_group()


def test_sharded_path_over_rccl_with_one_rank():
    """One box has one GPU, so RCCL cannot run two ranks here -- but a group of ONE over backend "nccl" still sends every collective of
    the sharded path (fp16 feature all-gather, int32 index / track all-gathers, the int64 checksum, barrier, float64 max all-reduce)
    through RCCL on device buffers and on the side stream: API, dtype and stream-semantics errors of the N > 1 path show up here."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(port, q))
    p.start()
    assert q.get(timeout=600) == 1.5
    p.join(timeout=120)
    assert p.exitcode == 0


def test_bench_gpus_flag_starts_the_ranks():
    """`python bench.py --gpus 2` called directly (the driver's call) on a one-GPU box: both ranks on cuda:0, gloo collectives.
    The line must say n_gpus = 2 = the size of the group that was formed, and count two windows per step."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(VIDSEG_DIST_BACKEND="gloo", VIDSEG_ONE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--narrow", "--steps", "2", "--warmup", "1",
                        "--no-secondary", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = lines[0]
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["scaling"] == "weak"
    assert abs(out["value"] - 2 * 14 * 2 / (out["ms_per_step"] * 2 / 1e3)) / out["value"] < 1e-2      # frames of BOTH ranks / max-rank time
    # configs[3]'s unit (SVD windows, one per rank) through the same launcher: covered at FULL size by
    # tests/test_gpu_c3_window.py::test_two_ranks_full_size_svd_windows and, for the launcher's flags, by tests/test_parallel_gloo.py::test_bench_self_launch


def _frames_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    from vidseg_diffusion_amd import parallel
    eng = _setup(dev)
    lat, c, uc, noise = _inputs(0, dev)                                  # every rank holds the WINDOW's inputs, runs its frame slice
    kw = dict(noise=noise, seed=17, feature_folder="/nonexistent/fr")
    h = parallel.frame_sharded_feature_pass(eng, lat, c, uc, rank=rank, world=world, exp_name=f"t{rank}", **kw)
    taps = parallel.frame_sharded_resolve(eng, h, num_masks=K, is_refine_mask=True, rank=rank, world=world,
                                          analysis=lambda t, F_, fh, fw, seed: {b: v.cpu().numpy() for b, v in t.items()})
    labels = parallel.segment_window_frame_sharded(eng, lat, c, uc, rank=rank, world=world, num_masks=K, is_refine_mask=True,
                                                   exp_name=f"f{rank}", **kw)
    q.put((rank, labels, taps))
    dist.barrier()
    dist.destroy_process_group()


def test_sd_window_sharded_by_frames():
    """SURVEY 8(e), SD: the frames of ONE window over two ranks (2 + 1 frames, both on cuda:0, gloo): every rank ends with the same
    gathered tap stacks and the same labels; the labels are the oracle's analysis of those taps bit for bit; the taps are the
    one-rank pass's up to the fp32 summation order of another batch size (GEMM tiling / split-K follow M)."""
    from oracle import analysis as OA
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_frames_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (lab, taps) for r, lab, taps in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(res[0][0], res[1][0])
    for b in (6, 7, 8):
        assert np.array_equal(res[0][1][b], res[1][1][b])
    taps = res[0][1]
    fh = fw = 8
    pad = [np.concatenate([np.zeros_like(taps[b]), taps[b]], 0) for b in (8, 7, 6)]
    np.random.seed(17)
    _, lab, _ = OA.match_gt_mask(OA.aggregate_blocks(pad), K, np.random.mtrand._rand)
    th, tw = OA.dense_tracking(pad[1], F, fh, fw)
    corr, _ = OA.correct_low_res_mask(lab.reshape(F, fh, fw), th, tw)
    assert np.array_equal(res[0][0].reshape(-1), corr.reshape(-1)), "frame-sharded labels differ from the oracle's analysis of the gathered taps"
    # the one-rank pass of the same window
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import segment_window
    dev = torch.device("cuda:0")
    eng = _setup(dev)
    FE.FeatureStore.clear()
    lat, c, uc, noise = _inputs(0, dev)
    segment_window(eng, lat, c, uc, num_masks=K, is_refine_mask=True, seed=17, noise=noise, feature_folder="/nonexistent/one", exp_name="w")
    st = FE.FeatureStore.folder("/nonexistent/one", "w")
    for b in (6, 7, 8):
        one = st[f"output_block_{b}_spatial_self_attn_q_time_24"][F:].float().cpu().numpy()
        e = float(np.linalg.norm(taps[b].astype(np.float32) - one) / np.linalg.norm(one))
        print(f"block {b}: frame-sharded taps vs the one-rank pass: nrms {e:.2e}")
        assert e <= 2e-3, (b, e)
    FE.FeatureStore.clear()
