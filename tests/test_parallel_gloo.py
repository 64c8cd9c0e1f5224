"""The N>1 path on CPU: world_size-2 gloo run of the window-sharded label chain (vidseg_diffusion_amd/parallel.py)
with oracle compute callbacks, compared with the reference's sequential window loop restated by the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import analysis as OA
from vidseg_diffusion_amd import synthetic

F, H, W, C, K = 3, 6, 5, 32, 4
N = H * W


def _window_inputs(win):
    blocks, _ = synthetic.attention_q_dumps(F, H, W, C, num_blocks=3, seed=40 + 100 * win)
    feat = OA.normalize_tokens(OA.aggregate_blocks(blocks)[F:]).reshape(F * N, C)
    th, tw = OA.dense_tracking(blocks[1], F, H, W)
    return feat, (th * W + tw).astype(np.int32)


def _sequential_reference(world, refine):
    ref_mask = ref_fm = None
    out = []
    for win in range(world):
        blocks, _ = synthetic.attention_q_dumps(F, H, W, C, num_blocks=3, seed=40 + 100 * win)
        np.random.seed(17)
        _, labels, fm = OA.match_gt_mask(OA.aggregate_blocks(blocks), K, np.random.mtrand._rand, ref_mask=ref_mask, ref_feature_map=ref_fm)
        if refine:
            th, tw = OA.dense_tracking(blocks[1], F, H, W)
            labels, _ = OA.correct_low_res_mask(labels.reshape(F, H, W), th, tw)
        out.append(labels)
        ref_mask, ref_fm = labels, fm
    return np.stack(out)


def _worker(rank, world, port, refine, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vidseg_diffusion_amd.parallel import ChainOps, resolve_windows
    feat, tracks = _window_inputs(rank)

    def first(feat0, seed0):
        assert seed0 == 17                                          # rank 0's seed, whatever this rank was given
        np.random.seed(seed0)
        f0 = feat0.numpy()
        centers, _, _ = OA.kmeans_fit(f0, K, np.random.mtrand._rand)
        fake = OA.kmeans_predict(f0[:N], centers)
        return torch.from_numpy(OA.knn_predict(f0[:N], fake, f0).astype(np.int32))

    ops = ChainOps(first_window_labels=first,
                   knn_top4=lambda r, qq: torch.from_numpy(OA.knn_top4(r.numpy(), qq.numpy()).astype(np.int32)),
                   vote4=lambda i, l: torch.from_numpy(OA.vote4(i.numpy(), l.numpy()).astype(np.int32)),
                   refine=(lambda t, l: torch.from_numpy(
                       OA.correct_low_res_mask(l.numpy().reshape(F, H, W), t.numpy() // W, t.numpy() % W)[0].astype(np.int32)).view(F, N))
                   if refine else None)
    labels = resolve_windows(torch.from_numpy(feat), torch.from_numpy(tracks) if refine else None, ops, rank, world, F,
                             seed=17 + 5 * rank, check=True)       # ranks > 0 hold OTHER seeds: window 0 must still use rank 0's
    q.put((rank, labels.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,refine", [(2, False), (2, True), (8, True)])
def test_chain_matches_sequential_reference(world, refine):
    """world 2 and world 8 (BASELINE configs[3]: 8 windows, one per GPU): every rank ends with the labels of ALL windows, equal
    to the reference's sequential window loop."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, refine, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _sequential_reference(world, refine)
    for r in range(world):
        assert np.array_equal(res[r], ref), f"rank {r} labels differ from the sequential window loop"


def test_window_slices_match_reference_loop():
    from vidseg_diffusion_amd.pipeline import window_slices
    # sd_pipeline_vspw.py:228-245: a 28-frame clip gives 3 windows, the last re-processing frames 14..27
    assert window_slices(28, 14) == [(0, 14), (14, 28), (14, 28)]
    assert window_slices(30, 14) == [(0, 14), (14, 28), (16, 30)]
    assert window_slices(5, 14) == [(0, 5)]


def test_bench_self_launch():
    """`python bench.py --gpus 2` called DIRECTLY (no torchrun around it, WORLD_SIZE unset -- how the driver calls it) must start two
    ranks itself and report n_gpus = 2 = the size of the process group it really formed; --gpus disagreeing with an inherited
    WORLD_SIZE must fail instead of printing a line for the wrong N."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["VIDSEG_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-dry-run"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["rccl_ranks"] == 2, r.stdout
    # the same launcher with the flags of BASELINE configs[3] (SVD windows, one per GPU)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "svd", "--steps", "2", "--warmup", "1",
                        "--launch-dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["rccl_ranks"] == 2, r.stdout
    env["WORLD_SIZE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-dry-run"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "disagree" in (r.stderr + r.stdout)


# ---------------------------------------------------------------------------------------------------------------------------------
# SD frame-level sharding (parallel.frame_slices / gather_frames / frame_sharded_resolve): the frames of ONE window over the ranks
# ---------------------------------------------------------------------------------------------------------------------------------
def test_frame_slices_cover_the_window():
    from vidseg_diffusion_amd.parallel import frame_slices
    assert frame_slices(14, 1) == [(0, 14)]
    assert frame_slices(14, 2) == [(0, 7), (7, 14)]
    assert frame_slices(14, 4) == [(0, 4), (4, 8), (8, 11), (11, 14)]
    assert frame_slices(14, 8) == [(0, 2), (2, 4), (4, 6), (6, 8), (8, 10), (10, 12), (12, 13), (13, 14)]
    for F_ in (1, 5, 14, 25):
        for w in (1, 2, 3, 8):
            sl = frame_slices(F_, w)
            assert len(sl) == w and sl[0][0] == 0 and sl[-1][1] == F_ and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
            assert max(hi - lo for lo, hi in sl) - min(hi - lo for lo, hi in sl) <= 1


def test_more_ranks_than_frames_is_refused_on_every_rank():
    """The slice table is the same on every rank, so the refusal is too: no rank may enter a collective while another raises."""
    from vidseg_diffusion_amd.parallel import _require_a_frame_per_rank, frame_slices
    _require_a_frame_per_rank(frame_slices(14, 8))
    _require_a_frame_per_rank(frame_slices(3, 3))
    for rank_view in range(5):                                        # what each of 5 ranks computes for 3 frames: the same table, the same error
        with pytest.raises(ValueError, match="at least one frame per rank"):
            _require_a_frame_per_rank(frame_slices(3, 5))


FF = 9                                                               # frames of the frame-sharded window: 8 ranks -> slices of 2,1,1,...


def _frames_window():
    blocks, _ = synthetic.attention_q_dumps(FF, H, W, C, num_blocks=3, seed=77)
    return blocks                                                    # three [2F, N, C] fp16 dumps (blocks 8, 7, 6 in Step 3's order)


def _frames_reference(refine):
    blocks = _frames_window()
    np.random.seed(17)
    _, labels, _ = OA.match_gt_mask(OA.aggregate_blocks(blocks), K, np.random.mtrand._rand)
    if refine:
        th, tw = OA.dense_tracking(blocks[1], FF, H, W)
        labels, _ = OA.correct_low_res_mask(labels.reshape(FF, H, W), th, tw)
    return labels.reshape(FF, N)


def _frames_worker(rank, world, port, refine, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vidseg_diffusion_amd.parallel import frame_slices, gather_frames
    blocks = _frames_window()
    sl = frame_slices(FF, world)
    lo, hi = sl[rank]
    # this rank's taps: the conditional half of ITS frames only (what its feature pass would leave), block order 8, 7, 6
    mine = {b: torch.from_numpy(np.ascontiguousarray(blk[FF + lo:FF + hi])) for b, blk in zip((8, 7, 6), blocks)}
    full = {b: gather_frames(mine[b], sl, world).numpy() for b in (8, 7, 6)}
    for b, blk in zip((8, 7, 6), blocks):
        assert np.array_equal(full[b], blk[FF:]), f"rank {rank}: gathered block {b} differs from the window's stack"
    pad = [np.concatenate([np.zeros_like(full[b]), full[b]], 0) for b in (8, 7, 6)]      # the oracle reads [2F, N, C] dumps
    np.random.seed(17)
    _, labels, _ = OA.match_gt_mask(OA.aggregate_blocks(pad), K, np.random.mtrand._rand)
    if refine:
        th, tw = OA.dense_tracking(pad[1], FF, H, W)
        labels, _ = OA.correct_low_res_mask(labels.reshape(FF, H, W), th, tw)
    q.put((rank, labels.reshape(FF, N)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,refine", [(2, True), (8, True)])
def test_frame_sharded_window_equals_one_rank(world, refine):
    """The exchange of the frame-sharded SD window (padded fp16 all-gather of uneven frame slices, parallel.gather_frames) rebuilds
    the window's tap stacks bit for bit on every rank, so Steps 3-3b give the one-rank masks (world 2: 5 + 4 frames; world 8: 2,1,...)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_frames_worker, args=(r, world, port, refine, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _frames_reference(refine)
    for r in range(world):
        assert np.array_equal(res[r], ref), f"rank {r}: frame-sharded labels differ from the one-rank window"
