"""Two ranks on ONE GPU at FULL size: the window-sharded path of BASELINE configs[3] (SVD, one 14-frame 576x1024 window per
rank) and of the headline's N > 1 line (SD, one 14-frame 512x512 window per rank), in the parity mode bench.py quotes (exact precision +
masks_only), gloo rendezvous with host-staged collectives (one box has one GPU; the collectives are the ones RCCL carries on a node).

The pytest process IS rank 0 and uses the module's shared full-width network; rank 1 is a spawned process that builds its own copy.
Checks: (1) both ranks end with the same label stack [2, F, N]; (2) it is, bit for bit, the sequential two-window loop of the reference
(sd_pipeline_vspw.py:228-409 / svd_pipeline_vspw.py:232-395: K-means on window 0, 4-NN chain into window 1, refinement per window) run in
this process; (3) rank 0's window carries the REFERENCE's masks for that window (tests/golden, generated from /root/reference)."""
import os
import socket
import sys
from datetime import timedelta

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FOLDER = "/nonexistent/two_rank_full"


def _window_kwargs(svd):
    return dict(num_masks=20, num_steps=25, t_start=17 if svd else 22, is_aggre_attn=True, is_refine_mask=svd, seed=17, masks_only=True)


def rank1_main(port, svd, q):
    """The spawned rank: its own full-width network (bench.build: the same seeded weights), window 1 of the bench's clip."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        import torch.distributed as dist
        import bench
        from vidseg_diffusion_amd import parallel
        torch.set_grad_enabled(False)
        dev = torch.device("cuda:0")
        eng, cfg, _sd, _n = bench.build(svd, False, dev)
        eng.model.diffusion_model.set_precision("exact")
        q.put(("ready", None))                                           # the parent forms the group only once this rank is known to be alive
        dist.init_process_group("gloo", rank=1, world_size=2, timeout=timedelta(seconds=600))
        lat, c, uc, noise = bench.make_inputs(dev, 1, cfg, svd=svd, lat_hw=(72, 128) if svd else (64, 64))
        labels = parallel.segment_windows_sharded(eng, lat, c, uc, noise=noise, rank=1, world=2, feature_folder=FOLDER, exp_name="r1",
                                                  **_window_kwargs(svd))
        q.put(("ok", labels))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as e:                                           # noqa: BLE001 -- the parent must not wait for the queue's timeout
        q.put(("error", repr(e)))
        raise


def run(net, cfg, svd):
    """Returns (labels of rank 0 [2, F, N], labels of rank 1, sequential loop's [2, F, N])."""
    import torch.distributed as dist
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd import parallel
    from vidseg_diffusion_amd.pipeline import WindowState, build_sd_engine, build_svd_engine, segment_window
    dev = torch.device("cuda:0")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=rank1_main, args=(port, svd, q))
    p.start()
    eng = build_svd_engine(net, num_frames=14, num_steps=25) if svd else build_sd_engine(net, num_steps=25, scale=5.0)
    kw = _window_kwargs(svd)
    hw = (72, 128) if svd else (64, 64)
    ins = [bench.make_inputs(dev, w, cfg, svd=svd, lat_hw=hw) for w in (0, 1)]
    net.set_precision("exact")
    try:
        # the sequential loop first: rank 1 is still building its network meanwhile
        FE.FeatureStore.clear()
        FE.MaskStore.clear()
        state, seq = WindowState(), []
        for w, (lat, c, uc, noise) in enumerate(ins):
            labels, state = segment_window(eng, lat, c, uc, state=state, noise=noise, feature_folder=FOLDER, exp_name=f"s{w}", keep_all_steps=False, **kw)
            seq.append(np.asarray(labels))
        seq = np.stack(seq)
        status, info = q.get(timeout=600)
        assert status == "ready", info
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=0, world_size=2, timeout=timedelta(seconds=600))
        try:
            lat, c, uc, noise = ins[0]
            mine = parallel.segment_windows_sharded(eng, lat, c, uc, noise=noise, rank=0, world=2, feature_folder=FOLDER, exp_name="r0", **kw)
            status, other = q.get(timeout=600)
            assert status == "ok", other
            dist.barrier()
        finally:
            dist.destroy_process_group()
    finally:
        net.set_precision("fp16")
        FE.FeatureStore.clear()
        FE.MaskStore.clear()
        p.join(timeout=120)
        if p.is_alive():
            p.kill()
    assert p.exitcode == 0
    return np.asarray(mine), np.asarray(other), seq
