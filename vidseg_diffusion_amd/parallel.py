"""Multi-GPU execution of the hot path: one process per GPU, windows of a clip sharded across ranks
(SURVEY.md §8(e)); `torch.distributed` backend "nccl" is RCCL over xGMI on this platform.

What shards and what is exchanged
---------------------------------
* The UNet feature pass of a 14-frame window touches no other window -> embarrassingly parallel (for SVD the
  window is the atomic unit because temporal layers mix its frames).
* The reference chains windows through KNN label propagation (feature_extraction.py:603-613, 639-640;
  sd_pipeline_vspw.py:381-387, 401): window b is labelled by a 4-NN vote against window b-1's tokens and
  window b-1's FINAL labels.  Only the vote depends on labels; the neighbour search (the 263-GFLOP part)
  depends on features alone.  So every rank searches its window's top-4 neighbours in window r-1's features
  concurrently, and the label chain itself is a cheap gather resolved identically on every rank.
* Exchange = ONE all-gather of the aggregated, normalised conditional-half features ([F*N, 640] fp16,
  18.4 MB/rank at config 2) so each rank holds its predecessor's tokens AND window 0's, then one all-gather of
  the small int32 results (neighbour indices [F*N,4], tracks [F,N]).  Window 0's K-means runs redundantly on
  every rank from the gathered copy (deterministic kernels: identical labels everywhere), concurrently with the
  neighbour searches -- no rank waits at a broadcast for rank 0's clustering.  No other data-path collective;
  no reduction.

The orchestration below is backend-agnostic (tested under gloo, world_size 2, on CPU with oracle compute
callbacks); `segment_windows_sharded` binds it to the HIP kernels.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np
import torch


@dataclass
class ChainOps:
    """Compute callbacks of the cross-window stage (HIP-backed in the product, oracle-backed in CPU tests)."""
    first_window_labels: Callable      # (feat0 [F*N, C], seed) -> labels int32 [F*N]  (K-means + predict + 4-NN vs frame 0)
    knn_top4: Callable                 # (ref_feat, query_feat) -> int32 [nq, 4]
    vote4: Callable                    # (nn_idx [nq,4], ref_labels [nref]) -> int32 [nq]
    refine: Optional[Callable] = None  # (tracks [F,N] int32, labels [F,N] int32) -> int32 [F,N]


def _host_staged(t: torch.Tensor) -> bool:
    """gloo cannot move device memory: stage through host buffers (CPU tests, and the 2-ranks-on-one-GPU test)."""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_gather(t: torch.Tensor, world: int):
    import torch.distributed as dist
    if _host_staged(t):
        return _all_gather(t.cpu(), world).to(t.device)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous())                  # concatenated along dim 0 (gloo and RCCL agree on this form)
    return out.view((world,) + tuple(t.shape))


def resolve_windows(feat: torch.Tensor, tracks: Optional[torch.Tensor], ops: ChainOps, rank: int, world: int, num_frames: int,
                    seed: int = 17, check: Optional[bool] = None):
    """feat: this rank's normalised tokens fp16 [F*N, C]; tracks: this rank's dense tracks int32 [F, N] or None; seed: this rank's
    window seed (sd_pipeline_vspw.py:255).  Returns final labels of ALL windows, int32 [world, F*N], identical on every rank.

    Window 0's K-means runs redundantly on every rank and must be seeded with WINDOW 0's seed, i.e. rank 0's: each rank's seed
    travels as one extra row of the int32 all-gather and every rank reads row 0's.  check (default: VIDSEG_CHECK_RANKS=1): one more
    tiny all-gather of a checksum of labels0, so a kernel that is not bit-deterministic across devices is detected, not assumed."""
    import os
    import torch.distributed as dist
    FN = feat.shape[0]
    all_feat = _all_gather(feat, world)                               # [W, F*N, C]   <- the RCCL all-gather over xGMI
    nn_idx = torch.full((FN + 1, 4), -1, dtype=torch.int32, device=feat.device)
    if rank != 0:
        nn_idx[:FN] = ops.knn_top4(all_feat[rank - 1], feat).to(torch.int32)
    nn_idx[FN, 0] = int(seed)
    all_idx = _all_gather(nn_idx, world)                              # [W, F*N + 1, 4] (issued before the K-means so it overlaps it)
    seed0 = int(all_idx[0, FN, 0].item())
    all_idx = all_idx[:, :FN]
    labels0 = ops.first_window_labels(all_feat[0], seed0).to(torch.int32)    # every rank, same bits (see module docstring)
    if os.environ.get("VIDSEG_CHECK_RANKS") == "1" if check is None else check:
        w = torch.arange(1, FN + 1, dtype=torch.int64, device=labels0.device)
        digest = ((labels0.to(torch.int64) + 1) * w).sum().reshape(1)
        alld = _all_gather(digest, world).reshape(-1)
        if not bool((alld == alld[0]).all()):
            raise RuntimeError(f"rank {rank}: window 0's redundantly computed labels differ between ranks (checksums {alld.tolist()})")
    all_tracks = _all_gather(tracks, world) if tracks is not None else None
    out = []
    prev = labels0
    for b in range(world):                                            # the sequential label chain, cheap integer work
        lab = prev if b == 0 else ops.vote4(all_idx[b], prev)
        if all_tracks is not None:
            lab = ops.refine(all_tracks[b], lab.view(num_frames, -1)).reshape(-1)
        out.append(lab)
        prev = lab
    return torch.stack(out)


def sharded_feature_pass(engine, latent, c, uc, *, noise=None, num_steps=25, t_start=22, seed=17, rank=0,
                         feature_folder="features_outputs_VSPW", exp_name=None, masks_only=False, feature_timestep=None,
                         inversion_type="add_noise"):
    """This rank's UNet feature pass (Steps 1-2), enqueued on the current stream; only the taps the cross-window stage reads
    are kept (decoder blocks 6-8 at sampler step `feature_timestep`, default the last one, num_steps - 1 -- the drivers'
    `feature_timestep="24"` for their 25 steps, sd_pipeline_vspw.py:633-645).  Returns the handle for `sharded_resolve`."""
    from .pipeline import first_latent, make_denoiser, save_feature_maps, seed_everything
    exp_name = exp_name or f"rank{rank}"
    F, _, lh, lw = latent.shape
    want = num_steps - 1 if feature_timestep is None else int(feature_timestep)
    if not (0 if inversion_type == "inversion" else t_start) <= want < num_steps:
        raise ValueError(f"feature_timestep {want} is outside the sampled steps [{t_start}, {num_steps})")
    if masks_only:                                      # opt-in pruning of the last step, see pipeline.feature_pass
        from .pipeline import feature_pass
        h = feature_pass(engine, latent, c, uc, num_steps=num_steps, t_start=t_start, feature_timestep=str(want), seed=seed,
                         feature_folder=feature_folder, exp_name=exp_name, noise=noise, keep_all_steps=False, masks_only=True,
                         inversion_type=inversion_type)
        return dict(F=F, fh=lh // 2, fw=lw // 2, seed=seed, feature_folder=feature_folder, exp_name=exp_name, done=h["done"],
                    feature_timestep=want)
    from . import ops
    ops.new_window()
    seed_everything(seed)
    sampler = engine.sampler
    denoiser = make_denoiser(engine, F)
    x, t_start = first_latent(engine, denoiser, latent, c, uc, num_steps, t_start, noise, inversion_type)
    net = engine.model.diffusion_model
    hook, mode0 = None, getattr(net, "tap_mode", None)
    if mode0 is not None:                               # Q/K taps only at the step that is dumped (see pipeline.feature_pass)
        def hook(i):
            net.tap_mode = mode0 if i == want else "none"
            net._set_taps()
    try:
        sampler(denoiser, x, cond=c, uc=uc, t_start=t_start, step_hook=hook,
                img_callback=lambda xt, i: save_feature_maps(engine, feature_folder, exp_name, i, xt=xt, block_filter=(6, 7, 8)) if i == want else None)
    finally:
        if hook is not None:
            net.tap_mode = mode0
            net._set_taps()
    done = torch.cuda.Event()
    done.record(torch.cuda.current_stream())
    return dict(F=F, fh=lh // 2, fw=lw // 2, seed=seed, feature_folder=feature_folder, exp_name=exp_name, done=done,
                feature_timestep=want)


def sharded_resolve(engine, h, *, num_masks=20, is_aggre_attn=True, is_refine_mask=False, rank=0, world=1):
    """Cross-window stage of one step on the current stream (may differ from the feature pass's): block aggregation, dense
    tracking, the all-gather / top-4 search / label chain of `resolve_windows`.  Returns int64 [world, F, N]."""
    from . import analysis as A
    from . import feature_extraction as FE
    torch.cuda.current_stream().wait_event(h["done"])
    F, fh, fw, seed = h["F"], h["fh"], h["fw"], h["seed"]
    N = fh * fw
    store = FE.FeatureStore.folder(h["feature_folder"], h["exp_name"])
    names = ["output_block_8", "output_block_7", "output_block_6"] if is_aggre_attn else \
        (["output_block_8"] if engine.video else ["output_block_7"])
    ts = h["feature_timestep"]
    blocks = [store[f"{n}_spatial_self_attn_q_time_{ts}"] for n in names]
    _, feat = A.mean_normalize(blocks, F * N, F * N)
    tracks = None
    if is_refine_mask:
        q7 = store[f"output_block_7_spatial_self_attn_q_time_{ts}"]
        tracks, _ = A.dense_tracking(q7[F:2 * F].contiguous(), F, fh, fw)

    def first_window(feat0, seed0):
        np.random.seed(seed0)                                             # window 0's seed_everything (SDP:255) = rank 0's seed
        km = A.kmeans_fit(feat0, num_masks, n_init=10)
        fake = A.kmeans_predict(feat0[:N], km.centers)                    # identity cluster->label map (no GT mask, FE:586-595)
        return A.knn_predict(feat0[:N].contiguous(), fake, feat0)

    ops = ChainOps(first_window_labels=first_window, knn_top4=A.knn_top4, vote4=A.vote4,
                   refine=(lambda t, l: A.trajectory_vote(t.contiguous(), l.contiguous(), fw)) if is_refine_mask else None)
    labels = resolve_windows(feat, tracks, ops, rank, world, F, seed=seed)
    out = labels.view(world, F, N).cpu().numpy().astype(np.int64)
    FE.FeatureStore.clear(h["feature_folder"], h["exp_name"])
    return out


class ShardedPipeline:
    """The multi-rank counterpart of pipeline.WindowPipeline: step i's cross-window stage (collectives included) runs on a second
    HIP stream while the feature passes of the next `lanes` steps are already queued (each lane on its own stream).  Every rank
    pushes/flushes in the same order, so the collectives of all ranks stay matched."""

    def __init__(self, engine, rank, world, lanes=1, **resolve_kw):
        self.engine, self.rank, self.world, self.resolve_kw = engine, rank, world, resolve_kw
        self.side = torch.cuda.Stream()
        self.lanes = [torch.cuda.Stream() for _ in range(lanes)] if lanes > 1 else [None]
        self.pending = []
        self.count = 0

    def _resolve(self, h):
        with torch.cuda.stream(self.side):
            return sharded_resolve(self.engine, h, rank=self.rank, world=self.world, **self.resolve_kw)

    def push(self, latent, c, uc, **feature_kw):
        lane = self.lanes[self.count % len(self.lanes)]
        self.count += 1
        if lane is None:
            h = sharded_feature_pass(self.engine, latent, c, uc, rank=self.rank, **feature_kw)
        else:
            from .pipeline import hand_to_stream
            lane.wait_stream(torch.cuda.current_stream())
            hand_to_stream(lane, latent, c, uc, feature_kw.get("noise"))
            with torch.cuda.stream(lane):
                h = sharded_feature_pass(self.engine, latent, c, uc, rank=self.rank, **feature_kw)
        self.pending.append(h)
        return self._resolve(self.pending.pop(0)) if len(self.pending) > len(self.lanes) else None

    def drain(self):
        out = [self._resolve(h) for h in self.pending]
        self.pending = []
        self.side.synchronize()
        return out

    def flush(self):
        out = self.drain()
        return out[-1] if out else None


def segment_windows_sharded(engine, latent, c, uc, *, noise=None, num_masks=20, num_steps=25, t_start=22, is_aggre_attn=True,
                            is_refine_mask=False, seed=17, rank=0, world=1, feature_folder="features_outputs_VSPW", exp_name=None,
                            masks_only=False, feature_timestep=None, inversion_type="add_noise"):
    """Each rank segments its own window (`latent` is THIS rank's [F,4,h,w]); returns int64 labels:
    world == 1 -> [F, N] (exactly pipeline.segment_window); world > 1 -> [world, F, N], same on every rank."""
    from .pipeline import segment_window
    exp_name = exp_name or f"rank{rank}"
    if world == 1:
        labels, _ = segment_window(engine, latent, c, uc, num_masks=num_masks, num_steps=num_steps, t_start=t_start,
                                   is_aggre_attn=is_aggre_attn, is_refine_mask=is_refine_mask, seed=seed, noise=noise,
                                   feature_folder=feature_folder, exp_name=exp_name, keep_all_steps=False, masks_only=masks_only,
                                   feature_timestep=str(num_steps - 1 if feature_timestep is None else int(feature_timestep)),
                                   inversion_type=inversion_type)
        return labels
    h = sharded_feature_pass(engine, latent, c, uc, noise=noise, num_steps=num_steps, t_start=t_start, seed=seed, rank=rank,
                             feature_folder=feature_folder, exp_name=exp_name, masks_only=masks_only, feature_timestep=feature_timestep,
                             inversion_type=inversion_type)
    return sharded_resolve(engine, h, num_masks=num_masks, is_aggre_attn=is_aggre_attn, is_refine_mask=is_refine_mask, rank=rank,
                           world=world)
