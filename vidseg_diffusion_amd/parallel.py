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


def exchange_seed(seed: int, world: int, device) -> int:
    """Rank 0's window seed on every rank.  Window 0's K-means runs redundantly on every rank and must draw WINDOW 0's numbers
    (sd_pipeline_vspw.py:255), i.e. rank 0's seed: one int64 all-gather of a single element (any seed numpy accepts survives),
    read back on the host.  Callers issue it BEFORE the step's feature pass is queued on a stream that is idle, so the read-back
    waits for nothing and the collectives of `resolve_windows` stay asynchronous."""
    if world == 1:
        return int(seed)
    import torch.distributed as dist
    t = torch.tensor([int(seed)], dtype=torch.int64, device="cpu" if dist.get_backend() == "gloo" else device)
    return int(_all_gather(t, world).reshape(-1)[0].item())


def resolve_windows(feat: torch.Tensor, tracks: Optional[torch.Tensor], ops: ChainOps, rank: int, world: int, num_frames: int,
                    seed: int = 17, check: Optional[bool] = None, seed0: Optional[int] = None):
    """feat: this rank's normalised tokens fp16 [F*N, C]; tracks: this rank's dense tracks int32 [F, N] or None; seed: this rank's
    window seed (sd_pipeline_vspw.py:255); seed0: rank 0's seed if the caller already exchanged it (`exchange_seed`, done by
    ShardedPipeline.push / segment_windows_sharded before the feature pass is queued), else it is exchanged here, first.
    Returns final labels of ALL windows, int32 [world, F*N], identical on every rank.

    check (default: VIDSEG_CHECK_RANKS=1): one more tiny all-gather of a checksum of labels0, so a kernel that is not
    bit-deterministic across devices is detected, not assumed."""
    import os
    FN = feat.shape[0]
    if seed0 is None:
        seed0 = exchange_seed(seed, world, feat.device)
    all_feat = _all_gather(feat, world)                               # [W, F*N, C]   <- the RCCL all-gather over xGMI
    nn_idx = torch.full((FN, 4), -1, dtype=torch.int32, device=feat.device)
    if rank != 0:
        nn_idx[:] = ops.knn_top4(all_feat[rank - 1], feat).to(torch.int32)
    all_idx = _all_gather(nn_idx, world)                              # [W, F*N, 4]: queued before the K-means, no host read in between
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
    labels = resolve_windows(feat, tracks, ops, rank, world, F, seed=seed, seed0=h.get("seed0"))
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
        seed0 = exchange_seed(feature_kw.get("seed", 17), self.world, latent.device)   # on the caller's (idle) stream, before anything is queued
        if lane is None:
            h = sharded_feature_pass(self.engine, latent, c, uc, rank=self.rank, **feature_kw)
        else:
            from .pipeline import hand_to_stream
            lane.wait_stream(torch.cuda.current_stream())
            hand_to_stream(lane, latent, c, uc, feature_kw.get("noise"))
            with torch.cuda.stream(lane):
                h = sharded_feature_pass(self.engine, latent, c, uc, rank=self.rank, **feature_kw)
        h["seed0"] = seed0
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
    seed0 = exchange_seed(seed, world, latent.device)
    h = sharded_feature_pass(engine, latent, c, uc, noise=noise, num_steps=num_steps, t_start=t_start, seed=seed, rank=rank,
                             feature_folder=feature_folder, exp_name=exp_name, masks_only=masks_only, feature_timestep=feature_timestep,
                             inversion_type=inversion_type)
    h["seed0"] = seed0
    return sharded_resolve(engine, h, num_masks=num_masks, is_aggre_attn=is_aggre_attn, is_refine_mask=is_refine_mask, rank=rank,
                           world=world)


# ----------------------------------------------------------------------------------------------------------------------------------
# SD frame-level sharding ("SD (C2-like)"): UNetModel has no cross-frame operator (openaimodel.py:831-954 -- GroupNorm
# statistics, attention and convolutions are all per sample), so the F frames of ONE window shard over the ranks: rank r runs the
# feature pass on its contiguous frame slice (each frame's CFG pair stays together: batch 2 * F_r), the conditional half's Q taps
# of decoder blocks 6-8 are all-gathered (one fp16 exchange per block, padded to the largest slice), and Steps 3-3b run on the
# gathered stack on every rank (deterministic kernels: same bits everywhere; rank 0's are the result).  The SVD VideoUNet mixes
# the frames of a window (temporal attention / conv), so for it the atomic unit stays the window (the functions above).
# ----------------------------------------------------------------------------------------------------------------------------------
def frame_slices(num_frames: int, world: int):
    """Contiguous frame ranges [(lo, hi)] per rank, sizes differing by at most one (the first num_frames % world ranks hold one more);
    ranks beyond the frame count hold an empty slice (which the frame-sharded entry points refuse, on every rank alike)."""
    base, extra = divmod(num_frames, world)
    out, lo = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((lo, lo + n))
        lo += n
    return out


def gather_frames(local: torch.Tensor, slices, world: int) -> torch.Tensor:
    """All-gather of per-rank frame slices [F_r, ...] into the window's [F, ...] (slices padded to the largest one: a rank with
    fewer frames sends zero rows that are dropped on arrival)."""
    fmax = max(hi - lo for lo, hi in slices)
    pad = torch.zeros((fmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    allp = _all_gather(pad, world)                                    # [W, fmax, ...]
    return torch.cat([allp[r, :hi - lo] for r, (lo, hi) in enumerate(slices)], 0)


def _require_a_frame_per_rank(slices):
    """More ranks than frames: decided from the slice table, which is identical on every rank, BEFORE any collective -- so every rank
    raises (a check on the empty ranks alone left the others blocked in the all-gather until the collective timed out)."""
    if any(hi == lo for lo, hi in slices):
        raise ValueError(f"frame sharding needs at least one frame per rank: {slices[-1][1]} frames over {len(slices)} ranks")


def frame_sharded_feature_pass(engine, latent, c, uc, *, noise=None, num_steps=25, t_start=22, seed=17, rank=0, world=1,
                               feature_folder="features_outputs_VSPW", exp_name=None, masks_only=False):
    """Steps 1-2 of ONE SD window on this rank's frame slice (`latent`, `c`, `uc`, `noise` are the WINDOW's full tensors, identical on
    every rank; a missing `noise` is drawn for the whole window under the window's seed, then sliced, so every frame sees the noise
    the one-rank pass gives it).  Returns the handle for `frame_sharded_resolve`."""
    from .pipeline import seed_everything
    if engine.video:
        raise ValueError("frame-level sharding is for the per-sample SD UNet; the SVD VideoUNet shards by window (segment_windows_sharded)")
    F = latent.shape[0]
    sl = frame_slices(F, world)
    _require_a_frame_per_rank(sl)                                         # every rank sees the same slices: all raise, none enters a collective
    lo, hi = sl[rank]
    if noise is None:
        seed_everything(seed)                                             # SDP:255, then add_noise's randn_like over the whole window (SAM:138)
        noise = torch.randn(latent.shape, dtype=latent.dtype, device=latent.device)
    exp_name = exp_name or f"frames{rank}"
    cut = lambda d: {k: v[lo:hi] for k, v in d.items()}                   # noqa: E731
    h = sharded_feature_pass(engine, latent[lo:hi].contiguous(), cut(c), cut(uc), noise=noise[lo:hi].contiguous(), num_steps=num_steps,
                             t_start=t_start, seed=seed, rank=rank, feature_folder=feature_folder, exp_name=exp_name, masks_only=masks_only)
    return dict(h=h, F=F, fh=latent.shape[2] // 2, fw=latent.shape[3] // 2, slices=sl, seed=seed, feature_folder=feature_folder,
                exp_name=exp_name, feature_timestep=num_steps - 1, device=latent.device)


def frame_sharded_resolve(engine, fh_, *, num_masks=20, is_aggre_attn=True, is_refine_mask=False, rank=0, world=1, analysis=None):
    """Exchange + Steps 3-3b of a frame-sharded window: the conditional half's block-6/7/8 Q taps of every rank's frames are
    all-gathered into the window's [F, N, 640] stacks, then the one-window analysis (3-block mean, max-abs normalise, K-means++ /
    Lloyd best-of-10, 4-NN, optional dense tracking + vote) runs on every rank.  Returns int64 [F, N], identical on every rank.
    analysis: test hook -- a callable (blocks {6,7,8: fp16 [F, N, C]}, F, fh, fw, seed) -> labels replacing the device analysis."""
    from . import analysis as A
    from . import feature_extraction as FE
    F, fh, fw, sl = fh_["F"], fh_["fh"], fh_["fw"], fh_["slices"]
    N = fh * fw
    h = fh_["h"]
    _require_a_frame_per_rank(sl)
    lo, hi = sl[rank]
    ts = fh_["feature_timestep"]
    names = (8, 7, 6) if is_aggre_attn else (7,)
    need = sorted(set(names) | ({7} if is_refine_mask else set()))
    try:
        torch.cuda.current_stream().wait_event(h["done"])
        store = FE.FeatureStore.folder(h["feature_folder"], h["exp_name"])
        Fr = hi - lo
        taps = {b: store[f"output_block_{b}_spatial_self_attn_q_time_{ts}"][Fr:2 * Fr] for b in need}     # conditional half (FE:550-551)
        if world > 1:
            taps = {b: gather_frames(taps[b].contiguous(), sl, world) for b in need}
        if analysis is not None:
            return analysis(taps, F, fh, fw, fh_["seed"])
        _, feat = A.mean_normalize([taps[b].contiguous() for b in names], 0, F * N)      # the gathered stacks hold the conditional half only
        np.random.seed(fh_["seed"])
        km = A.kmeans_fit(feat, num_masks, n_init=10)
        fake = A.kmeans_predict(feat[:N], km.centers)
        labels = A.knn_predict(feat[:N].contiguous(), fake, feat)
        if is_refine_mask:
            tracks, _ = A.dense_tracking(taps[7].contiguous(), F, fh, fw)
            labels = A.trajectory_vote(tracks.contiguous(), labels.view(F, N).contiguous(), fw).reshape(-1)
        return labels.view(F, N).cpu().numpy().astype(np.int64)
    finally:                                                              # the rank's dumps leave the store on every path (hook, error)
        FE.FeatureStore.clear(h["feature_folder"], h["exp_name"])


def segment_window_frame_sharded(engine, latent, c, uc, *, rank=0, world=1, num_masks=20, is_aggre_attn=True, is_refine_mask=False,
                                 **feature_kw):
    """One SD window over `world` ranks by frames; every rank passes the window's full inputs and receives the window's labels."""
    h = frame_sharded_feature_pass(engine, latent, c, uc, rank=rank, world=world, **feature_kw)
    return frame_sharded_resolve(engine, h, num_masks=num_masks, is_aggre_attn=is_aggre_attn, is_refine_mask=is_refine_mask, rank=rank,
                                 world=world)
