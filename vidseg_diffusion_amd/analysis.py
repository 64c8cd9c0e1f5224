"""Host side of the post-UNet analysis path on MI355X.

Python here only sequences HIP kernels (through the C ABI) and holds the small amount of host
logic the reference keeps on the host too: the numpy RandomState draws that make sklearn's
k-means++ deterministic (scripts/sampling/sd_pipeline_vspw.py:619-623 seeds numpy's global
stream; sklearn draws from it), the per-restart convergence bookkeeping of
sklearn/cluster/_kmeans.py:699-752 and the best-of-n_init rule (:1525-1531).  All arithmetic on
features runs on the GPU.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, require_gpu, stream

I32 = torch.int32
F64 = torch.float64
F16 = torch.float16


def _dev(t):
    return t.device


# --------------------------------------------------------------------------------------
# a13 + a14(2): aggregate decoder blocks, keep the conditional half, max-abs normalise
# --------------------------------------------------------------------------------------
def mean_normalize(blocks, row0: int, rows: int, want_mean: bool = False):
    """blocks: list of fp16 [R_total, C] (or [2F, N, C]) device tensors of identical shape.
    Returns (mean16 or None, norm16) for rows [row0, row0+rows) of the flattened token matrix:
    mean16 = fp16 mean over blocks (feature_extraction.py:745), norm16 = mean16 / max|mean16|
    per token (feature_extraction.py:554-555)."""
    require_gpu(*blocks)
    C = blocks[0].shape[-1]
    for b in blocks:
        if b.dtype != F16 or b.shape != blocks[0].shape:
            raise _lib.VidsegError("mean_normalize: blocks must be fp16 tensors of identical shape")
    out_norm = torch.empty((rows, C), dtype=F16, device=_dev(blocks[0]))
    out_mean = torch.empty((rows, C), dtype=F16, device=_dev(blocks[0])) if want_mean else None
    arr = (ctypes.c_void_p * len(blocks))(*[b.data_ptr() for b in blocks])
    call("vidseg_mean_normalize_f16", arr, len(blocks), row0, rows, C, ptr(out_mean), ptr(out_norm), stream())
    return out_mean, out_norm


# --------------------------------------------------------------------------------------
# sklearn KMeans(n_clusters, n_init=10) on the GPU, restarts batched
# --------------------------------------------------------------------------------------
class KMeansResult:
    """all_labels int32 [R, n] (device) / all_inertia float64 [R] (host) / all_n_iter: every restart, not only the winner -- what
    sklearn's fit loop sees before it keeps the best (_kmeans.py:1497-1531); read by the restart-equivalence parity test."""
    __slots__ = ("centers", "labels", "inertia", "n_iter", "best_restart", "total_lloyd_iters", "all_labels", "all_inertia", "all_n_iter")


KEEP_LAST = any(os.environ.get(k) for k in ("VIDSEG_KEEP_LAST", "VIDSEG_DEBUG_HASH"))   # tests / study tools set this: the two globals pin device tensors
LAST_CENTER_IDS = None
LAST_KMEANS = None          # the most recent KMeansResult of this process (diagnostics / parity tests; never read by the product path)


def _draw_kpp_uniforms(n, K, R, random_state):
    """Consume the legacy RandomState stream exactly as sklearn's _kmeans_plusplus does
    (cluster/_kmeans.py:225, :243) for R consecutive restarts."""
    T = 2 + int(np.log(K))
    p = np.ones(n, dtype=np.float64) / np.float64(n)
    cdf = p.cumsum()
    cdf /= cdf[-1]
    first = np.empty(R, dtype=np.int32)
    U = np.zeros((R, max(1, (K - 1) * T)), dtype=np.float64)
    for r in range(R):
        first[r] = cdf.searchsorted(random_state.random_sample(), side="right")
        for c in range(1, K):
            U[r, (c - 1) * T:c * T] = random_state.uniform(size=T)
    return T, first, U


def _kmeanspp_init(x16, mean, xsq, n, C, K, R, rs, dev, st):
    """k-means++ for R restarts (cluster/_kmeans.py:174-274): numpy's uniforms are pre-drawn on the host in sklearn's
    order, distances / cumulative sums / candidate search run on the device.  Returns centres [R, K, C] f64 (centred)."""
    T, first, U = _draw_kpp_uniforms(n, K, R, rs)
    Tmax = max(T, 1)
    ntiles = (n + 63) // 64
    closest = torch.full((R, n), float("inf"), dtype=F64, device=dev)
    dcand = torch.empty((R * Tmax, n), dtype=F64, device=dev)
    part = torch.empty((R * Tmax, ntiles), dtype=F64, device=dev)
    pot = torch.empty(R, dtype=F64, device=dev)
    cand = torch.zeros(2 * R * Tmax, dtype=I32, device=dev)             # two halves: round c reads half (c - 1) & 1, writes half c & 1
    cand[:R] = torch.from_numpy(first).to(dev)
    center_ids = torch.empty(R * K, dtype=I32, device=dev)
    U_dev = torch.from_numpy(U).to(dev)
    ustride = U.shape[1]
    for c in range(0, K + 1):
        tprev = 0 if c == 0 else (1 if c == 1 else T)
        tnext = 1 if c == 0 else (T if c < K else 0)
        u_ptr = U_dev.data_ptr() + 8 * (c - 1) * T if 1 <= c < K else None
        call("vidseg_kpp_round_v2", ptr(x16), ptr(mean), ptr(xsq), n, C, R, K, c, tprev, tnext, Tmax, u_ptr, ustride,
             ptr(closest), ptr(dcand), ptr(part), ptr(pot), ptr(cand), cand.numel(), ptr(center_ids), st)
    centers = torch.empty((R, K, C), dtype=F64, device=dev)
    call("vidseg_gather_rows_f64", ptr(x16), ptr(mean), C, ptr(center_ids), R * K, ptr(centers), st)
    if KEEP_LAST:
        global LAST_CENTER_IDS
        LAST_CENTER_IDS = center_ids                                  # diagnostics only (tools/kmeans_race.py): the seeds picked, [R * K]
    return centers


def kmeans_fit(x16: torch.Tensor, n_clusters: int, n_init: int = 10, max_iter: int = 300, tol: float = 1e-4,
               random_state=None, chunk: int = 256, init=None) -> KMeansResult:
    """KMeans(n_clusters, n_init).fit(x16) -- feature_extraction.py:562-570 / :52-54.

    x16: fp16 [n, C] device tensor (normalised tokens).  float64 arithmetic on the device like
    sklearn's (validate_data up-casts fp16 to float64, _kmeans.py:1458): centring, k-means++,
    Lloyd, inertia.  `random_state=None` means numpy's global RandomState, as in the reference.
    """
    require_gpu(x16)
    if x16.dtype != F16 or x16.dim() != 2:
        raise _lib.VidsegError("kmeans_fit: x16 must be a 2-D fp16 tensor")
    n, C = x16.shape
    K, R = int(n_clusters), int(n_init)
    if init is not None:                                              # KMeans(init=ndarray): a single run from these centres
        R = 1
    if n < K:
        raise ValueError(f"n_samples={n} should be >= n_clusters={K}.")     # sklearn's message
    rs = np.random.mtrand._rand if random_state is None else random_state
    dev = x16.device
    st = stream()
    mean = torch.empty(C, dtype=F64, device=dev)
    xsq = torch.empty(n, dtype=F64, device=dev)
    colvar = torch.empty(C, dtype=F64, device=dev)
    scratch = torch.empty(2 * ((n + 255) // 256) * C, dtype=F64, device=dev)
    call("vidseg_kmeans_prepare", ptr(x16), n, C, ptr(mean), ptr(xsq), ptr(colvar), ptr(scratch), st)

    if init is not None:
        init_t = torch.as_tensor(init, dtype=F64).to(dev).reshape(1, K, C)
        centers = (init_t - mean[None, None, :]).contiguous()           # _kmeans.py:1490 `init -= X_mean`
    else:
        centers = _kmeanspp_init(x16, mean, xsq, n, C, K, R, rs, dev, st)

    tol_ = float(colvar.mean().item()) * tol                        # _tolerance(), _kmeans.py:279-287
    labels = torch.full((R, n), -1, dtype=I32, device=dev)
    changed = torch.zeros(R, dtype=I32, device=dev)
    cnorm = torch.empty(R * K, dtype=F64, device=dev)
    sums = torch.zeros((R, K, C), dtype=F64, device=dev)              # exact raw member sums (see vidseg_lloyd_step)
    ub = torch.empty((R, n), dtype=F64, device=dev)
    lb = torch.empty((R, n), dtype=F64, device=dev)
    lst = torch.arange(n, dtype=I32, device=dev).repeat(R, 1).contiguous()
    nlist = torch.full((R,), n, dtype=I32, device=dev)
    chg = torch.empty((R, n, 2), dtype=I32, device=dev)
    delta = torch.zeros((R, K), dtype=F64, device=dev)
    dtop = torch.zeros((R, 3), dtype=F64, device=dev)
    reloc_d = torch.empty((R, n), dtype=F64, device=dev)               # empty-cluster relocation scratch (rarely touched)
    reloc_t = torch.empty((R, n), dtype=I32, device=dev)
    reloc = torch.zeros((R, 64, 3), dtype=I32, device=dev)
    nreloc = torch.zeros(R, dtype=I32, device=dev)
    shift2 = torch.zeros((R, K), dtype=F64, device=dev)
    counts = torch.zeros((R, K), dtype=I32, device=dev)
    # device-side convergence state: [active mask, strict mask, error flags, n_iter[R]]
    h_state = torch.zeros(3 + R, dtype=I32)
    h_state[0] = (1 << R) - 1
    state = h_state.to(dev)
    h_pin = torch.empty(3 + R, dtype=I32).pin_memory()
    total = 0
    it = 0
    cur_mask = (1 << R) - 1
    slots, colrow, nslots = _compact_slots(cur_mask, R, K, dev)
    while it < max_iter:
        poll = 4 if it < 16 else 8                                      # iterations enqueued between host polls
        for _ in range(min(poll, max_iter - it)):
            call("vidseg_lloyd_step", ptr(x16), ptr(mean), ptr(xsq), n, C, R, K, it, tol_, ptr(state), ptr(slots), nslots,
                 ptr(centers), ptr(cnorm), ptr(sums), ptr(counts), ptr(labels), ptr(ub), ptr(lb), ptr(lst), ptr(nlist), ptr(chg),
                 ptr(changed), ptr(shift2), ptr(delta), ptr(dtop), ptr(reloc_d), ptr(reloc_t), ptr(reloc), ptr(nreloc), st)
            it += 1
        h_pin.copy_(state, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        total = it
        if int(h_pin[2]) != 0:
            raise _lib.VidsegError(f"kmeans_fit: device error flags {int(h_pin[2]):#x}")
        if int(h_pin[0]) == 0:
            break
        if int(h_pin[0]) != cur_mask:                                    # drop converged restarts from the launch grids
            cur_mask = int(h_pin[0])
            slots, colrow, nslots = _compact_slots(cur_mask, R, K, dev)
    strict_mask = int(h_pin[1])
    n_iter = [int(h_pin[3 + r]) for r in range(R)]
    if os.environ.get("VIDSEG_DEBUG_KMEANS"):
        print("kmeans n_iter per restart", n_iter, "strict mask", bin(strict_mask), "total lock-step iterations", total)
    rerun = ((1 << R) - 1) & ~strict_mask
    if rerun:                                                          # _kmeans.py:736-748
        rstate = torch.tensor([rerun], dtype=I32).to(dev)
        slots, colrow, nslots = _compact_slots(rerun, R, K, dev)
        call("vidseg_lloyd_iter", ptr(x16), ptr(mean), n, C, R, K, ptr(rstate), ptr(slots), nslots, ptr(colrow), 0, ptr(centers),
             ptr(cnorm), ptr(labels), ptr(changed), None, None, chunk, None, None, st)
    ipart = torch.empty((R, (n + 255) // 256), dtype=F64, device=dev)
    inertia = torch.empty(R, dtype=F64, device=dev)
    call("vidseg_kmeans_inertia", ptr(x16), ptr(mean), n, C, R, K, ptr(centers), ptr(labels), ptr(ipart), ptr(inertia), st)
    h_inertia = inertia.cpu().numpy()
    h_labels = None
    best = 0
    for r in range(1, R):                                               # _kmeans.py:1525-1531
        if h_inertia[r] < h_inertia[best]:
            if h_labels is None:
                h_labels = labels.cpu().numpy()
            if not _is_same_clustering(h_labels[r], h_labels[best], K):
                best = r
    res = KMeansResult()
    bc = centers[best].clone()
    call("vidseg_add_mean_f64", ptr(bc), ptr(mean), K, C, st)            # best_centers += X_mean
    res.centers = bc
    res.labels = labels[best]
    res.inertia = float(h_inertia[best])
    res.n_iter = n_iter[best]
    res.best_restart = best
    res.total_lloyd_iters = total
    res.all_labels, res.all_inertia, res.all_n_iter = labels, h_inertia, n_iter
    if KEEP_LAST:
        global LAST_KMEANS
        LAST_KMEANS = res
    return res


def _compact_slots(mask, R, K, dev):
    """Restart ids still running (bit mask) -> (slots int32 [ns], colrow int32 [ns*K] = centre row r*K+k per compact column, ns)."""
    ids = [r for r in range(R) if (mask >> r) & 1]
    slots = torch.tensor(ids, dtype=I32)
    colrow = (slots[:, None] * K + torch.arange(K, dtype=I32)[None, :]).reshape(-1).to(I32)
    return slots.to(dev), colrow.contiguous().to(dev), len(ids)


def _is_same_clustering(l1, l2, K):
    """sklearn/cluster/_k_means_common.pyx:314-328 (vectorised: the map l1->l2 must be a function)."""
    pair = l1.astype(np.int64) * K + l2.astype(np.int64)
    upairs = np.unique(pair)
    return np.unique(upairs // K).size == upairs.size


def kmeans_predict(x16: torch.Tensor, centers: torch.Tensor) -> torch.Tensor:
    """KMeans.predict (feature_extraction.py:572, :55): E-step on the uncentred float64 data."""
    require_gpu(x16, centers)
    n, C = x16.shape
    K = centers.shape[0]
    dev = x16.device
    labels = torch.full((1, n), -1, dtype=I32, device=dev)
    changed = torch.zeros(1, dtype=I32, device=dev)
    cnorm = torch.empty(K, dtype=F64, device=dev)
    one = torch.ones(1, dtype=I32, device=dev)
    slots, colrow, _ = _compact_slots(1, 1, K, dev)
    call("vidseg_lloyd_iter", ptr(x16), None, n, C, 1, K, ptr(one), ptr(slots), 1, ptr(colrow), 0, ptr(centers), ptr(cnorm),
         ptr(labels), ptr(changed), None, None, 256, None, None, stream())
    return labels[0]


# --------------------------------------------------------------------------------------
# KNeighborsClassifier(n_neighbors=4).fit(ref, ref_labels).predict(query)
# --------------------------------------------------------------------------------------
def knn_predict(ref16: torch.Tensor, ref_labels: torch.Tensor, query16: torch.Tensor) -> torch.Tensor:
    """feature_extraction.py:608-613.  Brute-force float64 distances like sklearn's fp16 path."""
    require_gpu(ref16, ref_labels, query16)
    nq, C = query16.shape
    nr = ref16.shape[0]
    dev = query16.device
    qq = torch.empty(nq, dtype=F64, device=dev)
    yy = torch.empty(nr, dtype=F64, device=dev)
    st = stream()
    call("vidseg_row_sqnorm_f64", ptr(query16), nq, C, ptr(qq), st)
    call("vidseg_row_sqnorm_f64", ptr(ref16), nr, C, ptr(yy), st)
    out = torch.empty(nq, dtype=I32, device=dev)
    call("vidseg_knn_vote", ptr(query16), nq, ptr(ref16), nr, C, ptr(qq), ptr(yy), ptr(ref_labels), ptr(out), st)
    return out


def knn_top4(ref16: torch.Tensor, query16: torch.Tensor) -> torch.Tensor:
    """Label-independent half of the classifier: indices of the 4 nearest reference rows, int32 [nq, 4]."""
    require_gpu(ref16, query16)
    nq, C = query16.shape
    nr = ref16.shape[0]
    dev = query16.device
    qq = torch.empty(nq, dtype=F64, device=dev)
    yy = torch.empty(nr, dtype=F64, device=dev)
    st = stream()
    call("vidseg_row_sqnorm_f64", ptr(query16), nq, C, ptr(qq), st)
    call("vidseg_row_sqnorm_f64", ptr(ref16), nr, C, ptr(yy), st)
    out = torch.empty((nq, 4), dtype=I32, device=dev)
    call("vidseg_knn_top4", ptr(query16), nq, ptr(ref16), nr, C, ptr(qq), ptr(yy), ptr(out), st)
    return out


def vote4(nn_idx: torch.Tensor, ref_labels: torch.Tensor) -> torch.Tensor:
    """Label-dependent half: mode of the 4 neighbours' labels, smallest label on ties."""
    require_gpu(nn_idx, ref_labels)
    out = torch.empty(nn_idx.shape[0], dtype=I32, device=nn_idx.device)
    call("vidseg_vote4", ptr(nn_idx), ptr(ref_labels), nn_idx.shape[0], ptr(out), stream())
    return out


# --------------------------------------------------------------------------------------
# a16: dense tracking + trajectory vote
# --------------------------------------------------------------------------------------
def dense_tracking(cond16: torch.Tensor, num_frames: int, h: int, w: int, use_aux: bool = True, batch_size: int = 500):
    """feature_extraction.py:326-364 / :176-323 on the conditional half `cond16` fp16 [F, N, C].
    Returns int32 device tensor [F, N] of flat cell indices (h_idx*w + w_idx) plus the number of
    rows whose arg-max needed numpy's tie replay."""
    require_gpu(cond16)
    F, N, C = cond16.shape
    assert F == num_frames and N == h * w
    dev = cond16.device
    nb = N // batch_size + 1
    st = stream()
    normed = torch.empty((nb, F * N, C), dtype=F16, device=dev)
    call("vidseg_track_normalize", ptr(cond16), F * N, C, nb, ptr(normed), st)
    all_idx = torch.empty((F, N), dtype=I32, device=dev)
    all_idx[0] = torch.arange(N, dtype=I32, device=dev)
    blend = torch.empty((N, N), dtype=F16, device=dev)
    ties = torch.zeros(1, dtype=I32, device=dev)
    for f in range(F - 1):
        call("vidseg_track_step", ptr(normed), F, N, w, C, f, batch_size, all_idx[f].data_ptr(), int(use_aux), ptr(blend),
             all_idx[f + 1].data_ptr(), ptr(ties), st)
    return all_idx, ties


def trajectory_vote(all_idx: torch.Tensor, labels: torch.Tensor, w: int, spatial_filter: bool = True) -> torch.Tensor:
    """feature_extraction.py:392-421: signed-jump filter, Counter.most_common vote, last-writer-wins
    write-back.  labels int32 [F, N] -> corrected int32 [F, N]."""
    require_gpu(all_idx, labels)
    F, N = all_idx.shape
    dev = all_idx.device
    common = torch.empty(N, dtype=I32, device=dev)
    winner = torch.empty((F, N), dtype=I32, device=dev)
    out = torch.empty((F, N), dtype=I32, device=dev)
    call("vidseg_trajectory_vote", ptr(all_idx), ptr(labels), F, N, w, int(spatial_filter), ptr(common), ptr(winner), ptr(out),
         stream())
    return out
