"""Oracle (CPU restatement) of the post-UNet analysis path -- test infrastructure only.

Restates, in numpy, what the reference does between the dumped attention queries and the
per-frame cluster-id masks:

    scripts/sampling/feature_extraction.py  (FE)
      feature_extraction_main  FE:670-795   -> aggregate_blocks + dispatch
      match_gt_mask            FE:546-643   -> match_gt_mask
      save_inidividual_masks_kmeans FE:30-113 -> kmeans_masks
      dense_feature_matching_iterative FE:176-323, dense_tracking FE:326-364 -> dense_tracking
      correct_low_res_mask     FE:367-461   -> correct_low_res_mask

plus the third-party arithmetic those call (scikit-learn 1.7.2 ``KMeans`` /
``KNeighborsClassifier``; file:line below are in site-packages/sklearn).  The file hand-off
(.pt dumps, PNG mask folders) is replaced by arrays: with ``ref_unique_labels`` covering every
label the PNG round trip FE:380-389/500-521 is the identity (masks are written at feature
resolution, ``resize`` to the same size is a no-op, arg-max over one-hot masks returns the label).

Pinned by tests/golden/analysis_*.npz (tools/gen_golden_analysis.py runs the reference).
Known, documented deviation: the reference evaluates the cosine maps with an fp16 GEMM whose
fp32 accumulation order is backend-defined (torch CPU/MKL here, cuBLAS in production), so
1-ulp fp16 differences (~0.15 % of entries, measured) can move an arg-max between near-tied
cells.  The oracle uses the exact dot product rounded once to fp16; everything else is
bit-faithful.
"""
from __future__ import annotations

from collections import Counter

import numpy as np

from .npselect import argpartition_last

F16 = np.float16
F32 = np.float32
F64 = np.float64


# --------------------------------------------------------------------------------------
# a13: aggregation across decoder blocks  (FE:739-748)
# --------------------------------------------------------------------------------------
def aggregate_blocks(blocks):
    """torch.mean(torch.stack(blocks), dim=0) on fp16 tensors (FE:745).

    torch reduces Half with fp32 accumulation and rounds once (verified bit-exact against
    torch 2.10 CPU on 2e5 random triples): f16((a+b+c in f32) / n).
    """
    acc = blocks[0].astype(F32)
    for b in blocks[1:]:
        acc = acc + b.astype(F32)
    return (acc / F32(len(blocks))).astype(F16)


def normalize_tokens(feat):
    """feature_maps / np.max(np.abs(feature_maps), axis=-1, keepdims=True) in fp16 numpy
    (FE:554-555, FE:38-39).  numpy evaluates half division in fp32 and rounds once."""
    m = np.max(np.abs(feat), axis=-1, keepdims=True)
    return (feat.astype(F32) / m.astype(F32)).astype(F16)


# --------------------------------------------------------------------------------------
# scikit-learn KMeans (cluster/_kmeans.py), float64 on fp16-valued data
# --------------------------------------------------------------------------------------
def _row_norms_sq(X):
    return np.einsum("ij,ij->i", X, X)            # utils/extmath.py row_norms


def _euclidean_sq(A, X, x_sq):
    """metrics/pairwise.py:_euclidean_distances(A, X, Y_norm_squared=x_sq, squared=True)."""
    d = -2.0 * (A @ X.T)
    d += _row_norms_sq(A)[:, None]
    d += x_sq[None, :]
    np.maximum(d, 0, out=d)
    return d


def kmeans_plusplus(X, n_clusters, x_sq, random_state):
    """cluster/_kmeans.py:174-274 with unit sample weights.  Consumes the legacy RandomState
    stream exactly like sklearn: one random_sample() for `choice`, then
    uniform(size=2+int(log K)) per further centre."""
    n = X.shape[0]
    n_local_trials = 2 + int(np.log(n_clusters))
    centers = np.empty((n_clusters, X.shape[1]), dtype=X.dtype)
    indices = np.full(n_clusters, -1, dtype=np.int64)
    # RandomState.choice(n, p=w/w.sum()): cdf = p.cumsum(); cdf /= cdf[-1]; searchsorted(side="right")
    p = np.ones(n, dtype=F64) / F64(n)
    cdf = p.cumsum()
    cdf /= cdf[-1]
    center_id = int(cdf.searchsorted(random_state.random_sample(), side="right"))
    centers[0] = X[center_id]
    indices[0] = center_id
    closest = _euclidean_sq(centers[0, None], X, x_sq)              # [1, n]
    pot = float((closest @ np.ones(n, dtype=F64))[0])
    for c in range(1, n_clusters):
        rand_vals = random_state.uniform(size=n_local_trials) * pot
        cand = np.searchsorted(np.cumsum(closest.ravel(), dtype=F64), rand_vals)
        np.clip(cand, None, n - 1, out=cand)
        d = _euclidean_sq(X[cand], X, x_sq)
        np.minimum(closest, d, out=d)
        pots = d @ np.ones((n, 1), dtype=F64)
        best = int(np.argmin(pots))
        pot = float(pots[best, 0])
        closest = d[best][None]
        centers[c] = X[cand[best]]
        indices[c] = cand[best]
    return centers, indices


def _assign(X, centers):
    """E-step of cluster/_k_means_lloyd.pyx:_update_chunk_dense: argmin_j (|c_j|^2 - 2 x.c_j),
    first minimum wins."""
    d = -2.0 * (X @ centers.T)
    d += _row_norms_sq(centers)[None, :]
    return np.argmin(d, axis=1).astype(np.int32)


def _relocate_empty(X, centers_old, centers_new, weight, labels):
    """cluster/_k_means_common.pyx:167-211."""
    empty = np.where(weight == 0)[0]
    if empty.size == 0:
        return
    dist = ((X - centers_old[labels]) ** 2).sum(axis=1)
    far = np.argpartition(dist, -empty.size)[: -empty.size - 1 : -1]
    if dist.max() == 0:
        return
    for idx, new_id in enumerate(empty):
        f = far[idx]
        old_id = labels[f]
        centers_new[old_id] -= X[f]
        centers_new[new_id] = X[f]
        weight[new_id] = 1.0
        weight[old_id] -= 1.0


def lloyd_single(X, centers_init, max_iter, tol, raw=None, mean=None):
    """cluster/_kmeans.py:_kmeans_single_lloyd (:683-752).  Returns labels, inertia, centers, n_iter.

    M-step sums: sklearn adds the centred float64 rows chunk by chunk, thread by thread (_k_means_lloyd.pyx:140-170),
    so its last bits depend on the OpenMP thread count.  With `raw` (the un-centred fp16-valued rows) and `mean` the
    sums are taken over the RAW values -- float64 sums of fp16 values are exact, hence order-free -- and centred
    afterwards: sum_k - count_k * mean.  This is the definition the HIP path implements (vidseg_lloyd_step)."""
    K = centers_init.shape[0]
    centers = centers_init.copy()
    labels_old = np.full(X.shape[0], -1, dtype=np.int32)
    strict = False
    it = 0
    for it in range(max_iter):
        labels = _assign(X, centers)
        onehot = np.zeros((X.shape[0], K), dtype=F64)
        onehot[np.arange(X.shape[0]), labels] = 1.0
        weight = onehot.sum(axis=0)
        if raw is not None:
            centers_new = onehot.T @ raw - weight[:, None] * mean[None, :]
        else:
            centers_new = onehot.T @ X
        _relocate_empty(X, centers, centers_new, weight, labels)
        # _average_centers (_k_means_common.pyx:274-295): centers *= 1/weight; a cluster that
        # is still empty (only when relocation bailed out) sits on the biggest cluster.
        amax = int(np.argmax(weight))
        for j in range(K):                                   # in-place loop order matters for a still-empty cluster:
            if weight[j] > 0:                                # it copies the biggest cluster's row as it is at that moment
                centers_new[j] *= 1.0 / weight[j]            # (averaged only if amax < j), _k_means_common.pyx:286-295
            else:
                centers_new[j] = centers_new[amax]
        # _center_shift: per-cluster Euclidean norm, then (shift**2).sum() in _kmeans.py:725
        shift = np.sqrt(((centers_new - centers) ** 2).sum(axis=1))
        shift_tot = (shift ** 2).sum()
        centers = centers_new
        if np.array_equal(labels, labels_old):
            strict = True
            break
        if shift_tot <= tol:
            break
        labels_old = labels
    if not strict:
        labels = _assign(X, centers)
    inertia = float(((X - centers[labels]) ** 2).sum())
    return labels, inertia, centers, it + 1


def _is_same_clustering(l1, l2, K):
    mapping = np.full(K, -1, dtype=np.int64)
    for a, b in zip(l1.tolist(), l2.tolist()):
        if mapping[a] == -1:
            mapping[a] = b
        elif mapping[a] != b:
            return False
    return True


def kmeans_fit(X16, n_clusters, random_state, n_init=10, max_iter=300, tol=1e-4, init=None):
    """KMeans(n_clusters, n_init=10).fit(X) (cluster/_kmeans.py:1451-1547): fp16 input is
    up-cast to float64 by validate_data(dtype=[float64, float32]), mean-centred, tolerance
    scaled by the mean per-feature variance, best-of-n_init by strictly lower inertia unless
    the clustering is identical up to a permutation.  Returns (centers, labels, inertia)."""
    raw = np.ascontiguousarray(X16, dtype=F64)
    mean = raw.mean(axis=0)
    X = raw - mean
    tol_ = float(np.mean(np.var(X, axis=0)) * tol)
    x_sq = _row_norms_sq(X)
    best = None
    if init is not None:                                     # KMeans(init=ndarray): one run from the given centres (:1466-1474, :1490)
        n_init = 1
    for _ in range(n_init):
        if init is not None:
            c0 = np.ascontiguousarray(init, dtype=F64) - mean
        else:
            c0, _idx = kmeans_plusplus(X, n_clusters, x_sq, random_state)
        labels, inertia, centers, n_iter = lloyd_single(X, c0, max_iter, tol_, raw=raw, mean=mean)
        if best is None or (inertia < best[1] and not _is_same_clustering(labels, best[0], n_clusters)):
            best = (labels, inertia, centers, n_iter)
    labels, inertia, centers, _ = best
    return centers + mean, labels, inertia


def kmeans_predict(X16, centers):
    """KMeans.predict: E-step on the *uncentred* float64 data with cluster_centers_."""
    return _assign(np.ascontiguousarray(X16, dtype=F64), centers)


# --------------------------------------------------------------------------------------
# scikit-learn KNeighborsClassifier(n_neighbors=4), brute force on fp16 storage
# --------------------------------------------------------------------------------------
def knn_top4(ref_X16, query16, k=4, chunk=4096):
    """kneighbors(return_distance=False): indices of the k nearest reference rows per query.  fp16 storage disables
    the ArgKmin fast path, so sklearn goes through pairwise_distances_chunked -> euclidean_distances: both sides
    up-cast to float64, d = |x|^2 - 2 x.y + |y|^2 clipped at 0, then argpartition(kth=k-1)[:, :k]."""
    Y = np.ascontiguousarray(ref_X16, dtype=F64)
    yy = _row_norms_sq(Y)
    out = np.empty((query16.shape[0], k), dtype=np.int64)
    for s in range(0, query16.shape[0], chunk):
        Xq = np.ascontiguousarray(query16[s:s + chunk], dtype=F64)
        d = -2.0 * (Xq @ Y.T)
        d += _row_norms_sq(Xq)[:, None]
        d += yy[None, :]
        np.maximum(d, 0, out=d)
        out[s:s + chunk] = np.argpartition(d, k - 1, axis=1)[:, :k]
    return out


def vote4(nn_idx, ref_y):
    """neighbors/_classification.py predict with uniform weights: mode of the neighbours' classes,
    smallest class among the most frequent."""
    classes, y_idx = np.unique(np.asarray(ref_y), return_inverse=True)
    votes = y_idx[nn_idx]
    counts = np.zeros((votes.shape[0], classes.size), dtype=np.int32)
    np.add.at(counts, (np.arange(votes.shape[0])[:, None], votes), 1)
    return classes[np.argmax(counts, axis=1)]                     # first max == smallest class


def knn_predict(ref_X16, ref_y, query16, k=4, chunk=4096):
    """KNeighborsClassifier(n_neighbors=4).fit(ref, ref_y).predict(query) (feature_extraction.py:608-613)."""
    return vote4(knn_top4(ref_X16, query16, k, chunk), ref_y)


# --------------------------------------------------------------------------------------
# a14 / a15
# --------------------------------------------------------------------------------------
def match_gt_mask(feature_maps16, num_masks, random_state, ref_mask=None, ref_feature_map=None,
                  ref_unique_labels=None, gt_mask=None, use_gt_mask=False):
    """FE:546-643 without the PNG side effects.

    feature_maps16: fp16 [2F, N, C] (aggregated dump).  gt_mask: optional int [N] already
    NEAREST-resized to the feature grid (FE:581-583).
    Returns (unique_labels, ref_mask int64 [F*N], ref_feature_map fp16 [F*N, C]).
    """
    F = feature_maps16.shape[0] // 2
    feat = feature_maps16[F:]
    if feat.shape[-1] > 1:
        feat = normalize_tokens(feat)
    flat = feat.reshape(-1, feat.shape[-1])
    if ref_mask is None:
        centers, _, _ = kmeans_fit(flat, num_masks, random_state)
        fake = kmeans_predict(feat[0], centers)
        mask_np = fake if gt_mask is None else np.asarray(gt_mask).reshape(-1)
        if not use_gt_mask:
            ref_mask = np.zeros(fake.shape[0], dtype=np.int64)
            for lab in np.unique(fake):
                vals, cnt = np.unique(mask_np[fake == lab], return_counts=True)
                ref_mask[fake == lab] = vals[np.argmax(cnt)]
        else:
            ref_mask = mask_np
        ref_feature_map = feat[0]
    unique_labels = np.unique(ref_mask)
    labels = knn_predict(ref_feature_map, ref_mask, flat, k=4)
    return unique_labels, labels.astype(np.int64), flat


def kmeans_masks(feature_maps16, num_clusters, random_state):
    """FE:30-113 (mode "kmeans_masks", spatial): fit + predict on all conditional tokens."""
    F = feature_maps16.shape[0] // 2
    feat = normalize_tokens(feature_maps16)[F:]                  # FE:39 normalises before the split
    flat = feat.reshape(-1, feat.shape[-1])
    centers, _, _ = kmeans_fit(flat, num_clusters, random_state)
    return kmeans_predict(flat, centers).reshape(F, -1)


# --------------------------------------------------------------------------------------
# a16: dense tracking + trajectory vote
# --------------------------------------------------------------------------------------
def _l2norm_rows16(x16):
    """x / torch.norm(x, dim=-1, keepdim=True) on fp16 (FE:272-274): norm with wide
    accumulation rounded to fp16, division in fp32 rounded to fp16."""
    n = np.sqrt((x16.astype(F64) ** 2).sum(axis=-1, keepdims=True)).astype(F16)
    return (x16.astype(F32) / n.astype(F32)).astype(F16)


def _half_mul(scalar, arr16):
    """python-float * fp16 ndarray under NEP 50: scalar is cast to fp16, product in fp32, one rounding."""
    return (F32(F16(scalar)) * arr16.astype(F32)).astype(F16)


def dense_tracking(feature_maps16, num_frames, h, w, use_aux=True, batch_size=500):
    """FE:326-364 -> FE:176-323 with mask_path=None, attn_type="spatial", top_k=1.

    Every grid cell of frame 0 is tracked forward; returns int64 (all_h, all_w) of shape
    [F, N].  Quirks kept: query batches of 500 with num_batches = N//500 + 1 (FE:196-197);
    the target and aux (frame-0) maps are re-normalised *inside* the batch loop
    (FE:272-274) so batch b sees them normalised b+1 times; fp16 blend
    f/(f+1)*cos + 1/(f+1)*cos_aux (FE:290-291); arg-max = np.argpartition(x, -1)[-1:]
    on an fp16 row: a unique maximum is returned as is, and tied maxima are resolved by
    replaying numpy's scalar arg-introselect (oracle/npselect.py).
    """
    N = h * w
    cond = feature_maps16[num_frames:]
    cur = np.arange(N, dtype=np.int64)                           # flat index = h_idx*w + w_idx
    all_idx = [cur.copy()]
    nb = N // batch_size + 1
    for f in range(num_frames - 1):
        src = cond[f]
        trg = cond[f + 1]
        aux = cond[0]
        nxt = np.empty(N, dtype=np.int64)
        for b in range(nb):
            q = cur[b * batch_size:(b + 1) * batch_size]
            trg = _l2norm_rows16(trg)
            aux = _l2norm_rows16(aux)
            if q.size == 0:
                continue
            s = _l2norm_rows16(src[q])
            cos = (s.astype(F64) @ trg.astype(F64).T).astype(F16)
            cos_aux = (s.astype(F64) @ aux.astype(F64).T).astype(F16)
            if use_aux:
                a = _half_mul(f / (f + 1), cos)
                c = _half_mul(1 / (f + 1), cos_aux)
                cos = (a.astype(F32) + c.astype(F32)).astype(F16)
            am = np.argmax(cos, axis=1)
            ties = np.nonzero((cos == cos.max(axis=1, keepdims=True)).sum(axis=1) > 1)[0]
            for r in ties:
                am[r] = argpartition_last(cos[r])
            nxt[b * batch_size:b * batch_size + q.size] = am
        cur = nxt
        all_idx.append(cur.copy())
    all_idx = np.stack(all_idx)                                   # [F, N]
    return all_idx // w, all_idx % w


def correct_low_res_mask(labels, all_h, all_w, spatial_filter=True):
    """FE:367-461 after the tracking, on label maps `labels` int [F, h, w].

    Spatial filter (FE:392-409): a trajectory is dropped as soon as a *signed* step in h or w
    between consecutive frames exceeds 1.  Vote (FE:411-421): Counter.most_common(1) over the
    ORIGINAL labels along the trajectory (ties -> label met first), written back along the
    trajectory into the new maps, points processed in ascending order so the last writer wins.
    Returns int64 [F*h*w].
    """
    ori = np.asarray(labels)
    new = ori.copy()
    F = ori.shape[0]
    H = np.asarray(all_h)
    W = np.asarray(all_w)
    keep = np.ones(H.shape[1], dtype=bool)
    if spatial_filter:
        dh = H[1:] - H[:-1]
        dw = W[1:] - W[:-1]
        keep = ~np.any((dh > 1) | (dw > 1), axis=0)
    for p in np.nonzero(keep)[0]:
        th, tw = H[:, p], W[:, p]
        labs = [ori[t, th[t], tw[t]] for t in range(F)]
        common = Counter(labs).most_common(1)[0][0]
        for t in range(F):
            new[t, th[t], tw[t]] = common
    return new.reshape(-1).astype(np.int64), keep
