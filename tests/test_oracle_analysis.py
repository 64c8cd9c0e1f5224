"""Pin the analysis oracle (oracle/analysis.py) against the fixtures the REFERENCE produced
(tools/gen_golden_analysis.py ran scripts/sampling/feature_extraction.py from /root/reference).

Bit-exact for every integer output except the dense-tracking arg-max, where the reference's
fp16 GEMM has a backend-defined fp32 accumulation order (see oracle/analysis.py header): there
we require >= 99 % identical track cells on the small fixtures and, given the reference's own tracks,
bit-exact filter/vote/write-back.  On the two fixtures at the benchmarked sizes the agreement is lower
(measured 99.0 / 98.96 % of the cells at 14x32x32x640, 95.6 / 97.3 % at 14x36x64x640: fp16 cosines near 1
are 2^-11 apart, so with more cells per frame more neighbours tie, and a flipped cell changes the rest of
its 13-frame trajectory); the bar there is >= 95 % of the cells and >= 99.9 % of the corrected LABELS
(measured 99.93-100 %: the vote is robust to where a trajectory wanders inside an object).  The reference's
cosine GEMM is torch's CPU Half matmul -- on this build host oneDNN on AVX512-FP16 -- whose result differs
from fp16(exact dot), fp16(fp32 sgemm) and fp16(sequential fp32) alike on 0.08-0.17 % of the entries, i.e.
it is a property of the host CPU, not of the algorithm (the reference itself hard-codes device="cuda").

The two fixtures at the benchmarked sizes (f: BASELINE configs[1] 14x32x32x640 K=20, g: configs[2] 14x36x64x640 K=20, two
chained windows each) cost the numpy oracle minutes of float64 K-means; the CPU suite checks their aggregation, tracking and
vote stages and runs the K-means / 4-NN part only with VIDSEG_SLOW_TESTS=1 (measured: bit-exact, 200 s for f on 8 cores).
The HIP path is compared with those reference labels directly on the GPU (tests/test_gpu_analysis.py).
"""
import glob
import os

import numpy as np
import pytest

from oracle import analysis as A
from vidseg_diffusion_amd import synthetic

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "analysis_*.npz")))


def _load(path):
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[9:-4] for p in FIXTURES])
def test_oracle_matches_reference(path):
    g = _load(path)
    F, h, w, C, K, seed = (int(g[k]) for k in ("F", "h", "w", "C", "K", "seed"))
    large = F * h * w > 8000
    slow = os.environ.get("VIDSEG_SLOW_TESTS", "0") == "1"
    ref_mask = ref_fm = None
    for win in range(int(g["windows"])):
        if large and not slow and win > 0:
            break                                                 # the second window repeats the same stages
        blocks, sha = synthetic.attention_q_dumps(F, h, w, C, num_blocks=3, seed=seed + 100 * win)
        assert sha == str(g[f"w{win}_input_sha256"]), "synthetic generator drifted"
        agg = A.aggregate_blocks(blocks)
        if large and not slow:
            fm = A.normalize_tokens(agg[F:]).reshape(F * h * w, C)
            assert synthetic.sha256_of(fm) == str(g[f"w{win}_ref_feature_sha256"])
            labels = g[f"w{win}_match_labels"]
        else:
            np.random.seed(seed)
            rs = np.random.mtrand._rand                          # sklearn check_random_state(None)
            gt = g["gt_mask_resized"] if (bool(g["gt"]) and win == 0) else None
            ul, labels, fm = A.match_gt_mask(agg, K, rs, ref_mask=ref_mask, ref_feature_map=ref_fm, gt_mask=gt)
            assert np.array_equal(ul, g[f"w{win}_unique_labels"])
            assert synthetic.sha256_of(fm) == str(g[f"w{win}_ref_feature_sha256"])
            assert np.array_equal(labels, g[f"w{win}_match_labels"]), "K-means/KNN labels differ from reference"

        th, tw = A.dense_tracking(blocks[1], F, h, w)             # block 7 only (SDP:399-400)
        rh, rw = g[f"w{win}_track_h"].astype(np.int64), g[f"w{win}_track_w"].astype(np.int64)
        same = np.mean((th == rh) & (tw == rw))
        assert same >= (0.95 if large else 0.99), f"tracks agree on only {same:.4f}"
        # integer stages, given the reference's own tracks: bit-exact
        corr, _ = A.correct_low_res_mask(labels.reshape(F, h, w), rh, rw)
        assert np.array_equal(corr, g[f"w{win}_corrected_labels"])
        # and end-to-end with the oracle's tracks: IoU-level agreement
        corr2, _ = A.correct_low_res_mask(labels.reshape(F, h, w), th, tw)
        assert np.mean(corr2 == g[f"w{win}_corrected_labels"]) >= (0.999 if large else 0.99)
        ref_mask, ref_fm = g[f"w{win}_corrected_labels"], fm
        if win == 0 and "w0_kmeans_masks_labels" in g and not (large and not slow):
            np.random.seed(seed)
            km = A.kmeans_masks(blocks[0], K, np.random.mtrand._rand)
            assert np.array_equal(km, g["w0_kmeans_masks_labels"])


def test_npselect_port_matches_numpy_on_ties():
    """oracle/npselect.py replays numpy's fp16 arg-introselect: compare with numpy itself."""
    from oracle.npselect import argpartition_last
    g = np.random.Generator(np.random.PCG64(0))
    for n in [2, 3, 5, 7, 16, 64, 120, 256, 500, 576, 1024]:
        for trial in range(12):
            kind = trial % 3
            if kind == 0:
                x = g.integers(0, 4, size=n).astype(np.float16)
            elif kind == 1:
                x = g.standard_normal(n).astype(np.float16)
                x[g.integers(0, n, size=max(2, n // 8))] = x.max()
            else:
                x = (np.round(g.uniform(0.9, 1.0, size=n) * 2048) / 2048).astype(np.float16)
            assert argpartition_last(x) == int(np.argpartition(x, -1)[-1:][0])


def test_empty_cluster_relocation_matches_sklearn():
    """_relocate_empty_clusters_dense + _average_centers (cluster/_k_means_common.pyx:167-211, 274-295): initial centres
    that leave 1-3 clusters empty; golden = scikit-learn's own KMeans(init=ndarray) (tools/gen_golden_kmeans_relocate.py)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "kmeans_relocate.npz"))
    for nm in "abc":
        init = g[f"{nm}_init"]
        centers, labels, _ = A.kmeans_fit(g[f"{nm}_X"], init.shape[0], np.random.RandomState(0), init=init)
        assert np.array_equal(labels, g[f"{nm}_labels"]), nm
        assert np.abs(centers - g[f"{nm}_centers"]).max() < 1e-12, nm
