"""GPU parity of the analysis path (C ABI -> HIP kernels) against the oracle and against the
reference-produced golden fixtures.  Bit-exact for every integer output."""
import glob
import os

import numpy as np
import pytest
import torch

from vidseg_diffusion_amd import synthetic

pytestmark = pytest.mark.gpu

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "analysis_*.npz")))
BLOCKS = ["output_block_8", "output_block_7", "output_block_6"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from vidseg_diffusion_amd import _lib
    _lib.lib()                                  # fail loudly if the HIP extension is missing
    return torch.device("cuda:0")


def _to(dev, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_mean_normalize_bitexact(dev):
    from oracle import analysis as O
    from vidseg_diffusion_amd import analysis as A
    blocks, _ = synthetic.attention_q_dumps(5, 9, 7, 640, num_blocks=3, seed=3)
    F, N = 5, 63
    agg = O.aggregate_blocks(blocks)
    ref = O.normalize_tokens(agg[F:]).reshape(F * N, -1)
    mean16, norm16 = A.mean_normalize([_to(dev, b) for b in blocks], F * N, F * N, want_mean=True)
    assert np.array_equal(mean16.cpu().numpy(), agg[F:].reshape(F * N, -1))
    assert np.array_equal(norm16.cpu().numpy(), ref)


@pytest.mark.parametrize("shape", [(4, 8, 8, 32, 5), (14, 16, 16, 64, 10), (3, 20, 25, 640, 20)])
def test_kmeans_knn_vs_oracle(dev, shape):
    from oracle import analysis as O
    from vidseg_diffusion_amd import analysis as A
    F, h, w, C, K = shape
    N = h * w
    blocks, _ = synthetic.attention_q_dumps(F, h, w, C, num_blocks=1, seed=21)
    feat = O.normalize_tokens(blocks[0][F:]).reshape(F * N, C)
    rs = np.random.RandomState(5)
    centers, labels, inertia = O.kmeans_fit(feat, K, rs)
    km = A.kmeans_fit(_to(dev, feat), K, random_state=np.random.RandomState(5))
    assert np.array_equal(km.labels.cpu().numpy(), labels), "Lloyd labels differ from the oracle"
    np.testing.assert_allclose(km.centers.cpu().numpy(), centers, rtol=0, atol=1e-11)
    assert abs(km.inertia - inertia) <= 1e-9 * abs(inertia)
    pred = A.kmeans_predict(_to(dev, feat[:N]), km.centers).cpu().numpy()
    assert np.array_equal(pred, O.kmeans_predict(feat[:N], centers))
    ref_lab = (pred * 3 + 1).astype(np.int32)                      # arbitrary label values
    knn = A.knn_predict(_to(dev, feat[:N]), _to(dev, ref_lab), _to(dev, feat)).cpu().numpy()
    assert np.array_equal(knn, O.knn_predict(feat[:N], ref_lab, feat))


@pytest.mark.parametrize("shape", [(6, 12, 10, 64), (5, 24, 24, 48), (3, 20, 25, 32), (4, 16, 16, 640)])
def test_tracking_and_vote_vs_oracle(dev, shape):
    from oracle import analysis as O
    from vidseg_diffusion_amd import analysis as A
    F, h, w, C = shape
    N = h * w
    blocks, _ = synthetic.attention_q_dumps(F, h, w, C, num_blocks=1, seed=31)
    th, tw = O.dense_tracking(blocks[0], F, h, w)
    idx, ties = A.dense_tracking(_to(dev, blocks[0][F:]), F, h, w)
    assert np.array_equal(idx.cpu().numpy(), th * w + tw), "tracks differ from the oracle"
    g = np.random.Generator(np.random.PCG64(1))
    labels = g.integers(0, 6, size=(F, h, w)).astype(np.int32)
    labels[:, : h // 2] = 2
    ref, _ = O.correct_low_res_mask(labels, th, tw)
    out = A.trajectory_vote(idx, _to(dev, labels.reshape(F, N)), w).cpu().numpy()
    assert np.array_equal(out.reshape(-1), ref)


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[9:-4] for p in FIXTURES])
def test_feature_extraction_main_matches_reference_golden(dev, path):
    """End to end through the reference's own entry-point signature, against what the reference
    itself produced (tools/gen_golden_analysis.py).  Fixtures f / g are the benchmarked sizes (BASELINE configs[1]:
    14x32x32x640 K=20; configs[2]: 14x36x64x640 K=20), two chained windows each: the K-means / 4-NN labels of window 0 and
    the 14336^2 (32256^2) 4-NN propagation of window 1 are bit-exact; at those sizes the dense tracks differ from the
    reference's torch-CPU fp16 GEMM on ~1 % of the cells (backend-defined accumulation order), which moves a
    handful of corrected labels: there the bar is >= 99.9 % identical, and the chain continues from the reference's labels."""
    from vidseg_diffusion_amd import feature_extraction as FE
    g = np.load(path)
    F, h, w, C, K, seed = (int(g[k]) for k in ("F", "h", "w", "C", "K", "seed"))
    large = F * h * w > 8000
    base, exp = "/nonexistent/vidseg_test", "exp"
    FE.FeatureStore.clear()
    FE.MaskStore.clear()
    ref_mask = ref_fm = ref_ul = None
    for win in range(int(g["windows"])):
        blocks, sha = synthetic.attention_q_dumps(F, h, w, C, num_blocks=3, seed=seed + 100 * win)
        assert sha == str(g[f"w{win}_input_sha256"])
        for name, b in zip(BLOCKS, blocks):
            FE.FeatureStore.put(base, exp, f"{name}_spatial_self_attn_q_time_24", _to(dev, b))
        names = [f"{win:02d}{i:03d}" for i in range(F)]
        gt_path = None
        if bool(g["gt"]) and win == 0:
            import tempfile

            from PIL import Image
            gt_path = os.path.join(tempfile.mkdtemp(), "gt.png")
            Image.fromarray(g["gt_mask_resized"].reshape(h, w).astype(np.uint8)).save(gt_path)
        np.random.seed(seed)
        ul, ref_mask, ref_fm = FE.feature_extraction_main(
            "match_gt_mask", K, 22, ",".join(BLOCKS), exp, exp, "spatial_self_attn_q", h, w, "24", frame_name_list=names,
            base_folder=base, num_frames=F, ref_mask=ref_mask, ref_feature_map=ref_fm, ref_unique_labels=ref_ul,
            gt_mask_path=gt_path)
        if win == 0:
            ref_ul = ul
        assert np.array_equal(ul, g[f"w{win}_unique_labels"])
        assert np.array_equal(ref_mask, g[f"w{win}_match_labels"])
        assert synthetic.sha256_of(ref_fm.cpu().numpy()) == str(g[f"w{win}_ref_feature_sha256"])
        folder = os.path.join(base, exp, "match_gt_mask", "_".join(BLOCKS) + f"_spatial_self_attn_q_masks_{K}")
        _, ref_mask, _ = FE.feature_extraction_main(
            "correct_low_res_mask", K, 22, "output_block_7", exp, exp, "spatial_self_attn_q", h, w, "24",
            frame_name_list=names, base_folder=base, num_frames=F, ref_mask=ref_mask, ref_feature_map=ref_fm,
            ref_unique_labels=ref_ul, gt_mask_path=gt_path, mask_folder=folder)
        if large:
            same = float(np.mean(np.asarray(ref_mask) == g[f"w{win}_corrected_labels"]))
            print(f"{os.path.basename(path)} window {win}: corrected labels identical on {same:.5f}")
            assert same >= 0.999
            ref_mask = g[f"w{win}_corrected_labels"]
        else:
            assert np.array_equal(ref_mask, g[f"w{win}_corrected_labels"])
        if win == 0 and "w0_kmeans_masks_labels" in g.files:
            np.random.seed(seed)
            FE.feature_extraction_main("kmeans_masks", K, 22, "output_block_8", exp, exp, "spatial_self_attn_q", h, w, "24",
                                       frame_name_list=names, base_folder=base, num_frames=F)
            km = FE.MaskStore.get(os.path.join(base, exp, "kmeans_masks", f"output_block_8_spatial_self_attn_q_masks_{K}"))[0]
            assert np.array_equal(km.cpu().numpy(), g["w0_kmeans_masks_labels"])


def test_full_size_c2_vs_oracle(dev):
    """BASELINE config 2 analysis size: 14 frames x 32x32 tokens x 640 channels, K=20."""
    from oracle import analysis as O
    from vidseg_diffusion_amd import analysis as A
    F, h, w, C, K = 14, 32, 32, 640, 20
    N = h * w
    blocks, _ = synthetic.attention_q_dumps(F, h, w, C, num_blocks=3, seed=1)
    agg = O.aggregate_blocks(blocks)
    np.random.seed(1)
    ul, labels, fm = O.match_gt_mask(agg, K, np.random.mtrand._rand)
    _, feat = A.mean_normalize([_to(dev, b) for b in blocks], F * N, F * N)
    assert np.array_equal(feat.cpu().numpy(), fm)
    np.random.seed(1)
    km = A.kmeans_fit(feat, K)
    fake = A.kmeans_predict(feat[:N], km.centers)
    knn = A.knn_predict(feat[:N], fake, feat).cpu().numpy()
    assert np.array_equal(knn, labels)
    th, tw = O.dense_tracking(blocks[1], F, h, w)
    idx, _ = A.dense_tracking(_to(dev, blocks[1][F:]), F, h, w)
    assert np.array_equal(idx.cpu().numpy(), th * w + tw)
    ref, _ = O.correct_low_res_mask(labels.reshape(F, h, w), th, tw)
    out = A.trajectory_vote(idx, _to(dev, labels.reshape(F, N).astype(np.int32)), w).cpu().numpy()
    assert np.array_equal(out.reshape(-1), ref)


def test_empty_cluster_relocation_vs_sklearn_golden(dev):
    """Device-side _relocate_empty_clusters_dense: initial centres that leave 1-3 clusters empty; labels must equal
    scikit-learn's own (tests/golden/kmeans_relocate.npz) and the oracle's, centres within float64 rounding."""
    from oracle import analysis as OA
    from vidseg_diffusion_amd import analysis as A
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "kmeans_relocate.npz"))
    for nm in "abc":
        init, X = g[f"{nm}_init"], g[f"{nm}_X"]
        res = A.kmeans_fit(torch.from_numpy(X).to(dev), init.shape[0], random_state=np.random.RandomState(0), init=init)
        labels = res.labels.cpu().numpy()
        assert np.array_equal(labels, g[f"{nm}_labels"]), nm
        assert np.abs(res.centers.cpu().numpy() - g[f"{nm}_centers"]).max() < 1e-12, nm
        oc, ol, _ = OA.kmeans_fit(X, init.shape[0], np.random.RandomState(0), init=init)
        assert np.array_equal(labels, ol) and res.n_iter == int(g[f"{nm}_n_iter"])


def test_edge_cases_vs_oracle(dev):
    """Degenerate and ragged inputs the reference's third-party code defines behaviour for: n_samples < n_clusters (sklearn's
    ValueError), K = 1, n = K (every sample its own cluster), duplicated rows (zero distances), a single-frame window (no
    tracking pairs), a token count that is not a multiple of any tile size, empty query sets, wrong dtypes / CPU tensors."""
    from oracle import analysis as O
    from vidseg_diffusion_amd import _lib, analysis as A
    g = np.random.Generator(np.random.PCG64(123))
    X = (g.standard_normal((37, 24)) * 0.3).astype(np.float16)
    with pytest.raises(ValueError, match="should be >= n_clusters"):
        A.kmeans_fit(_to(dev, X[:3]), 5, random_state=np.random.RandomState(0))
    with pytest.raises(_lib.VidsegError):
        A.kmeans_fit(torch.from_numpy(X), 3)                                       # CPU tensor: no fallback
    with pytest.raises(_lib.VidsegError):
        A.kmeans_fit(_to(dev, X.astype(np.float32)), 3)                            # not fp16
    for K in (1, 37, 6):                                                          # K = 1, n = K, ragged n
        oc, ol, oi = O.kmeans_fit(X, K, np.random.RandomState(3))
        km = A.kmeans_fit(_to(dev, X), K, random_state=np.random.RandomState(3))
        assert np.array_equal(km.labels.cpu().numpy(), ol), K
        np.testing.assert_allclose(km.centers.cpu().numpy(), oc, rtol=0, atol=1e-11)
    Xd = np.repeat(X[:9], 5, axis=0)                                              # duplicates: 9 distinct rows, 45 samples
    oc, ol, _ = O.kmeans_fit(Xd, 4, np.random.RandomState(1))
    km = A.kmeans_fit(_to(dev, Xd), 4, random_state=np.random.RandomState(1))
    assert np.array_equal(km.labels.cpu().numpy(), ol)
    # 4-NN with fewer distinct neighbours than k and an empty query set
    ref, lab = X[:5], np.array([3, 1, 3, 0, 1], dtype=np.int32)
    assert np.array_equal(A.knn_predict(_to(dev, ref), _to(dev, lab), _to(dev, X)).cpu().numpy(), O.knn_predict(ref, lab, X))
    assert A.knn_predict(_to(dev, ref), _to(dev, lab), _to(dev, X[:0])).numel() == 0
    # single-frame window: dense tracking has no pairs, the vote leaves the labels alone
    blocks, _ = synthetic.attention_q_dumps(1, 5, 7, 32, num_blocks=1, seed=8)
    q = blocks[0][1:]                                                             # conditional half, F = 1
    tr, _ = A.dense_tracking(_to(dev, q), 1, 5, 7)
    th, tw = O.dense_tracking(q, 1, 5, 7)
    assert tr.shape == (1, 35) and np.array_equal(tr.cpu().numpy()[0] // 7, th[0]) and np.array_equal(tr.cpu().numpy()[0] % 7, tw[0])
    labs = g.integers(0, 3, size=(1, 35)).astype(np.int32)
    out = A.trajectory_vote(tr, _to(dev, labs), 7).cpu().numpy()
    corr, _ = O.correct_low_res_mask(labs.reshape(1, 5, 7).astype(np.int64), th, tw)
    assert np.array_equal(out.reshape(-1), corr)


def test_kmeans_is_bit_stable_beside_a_busy_stream(dev):
    """K-means on a side stream while the main stream keeps every CU busy (how the window pipeline runs it): every restart's seeds,
    iteration count, inertia and labels must be the same run after run.  k-means++ once kept the candidates of round c and of round
    c - 1 in one buffer; in round 1 (one candidate per restart in, T out) restart 0's block overwrote entries that later-starting
    blocks had yet to read -- invisible on an idle chip where all R blocks start together, 1 run in ~250 wrong beside a busy stream."""
    from vidseg_diffusion_amd import analysis as A, ops
    F, h, w, C, K = 14, 32, 32, 640, 20
    blocks, _ = synthetic.attention_q_dumps(F, h, w, C, num_blocks=3, seed=7)
    _, feat = A.mean_normalize([torch.from_numpy(b).to(dev) for b in blocks], F * h * w, F * h * w)
    ad = ops.act_dtype()
    x0 = torch.randn(28, 64, 64, 320, device=dev).to(ad)
    wc = ops.pack_conv3x3(torch.randn(320, 320, 3, 3) * 0.02, dev)
    bc = torch.zeros(320, device=dev)
    side = torch.cuda.Stream()
    first = None
    for it in range(300 if os.environ.get("VIDSEG_SLOW_TESTS") else 40):   # ~1 wrong run in 250 before the fix: the long form is the hunt
        for _ in range(60):
            ops.conv3x3(x0, wc, bc)                                  # full-chip launches queued on the main stream
        with torch.cuda.stream(side):
            np.random.seed(17)
            km = A.kmeans_fit(feat, K)
            sig = (tuple(km.all_n_iter), tuple(float(v) for v in km.all_inertia), A.LAST_CENTER_IDS.cpu().numpy().tobytes(),
                   km.all_labels.cpu().numpy().tobytes())
        torch.cuda.synchronize()
        if first is None:
            first = sig
        assert sig[0] == first[0] and sig[1] == first[1] and sig[2] == first[2] and sig[3] == first[3], \
            f"run {it}: K-means differs from run 0 (n_iter {sig[0]} vs {first[0]})"
