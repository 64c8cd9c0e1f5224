"""Oracle of Step 5 (scripts/sampling/process_output.py).  OpenCV is absent here, so the reference module itself cannot be
imported: the wrapping uint8 arithmetic is pinned against numpy evaluating the reference's own expression, the Gaussian blur
only against a direct 2-D evaluation of its documented definition (parity unpinned, see oracle/process_output.py)."""
import numpy as np

from oracle import process_output as PO


def test_uint8_difference_is_the_reference_expression():
    g = np.random.Generator(np.random.PCG64(3))
    a = g.integers(0, 256, (9, 11, 3), dtype=np.uint8)
    b = g.integers(0, 256, (9, 11, 3), dtype=np.uint8)
    ref = np.sqrt(np.sum((a - b) ** 2, axis=2))                      # PO:13 verbatim semantics (uint8 wrap in - and **)
    d = ((a.astype(np.int64) - b.astype(np.int64)) % 256) ** 2 % 256
    assert np.array_equal(ref, np.sqrt(d.sum(axis=2).astype(np.float64)))
    assert ref.dtype == np.float64 and ref.max() <= np.sqrt(3 * 255.0)


def test_gaussian_blur_definition():
    g = np.random.Generator(np.random.PCG64(4))
    d = g.random((13, 10)) * 30
    k = PO.gaussian_kernel_5_3()
    f = np.exp(-((np.arange(5) - 2.0) ** 2) / 18.0)
    assert abs(k.sum() - 1.0) < 1e-15 and np.allclose(k, k[::-1]) and np.allclose(k, f / f.sum(), rtol=1e-15, atol=0)
    p = np.pad(d, 2, mode="reflect")                                 # BORDER_REFLECT_101
    direct = np.zeros_like(d)
    for dy in range(5):
        for dx in range(5):
            direct += k[dy] * k[dx] * p[dy:dy + 13, dx:dx + 10]
    assert np.allclose(PO.gaussian_blur_5_3(d), direct, rtol=0, atol=1e-12)


def test_seg_maps_argmax_and_filter():
    g = np.random.Generator(np.random.PCG64(5))
    K, F, H, W = 3, 2, 12, 16
    pos = g.standard_normal((K, F, 3, H, W)).astype(np.float32)
    neg = pos.copy()
    neg[1, :, :, 2:8, 3:9] += 0.9                                    # mask 1 reacts in a box
    neg[2, :, :, 6:11, 10:15] -= 0.7
    seg, maps = PO.seg_maps(pos, neg, [4, 7, 9])
    assert maps[0].max() == 0 and seg[0, 4, 5] == 7 and seg[0, 8, 12] == 9 and seg[0, 0, 0] == 4      # all-zero column -> first label
    w = np.zeros((K, F, H, W), dtype=np.uint8)
    w[2] = 255
    seg_f, _ = PO.seg_maps(pos, neg, [4, 7, 9], weights=w, filter_s=0.0)
    assert seg_f[0, 8, 12] == 9 and seg_f[0, 4, 5] != 7


def test_gaussian_blur_vs_scipy_separable():
    """An independent implementation of the documented definition: scipy.ndimage.correlate1d along both axes with 'mirror'
    borders (= BORDER_REFLECT_101) and cv2.getGaussianKernel(5, 3)'s formula."""
    from scipy import ndimage
    g = np.random.Generator(np.random.PCG64(6))
    d = np.sqrt(g.integers(0, 3 * 255, (37, 29)).astype(np.float64))
    f = np.exp(-((np.arange(5) - 2.0) ** 2) / (2 * 3.0 ** 2))
    k = f / f.sum()
    ref = ndimage.correlate1d(ndimage.correlate1d(d, k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    got = PO.gaussian_blur_5_3(d)
    assert np.abs(got - ref).max() < 1e-12
    assert np.array_equal(np.clip(got, 0, 255).astype(np.uint8), np.clip(ref, 0, 255).astype(np.uint8))


def test_jpeg_roundtrip_is_pils_codec_and_changes_the_maps():
    """PO:18-19 / 119: the difference maps pass through a JPEG file.  The oracle calls the same PIL codec the reference does."""
    import io
    from PIL import Image
    g = np.random.Generator(np.random.PCG64(7))
    img = (g.random((40, 56)) * 200).astype(np.uint8)
    img[10:20, 10:30] = 250
    back = PO.jpeg_roundtrip(img)
    buf = io.BytesIO()
    Image.fromarray(img).convert("L").save(buf, format="JPEG")
    assert np.array_equal(back, np.array(Image.open(io.BytesIO(buf.getvalue()))))
    assert back.shape == img.shape and back.dtype == np.uint8 and not np.array_equal(back, img)
    K, F, H, W = 2, 1, 24, 32
    pos = g.standard_normal((K, F, 3, H, W)).astype(np.float32)
    neg = pos + g.standard_normal((K, F, 3, H, W)).astype(np.float32) * 0.4
    a, _ = PO.seg_maps(pos, neg, [1, 2])
    b, _ = PO.seg_maps(pos, neg, [1, 2], jpeg=True)
    assert a.shape == b.shape == (F, H, W)
