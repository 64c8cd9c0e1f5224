"""Step 5 on the device vs the oracle (bit-exact: byte / integer work plus a float64 blur evaluated in the same order)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(seed, K, F, H, W):
    g = np.random.Generator(np.random.PCG64(seed))
    pos = (g.standard_normal((K, F, 3, H, W)) * 0.8).astype(np.float32)
    neg = (pos + g.standard_normal((K, F, 3, H, W)).astype(np.float32) * g.random((K, 1, 1, 1, 1)).astype(np.float32) * 0.6).astype(np.float32)
    neg[0] = pos[0]                                                   # a mask with no reaction at all: all-zero map, max = 0
    return pos, neg


@pytest.mark.parametrize("K,F,H,W", [(3, 2, 16, 24), (5, 3, 33, 47), (2, 1, 64, 64), (2, 2, 3, 5), (2, 3, 70, 130)])
def test_seg_map_vs_oracle(K, F, H, W):
    from oracle import process_output as OPO
    from vidseg_diffusion_amd import process_output as PO
    dev = torch.device("cuda:0")
    pos, neg = _case(K * 100 + H, K, F, H, W)
    labels = [3, 5, 8, 11, 12][:K]
    ref_seg, ref_maps = OPO.seg_maps(pos, neg, labels)
    decoded = {}
    for k, lab in enumerate(labels):
        decoded[(1.0, lab)] = torch.from_numpy(pos[k]).to(dev)
        decoded[(-1.0, lab)] = torch.from_numpy(neg[k]).to(dev)
    for k, lab in enumerate(labels):
        m, mx = PO.difference_map(decoded[(1.0, lab)], decoded[(-1.0, lab)])
        assert np.array_equal(m.cpu().numpy(), ref_maps[k]), f"difference map of mask {lab}"
        assert np.array_equal(mx.cpu().numpy(), ref_maps[k].reshape(F, -1).max(axis=1))
    seg = PO.get_seg_map(decoded, labels)
    assert np.array_equal(seg.cpu().numpy(), ref_seg)
    # filtered variant (PO:31-40) with the Step 3 label maps resized by PIL like the reference's mask PNGs
    g = np.random.Generator(np.random.PCG64(9))
    label_maps = np.asarray(labels)[g.integers(0, K, (F, H // 8 + 1, W // 8 + 1))]
    w = PO.mask_weights(label_maps, labels, (H, W))
    ref_f, _ = OPO.seg_maps(pos, neg, labels, weights=w, filter_s=0.7)
    seg_f = PO.get_seg_map(decoded, labels, label_maps=label_maps, filter_difference=True, filter_s=0.7)
    assert np.array_equal(seg_f.cpu().numpy(), ref_f)


def test_decode_to_seg_map_end_to_end():
    """Steps 4b-5 on the narrow first stage: decode two latents per mask, difference, arg-max -- device vs oracle on the device's
    own decoded frames (the decode itself is covered by tests/test_gpu_vae.py)."""
    import os
    from oracle import process_output as OPO
    from tests.test_oracle_vae import narrow_decoder_state_dict
    from vidseg_diffusion_amd import process_output as PO
    from vidseg_diffusion_amd.vae import decode_first_stage
    dev = torch.device("cuda:0")
    gz = np.load(os.path.join(os.path.dirname(__file__), "golden", "vae_decoder_narrow.npz"))
    net, _, sd = narrow_decoder_state_dict(gz["pq_bias"])
    net.load_state_dict(sd)
    g = np.random.Generator(np.random.PCG64(21))
    labels, decoded = [0, 2, 3], {}
    for lab in labels:
        z = (g.standard_normal((2, 4, 8, 8)) * 0.7).astype(np.float32)
        dz = np.zeros_like(z)
        dz[:, :, 2 * (lab % 3):2 * (lab % 3) + 3, 1 + lab:4 + lab] = 0.5
        decoded[(1.0, lab)] = decode_first_stage(net, torch.from_numpy(z + dz).to(dev), 0.18215)
        decoded[(-1.0, lab)] = decode_first_stage(net, torch.from_numpy(z - dz).to(dev), 0.18215)
    seg = PO.get_seg_map(decoded, labels).cpu().numpy()
    pos = np.stack([decoded[(1.0, l)].cpu().numpy() for l in labels])
    neg = np.stack([decoded[(-1.0, l)].cpu().numpy() for l in labels])
    ref, _ = OPO.seg_maps(pos, neg, labels)
    assert seg.shape == (2, 64, 64) and np.array_equal(seg, ref)
    assert len(np.unique(seg)) > 1


def test_jpeg_compat_mode_vs_oracle():
    """process_output.py:18-19, 119: with jpeg_compat the difference maps take the reference's JPEG save / re-load (PIL's codec on
    both sides) before they are normalised: device == oracle bit for bit, and the maxima are those of the re-loaded images."""
    from oracle import process_output as OPO
    from vidseg_diffusion_amd import process_output as PO
    dev = torch.device("cuda:0")
    K, F, H, W = 4, 2, 48, 40
    pos, neg = _case(77, K, F, H, W)
    labels = [1, 4, 6, 9]
    decoded = {}
    for k, lab in enumerate(labels):
        decoded[(1.0, lab)] = torch.from_numpy(pos[k]).to(dev)
        decoded[(-1.0, lab)] = torch.from_numpy(neg[k]).to(dev)
    ref_j, _ = OPO.seg_maps(pos, neg, labels, jpeg=True)
    seg_j = PO.get_seg_map(decoded, labels, jpeg_compat=True).cpu().numpy()
    assert np.array_equal(seg_j, ref_j)
    m, _ = PO.difference_map(decoded[(1.0, 4)], decoded[(-1.0, 4)])
    mj, mxj = PO.jpeg_roundtrip(m)
    assert np.array_equal(mj.cpu().numpy()[0], OPO.jpeg_roundtrip(m.cpu().numpy()[0]))
    assert np.array_equal(mxj.cpu().numpy(), mj.cpu().numpy().reshape(F, -1).max(axis=1))


@pytest.mark.parametrize("filt", [False, True])
def test_file_based_get_seg_map_main(tmp_path, filt):
    """The reference's entry point and folder layout (process_output.py:42-167): PNG frames of the +lambda / -lambda decodes in,
    JPEG difference maps and raw PNG / colour JPEG segmentation maps out -- equal to the oracle evaluating the reference's
    expressions on the same files' pixels (uint8 images, JPEG round trip, LANCZOS-resized Step 3 mask PNGs)."""
    import os
    from PIL import Image
    from oracle import process_output as OPO
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd import process_output as PO
    K, F, H, W, lam, base_count = 3, 2, 40, 56, 50.0, 3
    labels = np.array([2, 5, 7])
    names = ["00012", "00013"]
    g = np.random.Generator(np.random.PCG64(31))
    base, exp = str(tmp_path), "clip"
    pos = g.integers(0, 256, (K, F, H, W, 3), dtype=np.uint8)
    neg = pos.copy()
    for k in range(K):
        neg[k, :, 5 + 9 * k:15 + 9 * k, 4 + 12 * k:24 + 12 * k] = g.integers(0, 256, (F, 10, 20, 3), dtype=np.uint8)
    for k, lab in enumerate(labels):
        for sign, arr in ((lam, pos), (-lam, neg)):
            d = os.path.join(base, exp, "modulated_output", f"{base_count:06d}_l_{sign}_mask_{lab}")
            os.makedirs(d)
            for f, n in enumerate(names):
                Image.fromarray(arr[k, f]).save(os.path.join(d, f"{n}.png"))
    lm = labels[g.integers(0, K, (F, 10, 14))].astype(np.int32)
    mask_folder = os.path.join(base, exp, "match_gt_mask", "masks_3")
    FE._write_png_masks(mask_folder, torch.from_numpy(lm), names, 24, labels, True)
    seg = PO.get_seg_map_main(exp, base_count, lam, K, F, filt, filter_s=0.7, unique_labels=labels, base_folder=base,
                              mask_folder=mask_folder, frame_name_list=names, feature_timestep="24")
    weights = PO.mask_weights(lm, labels, (H, W)) if filt else None
    ref, _ = OPO.seg_maps_u8(pos, neg, labels, weights=weights, filter_s=0.7, jpeg=True)
    assert seg.shape == (F, H, W) and np.array_equal(seg, ref)
    suffix = "_f_0.7" if filt else ""
    for f, n in enumerate(names):
        raw = np.array(Image.open(os.path.join(base, exp, f"segmentation_map_raw{suffix}", f"{base_count:06d}_l_{lam}", f"{n}.png")))
        assert np.array_equal(raw, ref[f])
        assert os.path.exists(os.path.join(base, exp, f"segmentation_map{suffix}", f"{base_count:06d}_l_{lam}", f"{n}.jpg"))
        jm = np.array(Image.open(os.path.join(base, exp, "difference_map", "original_map", f"{base_count:06d}_l_{lam}_mask_5", f"{n}.jpg")))
        assert jm.shape == (H, W)
    assert len(np.unique(seg)) > 1
