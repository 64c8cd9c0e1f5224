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


@pytest.mark.parametrize("K,F,H,W", [(3, 2, 16, 24), (5, 3, 33, 47), (2, 1, 64, 64)])
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
