"""Oracle (CPU restatement) of Step 5, scripts/sampling/process_output.py -- test infrastructure only.

    compute_difference      PO:8-29   uint8 frames, `np.sqrt(np.sum((a - b) ** 2, axis=2))` with numpy's wrapping uint8
                                      arithmetic, cv2.GaussianBlur(., (5, 5), 3), Image.fromarray(.).convert("L")
    filter_difference_map   PO:31-40
    get_seg_map_main        PO:75-167 map / (max + 1e-5), arg-max over masks, label lookup
    frame -> uint8          SDP:152-168 clamp((x + 1) / 2, 0, 1) * 255, .astype(np.uint8)

PARITY UNPINNED for the blur: OpenCV is not installed in the build container, so cv2.GaussianBlur is restated
from its documentation (getGaussianKernel(5, 3) = normalised exp(-(i-2)^2 / 18), separable, BORDER_REFLECT_101, float64) and
checked against a direct 2-D numpy evaluation and against scipy.ndimage's independent separable correlation with mirror
borders (tests/test_oracle_process_output.py); the result is truncated to uint8, so only a float64 value within ~1e-13 of an
integer could depend on the summation order.  The wrapping uint8 arithmetic IS pinned (it is numpy's own, evaluated literally
below).  The reference's JPEG save / re-load of every difference map (PO:19, 119) is reproduced with `jpeg=True` through PIL,
the codec the reference itself calls (the HBM-resident default skips it).
"""
from __future__ import annotations

import numpy as np


def frames_to_uint8(x):
    """x: float32 [F, 3, H, W] decoded frames -> uint8 [F, H, W, 3] (SDP:152, 162-168)."""
    x = np.asarray(x, dtype=np.float32)
    t = np.clip((x + np.float32(1.0)) / np.float32(2.0), np.float32(0.0), np.float32(1.0))
    return (np.transpose(t, (0, 2, 3, 1)) * np.float32(255.0)).astype(np.uint8)


def gaussian_kernel_5_3():
    # exp(-(i-2)^2 / 18) / sum, written out (the device uses the same literals; tests check them against the formula)
    return np.array([0.1782032576265784, 0.2105222740037377, 0.22254893673936782, 0.2105222740037377, 0.1782032576265784])


def gaussian_blur_5_3(d):
    """Separable 5x5, sigma 3, reflect-101 borders, float64; symmetric taps summed in pairs (centre, +-1, +-2)."""
    k = gaussian_kernel_5_3()
    p = np.pad(d, 2, mode="reflect")
    H, W = d.shape
    rows = k[2] * p[:, 2:2 + W] + k[1] * (p[:, 1:1 + W] + p[:, 3:3 + W]) + k[0] * (p[:, 0:W] + p[:, 4:4 + W])
    return k[2] * rows[2:2 + H] + k[1] * (rows[1:1 + H] + rows[3:3 + H]) + k[0] * (rows[0:H] + rows[4:4 + H])


def compute_difference(img1_u8, img2_u8):
    """PO:8-29 for one frame: two uint8 [H, W, 3] images -> the "L" difference image uint8 [H, W]."""
    difference = np.sqrt(np.sum((img1_u8 - img2_u8) ** 2, axis=2))          # uint8 - uint8 and ** 2 wrap, like the reference's line
    difference = gaussian_blur_5_3(difference)
    return np.clip(difference, 0.0, 255.0).astype(np.uint8)                  # PIL "F" -> "L": clip, truncate


def jpeg_roundtrip(img_u8):
    """PO:18-19, 119: Image.fromarray(.).convert("L").save(x.jpg) with PIL's defaults, then np.array(Image.open(x.jpg))."""
    import io
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(img_u8).convert("L").save(buf, format="JPEG")
    buf.seek(0)
    return np.array(Image.open(buf))


def seg_maps(pos, neg, labels, weights=None, filter_s=0.7, jpeg=False):
    """pos/neg: float32 [K, F, 3, H, W] decoded +lambda / -lambda frames per mask (in `labels` order); weights: uint8
    [K, F, H, W] resized mask images or None.  Returns uint8 [F, H, W] (PO:119-161)."""
    K = pos.shape[0]
    return seg_maps_u8(np.stack([frames_to_uint8(pos[k]) for k in range(K)]), np.stack([frames_to_uint8(neg[k]) for k in range(K)]),
                       labels, weights, filter_s, jpeg)


def seg_maps_u8(pos_u8, neg_u8, labels, weights=None, filter_s=0.7, jpeg=False):
    """The same from the uint8 HWC images [K, F, H, W, 3] the reference reads back from its PNGs; jpeg: the difference maps go
    through the JPEG save / re-load before they are normalised (the reference's file-based behaviour)."""
    K, F = pos_u8.shape[:2]
    maps = np.stack([np.stack([compute_difference(a, b) for a, b in zip(pos_u8[k], neg_u8[k])]) for k in range(K)])
    if jpeg:
        maps = np.stack([np.stack([jpeg_roundtrip(maps[k, f]) for f in range(F)]) for k in range(K)])
    all_maps = []
    for k in range(K):
        per_frame = []
        for f in range(F):
            dm = maps[k, f] / (np.max(maps[k, f]) + 1e-5)
            if weights is not None:
                m = weights[k, f] / 255.0
                dm = dm * m + filter_s * dm * (1 - m)
            per_frame.append(dm)
        all_maps.append(per_frame)
    out = np.zeros(maps.shape[1:], dtype=np.uint8)
    lab = np.asarray(labels)
    for f in range(F):
        out[f] = lab[np.argmax(np.array([all_maps[k][f] for k in range(K)]), axis=0)].astype(np.uint8)
    return out, maps
