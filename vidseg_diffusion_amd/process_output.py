"""Step 5 of the reference driver on the device: `scripts/sampling/process_output.py` (compute_difference PO:8-29,
filter_difference_map PO:31-40, get_seg_map_main PO:75-167) over decoded frames that never leave HBM.

The reference writes every modulated decode as PNG frames, re-reads them, writes each difference map as a JPEG and re-reads
that.  Here the decoded +lambda / -lambda frames of one mask go straight into `vidseg_seg_difference`, the K x F "L" maps stay
in one uint8 tensor and `vidseg_seg_argmax` produces the label map.  What is NOT reproduced is the JPEG round trip of the
difference maps (a lossy codec between two steps of arithmetic); everything else follows the reference line by line,
including numpy's wrapping uint8 subtraction and square (PO:13).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import VidsegError, call, ptr, stream

_lib.register({
    "vidseg_seg_difference": [_lib._P, _lib._P, _lib._I, _lib._I, _lib._I, _lib._P, _lib._P, _lib._P],
    "vidseg_seg_argmax": [_lib._P, _lib._P, _lib._P, _lib._D, _lib._P, _lib._I, _lib._I, _lib._I, _lib._I, _lib._P, _lib._P],
})


def difference_map(pos, neg):
    """compute_difference for the F frames of one mask: pos/neg fp32 NCHW [F, 3, H, W] (decode_first_stage outputs) ->
    (uint8 [F, H, W] "L" images, uint32 [F] per-frame maxima)."""
    if not (pos.is_cuda and neg.is_cuda):
        raise VidsegError("process_output runs on a HIP device only (no CPU fallback)")
    F, C, H, W = pos.shape
    assert C == 3 and neg.shape == pos.shape
    pos, neg = pos.float().contiguous(), neg.float().contiguous()
    out = torch.empty((F, H, W), dtype=torch.uint8, device=pos.device)
    fmax = torch.empty((F,), dtype=torch.int32, device=pos.device)
    call("vidseg_seg_difference", ptr(pos), ptr(neg), F, H, W, ptr(out), ptr(fmax), stream())
    return out, fmax


def mask_weights(label_maps, labels, size):
    """The filter's mask images (PO:36): for each label the binary 0/255 mask of every frame at feature resolution, resized
    to the frame with PIL's LANCZOS exactly like the reference does with the Step 3 PNGs.  label_maps: int [F, h, w]."""
    from PIL import Image
    lm = np.asarray(label_maps.cpu() if torch.is_tensor(label_maps) else label_maps)
    H, W = size
    out = np.empty((len(labels), lm.shape[0], H, W), dtype=np.uint8)
    for k, lab in enumerate(labels):
        for f in range(lm.shape[0]):
            img = Image.fromarray(((lm[f] == int(lab)) * 255).astype(np.uint8))
            out[k, f] = np.array(img.resize((W, H), Image.LANCZOS))
    return out


def seg_map(maps, maxima, labels, weights=None, filter_s=0.7):
    """get_seg_map_main's arg-max: maps uint8 [K, F, H, W], maxima int32 [K, F], labels int [K] (the `mask_iterator`),
    weights uint8 [K, F, H, W] or None (filter_difference=False) -> uint8 [F, H, W] raw segmentation map."""
    K, F, H, W = maps.shape
    dev = maps.device
    lab = torch.as_tensor(np.asarray(labels, dtype=np.int32), device=dev)
    w = None if weights is None else torch.as_tensor(weights, device=dev).contiguous()
    seg = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
    call("vidseg_seg_argmax", ptr(maps.contiguous()), ptr(maxima.contiguous()), ptr(w), float(filter_s), ptr(lab), K, F, H, W, ptr(seg), stream())
    return seg


def get_seg_map(decoded, labels, *, label_maps=None, filter_difference=False, filter_s=0.7):
    """Step 5 for one window.  decoded: {(sign, label): fp32 NCHW [F, 3, H, W]} -- the decoded outputs of
    pipeline.modulation_sweep (sign +1.0 / -1.0); labels: the `unique_labels` of Step 3 in iteration order;
    label_maps (int [F, h, w], Step 3's masks) is needed when filter_difference is set.  Returns uint8 [F, H, W]."""
    labels = [int(v) for v in np.asarray(labels).reshape(-1)]
    maps, maxima = zip(*[difference_map(decoded[(1.0, lab)], decoded[(-1.0, lab)]) for lab in labels])
    maps, maxima = torch.stack(maps), torch.stack(maxima)
    weights = None
    if filter_difference:
        if label_maps is None:
            raise ValueError("filter_difference needs the Step 3 label maps")
        weights = mask_weights(label_maps, labels, maps.shape[-2:])
    return seg_map(maps, maxima, labels, weights, filter_s)


def get_seg_map_main(exp_name, basecount, modulate_lambda, num_masks, num_frames, filter_difference, filter_s=0.7, resize_height=28,
                     resize_width=52, unique_labels=None, base_folder=None, mask_folder=None, frame_name_list=None, feature_timestep="24",
                     is_smooth=False, batch_id=None, color_map_path=None, color_map_mapping="order"):
    """The reference's entry point (PO:75-167) works on the PNG / JPEG files its Step 4 wrote.  This package keeps the modulated
    decodes in HBM, so the file-based form is not mirrored: call `pipeline.segmentation_map_window` (Steps 4-5 in one go) or
    `get_seg_map(decoded, labels, ...)` on the decoded frames instead."""
    raise VidsegError("get_seg_map_main(files on disk) is not mirrored: use pipeline.segmentation_map_window or "
                      "process_output.get_seg_map on HBM-resident decoded frames")
