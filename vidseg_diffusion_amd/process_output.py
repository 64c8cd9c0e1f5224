"""Step 5 of the reference driver on the device: `scripts/sampling/process_output.py` (compute_difference PO:8-29,
filter_difference_map PO:31-40, get_seg_map_main PO:75-167) over decoded frames that never leave HBM.

The reference writes every modulated decode as PNG frames, re-reads them, writes each difference map as a JPEG and re-reads
that.  Two forms here:
  * HBM-resident (`get_seg_map`, `pipeline.segmentation_map_window`): the decoded +lambda / -lambda frames of one mask go straight
    into `vidseg_seg_difference`, the K x F "L" maps stay in one uint8 tensor and `vidseg_seg_argmax` produces the label map.  The
    JPEG save / re-load of the difference maps (PO:19, 119 -- a lossy codec between two steps of arithmetic) is skipped by
    default and reproduced with `jpeg_compat=True` (PIL's codec, the one the reference calls, on the host);
  * file-based (`get_seg_map_main`, the reference's entry point and folder layout): PNG frames of Step 4 in, JPEG difference maps,
    raw PNG + colour JPEG segmentation maps out, the difference / arg-max arithmetic on the device.
Everything follows the reference line by line, including numpy's wrapping uint8 subtraction and square (PO:13).
"""
from __future__ import annotations

import io
import os

import numpy as np
import torch

from . import _lib
from ._lib import VidsegError, call, ptr, stream

_lib.register({
    "vidseg_seg_difference": [_lib._P, _lib._P, _lib._I, _lib._I, _lib._I, _lib._P, _lib._P, _lib._P],
    "vidseg_seg_difference_u8": [_lib._P, _lib._P, _lib._I, _lib._I, _lib._I, _lib._P, _lib._P, _lib._P],
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


def difference_map_u8(pos_u8, neg_u8):
    """compute_difference on the images as the reference reads them from the Step 4 PNGs: uint8 HWC [F, H, W, 3] device
    tensors -> (uint8 [F, H, W] "L" images, int32 [F] maxima)."""
    if not (pos_u8.is_cuda and neg_u8.is_cuda):
        raise VidsegError("process_output runs on a HIP device only (no CPU fallback)")
    F, H, W, C = pos_u8.shape
    assert C == 3 and neg_u8.shape == pos_u8.shape and pos_u8.dtype == torch.uint8 and neg_u8.dtype == torch.uint8
    pos_u8, neg_u8 = pos_u8.contiguous(), neg_u8.contiguous()
    out = torch.empty((F, H, W), dtype=torch.uint8, device=pos_u8.device)
    fmax = torch.empty((F,), dtype=torch.int32, device=pos_u8.device)
    call("vidseg_seg_difference_u8", ptr(pos_u8), ptr(neg_u8), F, H, W, ptr(out), ptr(fmax), stream())
    return out, fmax


def jpeg_roundtrip(maps_u8):
    """PO:18-19 + PO:119: every "L" difference image saved as a JPEG (PIL defaults: quality 75) and loaded again; the maxima the
    arg-max normalises by are those of the RE-LOADED images (PO:121).  maps_u8: uint8 [..., H, W] device tensor ->
    (same shape uint8 device tensor, int32 [...] maxima).  Host codec (PIL, the library the reference calls): compat mode only."""
    from PIL import Image
    host = maps_u8.cpu().numpy()
    flat = host.reshape(-1, host.shape[-2], host.shape[-1])
    out = np.empty_like(flat)
    for i in range(flat.shape[0]):
        buf = io.BytesIO()
        Image.fromarray(flat[i]).convert("L").save(buf, format="JPEG")
        buf.seek(0)
        out[i] = np.array(Image.open(buf))
    mx = out.reshape(flat.shape[0], -1).max(axis=1).astype(np.int32).reshape(host.shape[:-2])
    return torch.from_numpy(out.reshape(host.shape)).to(maps_u8.device), torch.from_numpy(mx).to(maps_u8.device)


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


def get_seg_map(decoded, labels, *, label_maps=None, filter_difference=False, filter_s=0.7, jpeg_compat=False):
    """Step 5 for one window.  decoded: {(sign, label): fp32 NCHW [F, 3, H, W]} -- the decoded outputs of
    pipeline.modulation_sweep (sign +1.0 / -1.0); labels: the `unique_labels` of Step 3 in iteration order;
    label_maps (int [F, h, w], Step 3's masks) is needed when filter_difference is set; jpeg_compat reproduces the
    reference's JPEG save / re-load of every difference map (`jpeg_roundtrip`).  Returns uint8 [F, H, W]."""
    labels = [int(v) for v in np.asarray(labels).reshape(-1)]
    maps, maxima = zip(*[difference_map(decoded[(1.0, lab)], decoded[(-1.0, lab)]) for lab in labels])
    maps, maxima = torch.stack(maps), torch.stack(maxima)
    if jpeg_compat:
        maps, maxima = jpeg_roundtrip(maps)
    weights = None
    if filter_difference:
        if label_maps is None:
            raise ValueError("filter_difference needs the Step 3 label maps")
        weights = mask_weights(label_maps, labels, maps.shape[-2:])
    return seg_map(maps, maxima, labels, weights, filter_s)


def default_color_map(n=256):
    """Deterministic stand-in palette [n, 3] for the reference's `scripts/util/color_map_soft.txt` (a data file of the reference
    tree, used for the colour visualisation only -- the raw label PNG does not depend on it)."""
    i = np.arange(n, dtype=np.int64)
    return np.stack([(i * 97 + 40) % 256, (i * 57 + 120) % 256, (i * 151 + 200) % 256], axis=1).astype(np.float64)


def get_seg_map_main(exp_name, basecount, modulate_lambda, num_masks, num_frames, filter_difference, filter_s=0.7, resize_height=28,
                     resize_width=52, unique_labels=None, base_folder=None, mask_folder=None, frame_name_list=None, feature_timestep="24",
                     is_smooth=False, batch_id=None, color_map_path=None, color_map_mapping="order"):
    """The reference's file-based entry point (PO:75-167), same arguments and the same folder layout:

        in   {base}/{exp}/modulated_output/{basecount:06d}_l_{+-lambda}_mask_{i}/{frame}.png           (Step 4's decodes, SDP:152-168)
             {mask_folder}/kmeans_time_{t}_frame_{frame}/mask_{i}.png                                  (Step 3, filter_difference only)
        out  {base}/{exp}/difference_map/original_map/{...}_mask_{i}/{frame}.jpg, .../vis_map/...       (PO:19, 27)
             {base}/{exp}/segmentation_map_raw[_f_{s}]/{basecount:06d}_l_{lambda}/{frame}.png           (PO:155-157)
             {base}/{exp}/segmentation_map[_f_{s}]/{basecount:06d}_l_{lambda}/{frame}.jpg               (PO:158-163)

    The PNG / JPEG codecs are PIL's on the host (as in the reference); the wrapped-uint8 distance, the blur, the normalisation,
    the filter and the arg-max run on the device.  Returns the raw segmentation maps uint8 [F, H, W] (numpy), which the
    reference only writes to disk.  The Step 3 masks come from `mask_folder`'s PNGs or, if that folder does not exist, from the
    MaskStore entry of the same name."""
    from PIL import Image
    from .feature_extraction import MaskStore
    if base_folder is None:                                                   # PO:44-46
        base_folder = "outputs"
        modulated_map_folder = f"outputs/modulate_video_sample/svd/{exp_name}"
    else:
        modulated_map_folder = os.path.join(base_folder, f"{exp_name}/modulated_output")
    dev = torch.device("cuda", torch.cuda.current_device())
    mask_iterator = np.asarray(unique_labels) if unique_labels is not None else np.arange(num_masks)
    names = [frame_name_list[f] if frame_name_list is not None else f for f in range(num_frames)]
    out_root = os.path.join(base_folder, f"{exp_name}/difference_map/original_map/")
    vis_root = os.path.join(base_folder, f"{exp_name}/difference_map/vis_map/")
    maps = []
    for i in mask_iterator:                                                   # generate_difference_map, PO:42-72
        tag = f"{basecount:06d}_l_{modulate_lambda}_mask_{i}"
        d1 = os.path.join(modulated_map_folder, tag)
        d2 = os.path.join(modulated_map_folder, f"{basecount:06d}_l_{-modulate_lambda}_mask_{i}")
        pos = np.stack([np.array(Image.open(os.path.join(d1, f"{n}.png"))) for n in names])
        neg = np.stack([np.array(Image.open(os.path.join(d2, f"{n}.png"))) for n in names])
        m, mx = difference_map_u8(torch.from_numpy(pos).to(dev), torch.from_numpy(neg).to(dev))
        mh, mxh = m.cpu().numpy(), mx.cpu().numpy()
        os.makedirs(os.path.join(out_root, tag), exist_ok=True)
        os.makedirs(os.path.join(vis_root, tag), exist_ok=True)
        back = np.empty_like(mh)
        for f, n in enumerate(names):
            path = os.path.join(out_root, tag, f"{n}.jpg")
            Image.fromarray(mh[f]).convert("L").save(path)                    # PO:18-19
            vis = mh[f].astype(np.float64) / max(float(mxh[f]), 1e-12) * 255  # PO:23-27 (visualisation only)
            Image.fromarray(vis).convert("L").save(os.path.join(vis_root, tag, f"{n}.jpg"))
            back[f] = np.array(Image.open(path))                              # PO:119
        maps.append(back)
    maps = torch.from_numpy(np.stack(maps)).to(dev)                           # [K, F, H, W]
    K, F, H, W = maps.shape
    maxima = maps.reshape(K, F, -1).max(dim=2).values.to(torch.int32)         # PO:121: the max of the RE-LOADED image
    weights = None
    if filter_difference:                                                     # PO:122-134 (the "kmeans" branch is the one reachable)
        if mask_folder is None:
            mask_folder = f"features_outputs/kmeans_masks/{exp_name}/output_block_8_spatial_self_attn_q_masks_{num_masks}"
        weights = np.empty((K, F, H, W), dtype=np.uint8)
        entry = None if os.path.isdir(mask_folder) else MaskStore.get(mask_folder)
        for k, i in enumerate(mask_iterator):
            for f, n in enumerate(names):
                if entry is None:
                    img = Image.open(os.path.join(mask_folder, f"kmeans_time_{feature_timestep}_frame_{n}", f"mask_{i}.png"))
                else:
                    lab = entry[0]
                    lab = (lab.cpu().numpy() if torch.is_tensor(lab) else np.asarray(lab)).reshape(len(names), -1)[f]
                    side = int(round(lab.size ** 0.5))
                    img = Image.fromarray(((lab.reshape(-1, side) == int(i)) * 255).astype(np.uint8))
                weights[k, f] = np.array(img.resize((W, H), Image.LANCZOS))   # PO:35-36
    seg = seg_map(maps, maxima, [int(v) for v in mask_iterator], weights, filter_s).cpu().numpy()
    suffix = f"_f_{filter_s}" if filter_difference else ""
    seg_folder = os.path.join(base_folder, f"{exp_name}/segmentation_map{suffix}", f"{basecount:06d}_l_{modulate_lambda}")
    raw_folder = os.path.join(base_folder, f"{exp_name}/segmentation_map_raw{suffix}", f"{basecount:06d}_l_{modulate_lambda}")
    os.makedirs(seg_folder, exist_ok=True)
    os.makedirs(raw_folder, exist_ok=True)
    if color_map_path is None:
        color_map_path = "scripts/util/color_map_soft.txt"
    color_map = np.loadtxt(color_map_path, delimiter=",") if os.path.exists(color_map_path) else default_color_map()
    order = {int(v): k for k, v in enumerate(mask_iterator)}
    for f, n in enumerate(names):
        Image.fromarray(seg[f].astype(np.uint8)).save(os.path.join(raw_folder, f"{n}.png"))
        idx = np.vectorize(order.get)(seg[f]) if color_map_mapping == "order" else seg[f]
        Image.fromarray(color_map[idx].astype(np.uint8)).save(os.path.join(seg_folder, f"{n}.jpg"))
    return seg
