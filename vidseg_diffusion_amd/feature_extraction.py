"""Drop-in mirror of the reference's analysis entry point on MI355X.

`feature_extraction_main` keeps the signature, modes, argument meaning and return triple of
scripts/sampling/feature_extraction.py:670-795, so scripts/sampling/*_pipeline_vspw.py can import
it unchanged.  Differences, all at the hand-off (never in the arithmetic):

* dumps are looked up in the in-HBM `FeatureStore` first (what the UNet taps fill); only if a
  name is missing are the reference's ``{base}/{exp}/feature_maps/{name}.pt`` files read (and
  moved to the GPU) -- feature_extraction.py:646-668;
* masks are kept as int32 label maps in `MaskStore`; the reference's PNG folders
  (feature_extraction.py:618-636, :446-458) are written only when `WRITE_PNG` is true (needed
  by the unmodified Steps 4-5 of the drivers, scripts/sampling/sd_pipeline_vspw.py:64-101);
* `ref_feature_map` is returned as an fp16 device tensor (numpy fp16 accepted on input).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import analysis
from ._lib import VidsegError

WRITE_PNG = bool(int(os.environ.get("VIDSEG_WRITE_PNG", "0")))


class FeatureStore:
    """In-HBM replacement for the reference's ``feature_maps/*.pt`` dump directory
    (scripts/sampling/sd_pipeline_vspw.py:103-139).  Keyed like the files:
    ``{base_folder}/{exp}/feature_maps`` -> ``{block}_{feature_type}_time_{t}``."""
    _stores: dict = {}

    @classmethod
    def folder(cls, base_folder, exp):
        return cls._stores.setdefault(os.path.join(base_folder, exp, "feature_maps"), {})

    @classmethod
    def put(cls, base_folder, exp, name, tensor):
        cls.folder(base_folder, exp)[name] = tensor

    @classmethod
    def clear(cls, base_folder=None, exp=None):
        if base_folder is None:
            cls._stores.clear()
        else:
            cls._stores.pop(os.path.join(base_folder, exp, "feature_maps"), None)

    @classmethod
    def export_pt(cls, base_folder, exp, names=None):
        """Write the store as the reference's dump directory: ``{base}/{exp}/feature_maps/{name}.pt`` holding the same
        tensors ``save_feature_map`` would have saved (sd_pipeline_vspw.py:131-139: fp16 q/k [2F, N, C], xt fp32
        [F, 4, h, w]) so that the unmodified reference scripts (Steps 4-5, feature_extraction_main) can read them.
        Returns the list of files written."""
        folder = os.path.join(base_folder, exp, "feature_maps")
        os.makedirs(folder, exist_ok=True)
        written = []
        for name, t in cls.folder(base_folder, exp).items():
            if names is not None and name not in names:
                continue
            path = os.path.join(folder, name + ".pt")
            torch.save(t.detach().cpu(), path)
            written.append(path)
        return written


class MaskStore:
    """int32 [F, N] label maps keyed by the reference's mask folder path."""
    _masks: dict = {}

    @classmethod
    def put(cls, folder, labels, frame_names, unique_labels):
        cls._masks[folder] = (labels, list(frame_names) if frame_names is not None else None, unique_labels)

    @classmethod
    def get(cls, folder):
        return cls._masks.get(folder)

    @classmethod
    def clear(cls):
        cls._masks.clear()


def _device():
    if not torch.cuda.is_available():
        raise VidsegError("no HIP device: vidseg_diffusion_amd has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def load_experiments_features(feature_maps_paths, blocks, feature_type, t, frame_id=None):
    """feature_extraction.py:646-668 (frame_id=None form): one tensor per experiment path."""
    out = []
    for path in feature_maps_paths:
        name = f"{blocks}_{feature_type}_time_{t}"
        store = FeatureStore._stores.get(path, {})
        if name in store:
            out.append(store[name])
            continue
        file = os.path.join(path, name + ".pt")
        if not os.path.exists(file):
            raise FileNotFoundError(file)
        fm = torch.load(file, map_location="cpu")
        if "attn" not in feature_type and fm.dim() == 4:
            fm = fm.permute(0, 2, 3, 1).reshape(fm.shape[0], -1, fm.shape[1])
        out.append(fm.to(_device()).contiguous())
    return out


def _write_png_masks(folder, labels_hw, frame_names, timestep, unique_labels, rm_existing):
    import shutil
    from PIL import Image
    lab = labels_hw.cpu().numpy()
    for i in range(lab.shape[0]):
        name = frame_names[i] if frame_names is not None else i
        d = os.path.join(folder, f"kmeans_time_{timestep}_frame_{name}")
        if rm_existing and os.path.exists(d):
            shutil.rmtree(d)
        os.makedirs(d, exist_ok=True)
        for l in unique_labels:
            m = np.where(lab[i] == l, 255, 0).astype(np.uint8)
            Image.fromarray(m).convert("L").save(os.path.join(d, f"mask_{int(l)}.png"))


def _as_dev_f16(x, dev):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x.to(device=dev, dtype=torch.float16).contiguous()


def match_gt_mask(feature_blocks, gt_mask_path, feature_height, feature_width, output_folder, num_masks,
                  selected_timestep=24, frame_name_list=None, ref_mask=None, ref_feature_map=None,
                  ref_unique_labels=None, use_gt_mask=False):
    """feature_extraction.py:546-643.  `feature_blocks`: list of fp16 [2F, N, C] dumps to average."""
    dev = feature_blocks[0].device
    F2, N, C = feature_blocks[0].shape
    F = F2 // 2                                                             # FE:550
    h, w = feature_height, feature_width
    if C > 1:
        _, feat = analysis.mean_normalize(feature_blocks, F * N, F * N)     # cond half, FE:551-555
    else:
        raise VidsegError("match_gt_mask: single-channel features are not supported")
    folder = output_folder + f"_masks_{num_masks}"                          # FE:564-565 (last path component renamed)
    if ref_mask is None:
        km = analysis.kmeans_fit(feat, num_masks, n_init=10)                # FE:562-570
        fake = analysis.kmeans_predict(feat[:N], km.centers).cpu().numpy()  # FE:572
        if gt_mask_path is not None:
            from PIL import Image
            mask_np = np.array(Image.open(gt_mask_path).resize((w, h), Image.NEAREST)).flatten()
        else:
            mask_np = fake
        if not use_gt_mask:                                                 # FE:589-595
            ref_mask = np.zeros(h * w).astype(int)
            for fake_label in np.unique(fake):
                values, counts = np.unique(mask_np[fake == fake_label], return_counts=True)
                ref_mask[fake == fake_label] = values[np.argmax(counts)]
        else:
            assert gt_mask_path is not None
            ref_mask = mask_np
        ref_feature_map = feat[:N]
    ref_mask_np = np.asarray(ref_mask)
    if ref_unique_labels is None:
        ref_unique_labels = np.unique(ref_mask_np)
    unique_labels = np.unique(ref_mask_np)
    ref_fm = _as_dev_f16(ref_feature_map, dev)
    ref_lab = torch.from_numpy(ref_mask_np.astype(np.int32)).to(dev)
    labels = analysis.knn_predict(ref_fm, ref_lab, feat)                    # FE:608-613
    labels_fn = labels.view(F, N)
    MaskStore.put(folder, labels_fn, frame_name_list, ref_unique_labels)
    if WRITE_PNG:
        _write_png_masks(folder, labels_fn.view(F, h, w), frame_name_list, selected_timestep, ref_unique_labels, True)
    return unique_labels, labels.cpu().numpy().astype(np.int64), feat       # FE:639-643


def save_inidividual_masks_kmeans(feature_maps, selected_timestep, output_folder, num_frames=14, num_clusters=10,
                                  feature_height=16, feature_width=16, attn_type="spatial", frame_name_list=None, **_unused):
    """feature_extraction.py:30-113, spatial attention type only (the only one the drivers use)."""
    if attn_type != "spatial":
        raise VidsegError(f"kmeans_masks: attn_type {attn_type!r} is not on the supported path")
    F2, N, C = feature_maps.shape
    F = num_frames
    _, feat = analysis.mean_normalize([feature_maps], F * N, (F2 - F) * N)  # FE:39, :45
    km = analysis.kmeans_fit(feat, num_clusters, n_init=10)                 # FE:52-54
    labels = analysis.kmeans_predict(feat, km.centers).view(F2 - F, N)      # FE:55-56
    folder = output_folder + f"_masks_{num_clusters}"
    unique_labels = np.arange(num_clusters)
    MaskStore.put(folder, labels, frame_name_list, unique_labels)
    if WRITE_PNG:
        _write_png_masks(folder, labels.view(-1, feature_height, feature_width), frame_name_list, selected_timestep,
                         unique_labels, False)
    return unique_labels


def correct_low_res_mask(feature_maps, mask_folder, num_clusters=10, feature_height=16, feature_width=16,
                         attn_type="spatial", num_frames=14, timestep=24, frame_name_list=None, ref_unique_labels=None,
                         spatial_filter=True, ref_mask=None):
    """feature_extraction.py:367-461."""
    if attn_type != "spatial":
        raise VidsegError(f"correct_low_res_mask: attn_type {attn_type!r} is not on the supported path")
    dev = feature_maps.device
    F, h, w = num_frames, feature_height, feature_width
    N = h * w
    cond = feature_maps[F:F + F].contiguous()                                # FE:221-227
    all_idx, _ = analysis.dense_tracking(cond, F, h, w)                      # FE:370-373
    entry = MaskStore.get(mask_folder)
    if entry is not None:
        labels = entry[0].view(F, N).to(torch.int32)
    elif os.path.isdir(mask_folder):                                         # FE:380-389, 500-521 (PNG round trip)
        from PIL import Image
        maps = []
        labs = ref_unique_labels if ref_unique_labels is not None else np.arange(num_clusters)
        for i in range(F):
            name = frame_name_list[i] if frame_name_list is not None else i
            masks = [np.array(Image.open(os.path.join(mask_folder, f"kmeans_time_24_frame_{name}", f"mask_{int(l)}.png"))
                              .resize((w, h))) for l in labs]
            maps.append(np.asarray(labs)[np.argmax(masks, axis=0)])
        labels = torch.from_numpy(np.stack(maps).reshape(F, N).astype(np.int32)).to(dev)
    elif ref_mask is not None:
        labels = torch.from_numpy(np.asarray(ref_mask).reshape(F, N).astype(np.int32)).to(dev)
    else:
        raise FileNotFoundError(mask_folder)
    new = analysis.trajectory_vote(all_idx, labels.contiguous(), w, spatial_filter)  # FE:392-421
    out_folder = mask_folder + "_corrected"                                  # FE:446-447
    MaskStore.put(out_folder, new, frame_name_list, ref_unique_labels)
    if WRITE_PNG:
        _write_png_masks(out_folder, new.view(F, h, w), frame_name_list, timestep, ref_unique_labels, False)
    return ref_unique_labels, new.reshape(-1).cpu().numpy().astype(np.int64), None   # FE:460-461


def feature_extraction_main(mode, num_clusters, t_start, block_name, experiment_name, fit_experiments, feature_types,
                            feature_height, feature_width, selected_timestep, frame_name_list=None, base_folder=None,
                            ref_mask=None, ref_feature_map=None, ref_unique_labels=None, gt_mask_path=None,
                            num_frames=None, mask_folder=None, use_gt_mask=False):
    """scripts/sampling/feature_extraction.py:670-795."""
    exp_path_root = "features_outputs" if base_folder is None else base_folder
    selected_timestep = [int(t) for t in selected_timestep.split(",") if t]
    fit_experiments = [item for item in fit_experiments.split(",") if item]
    if num_frames is None:
        num_frames = 14
    block_name = block_name.split(",")
    if len(block_name) == 1:
        block_name = block_name[0]
    paths = [os.path.join(exp_path_root, e, "feature_maps") for e in fit_experiments]
    feature_types = [item for item in feature_types.split(",") if item]
    if mode not in ("kmeans_masks", "correct_low_res_mask", "match_gt_mask"):
        raise ValueError(f"mode {mode} not supported")
    out_root = os.path.join(exp_path_root, experiment_name, mode)
    unique_labels = None
    for t in selected_timestep:
        for feature_type in feature_types:
            if "temporal" in feature_type:
                attn_type = "temporal"
            elif "features" in feature_type:
                attn_type = "features"
            else:
                attn_type = "spatial"
            if isinstance(block_name, list):
                blocks = [torch.cat(load_experiments_features(paths, b, feature_type, t), dim=0) if len(paths) > 1
                          else load_experiments_features(paths, b, feature_type, t)[0] for b in block_name]
                out_path = os.path.join(out_root, f"{'_'.join(block_name)}_{feature_type}")
            else:
                fl = load_experiments_features(paths, block_name, feature_type, t)
                blocks = [torch.cat(fl, dim=0) if len(fl) > 1 else fl[0]]
                out_path = os.path.join(out_root, f"{block_name}_{feature_type}")
            if mode == "kmeans_masks":
                if len(blocks) > 1:
                    blocks = [analysis.mean_normalize(blocks, 0, blocks[0].shape[0] * blocks[0].shape[1], want_mean=True)[0]
                              .view_as(blocks[0])]
                unique_labels = save_inidividual_masks_kmeans(
                    blocks[0], t, out_path, num_frames=num_frames, num_clusters=num_clusters, feature_height=feature_height,
                    feature_width=feature_width, attn_type=attn_type, frame_name_list=frame_name_list)
            elif mode == "match_gt_mask":
                unique_labels, ref_mask, ref_feature_map = match_gt_mask(
                    blocks, gt_mask_path=gt_mask_path, feature_height=feature_height, feature_width=feature_width,
                    output_folder=out_path, num_masks=num_clusters, selected_timestep=t, frame_name_list=frame_name_list,
                    ref_mask=ref_mask, ref_feature_map=ref_feature_map, ref_unique_labels=ref_unique_labels,
                    use_gt_mask=use_gt_mask)
            else:
                if len(blocks) > 1:
                    blocks = [analysis.mean_normalize(blocks, 0, blocks[0].shape[0] * blocks[0].shape[1], want_mean=True)[0]
                              .view_as(blocks[0])]
                unique_labels, ref_mask, ref_feature_map = correct_low_res_mask(
                    blocks[0], mask_folder=mask_folder, num_clusters=num_clusters, feature_height=feature_height,
                    feature_width=feature_width, attn_type=attn_type, timestep=t, num_frames=num_frames,
                    frame_name_list=frame_name_list, ref_unique_labels=ref_unique_labels, ref_mask=ref_mask)
    return unique_labels, ref_mask, ref_feature_map
