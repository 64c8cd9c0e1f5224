"""Generate tests/golden/analysis_*.npz by running the REFERENCE's own
scripts/sampling/feature_extraction.py (imported read-only from /root/reference) on the
deterministic synthetic dumps of vidseg_diffusion_amd/synthetic.py.

Build-container only.  Each fixture stores the generator arguments, a sha256 of the inputs
(regenerated from the seed by the tests, never stored) and the reference's outputs.

    python tools/gen_golden_analysis.py
"""
import functools
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from ref_import import REF, import_reference  # noqa: E402
from vidseg_diffusion_amd import synthetic  # noqa: E402

CASES = [
    # name, F, h, w, C, K, seed, windows, gt
    dict(name="a_4x8x8_c32_k5", F=4, h=8, w=8, C=32, K=5, seed=11, windows=2, gt=False),
    dict(name="b_6x12x10_c64_k8", F=6, h=12, w=10, C=64, K=8, seed=12, windows=2, gt=True),
    dict(name="c_5x24x24_c48_k6", F=5, h=24, w=24, C=48, K=6, seed=13, windows=1, gt=False),
    dict(name="d_3x20x25_c32_k4", F=3, h=20, w=25, C=32, K=4, seed=14, windows=1, gt=False),
    dict(name="e_14x16x16_c64_k10", F=14, h=16, w=16, C=64, K=10, seed=15, windows=1, gt=False),
    # the benchmarked sizes (BASELINE configs[1] and configs[2]): 14 frames, 640 channels, K = 20; two chained windows, so the
    # 14336 x 14336 (32256 x 32256) 4-NN label propagation of windows > 0 (feature_extraction.py:603-613) is in the fixture
    dict(name="f_14x32x32_c640_k20", F=14, h=32, w=32, C=640, K=20, seed=1, windows=2, gt=False),
    dict(name="g_14x36x64_c640_k20", F=14, h=36, w=64, C=640, K=20, seed=2, windows=2, gt=False, kmeans_masks=False),
]
BLOCKS = ["output_block_8", "output_block_7", "output_block_6"]


def run_case(fe, case, out_dir):
    F, h, w, C, K = case["F"], case["h"], case["w"], case["C"], case["K"]
    base = tempfile.mkdtemp(prefix="vidseg_golden_")
    exp = "exp"
    fm = os.path.join(base, exp, "feature_maps")
    os.makedirs(fm)
    rec = dict(F=F, h=h, w=w, C=C, K=K, seed=case["seed"], windows=case["windows"], gt=case["gt"])
    ref_mask = ref_feature_map = ref_unique_labels = None
    captured = {}
    orig_dt = fe.dense_tracking

    def capture(*a, **k):
        r = orig_dt(*a, **k)
        captured["h"], captured["w"] = np.array(r[0]), np.array(r[1])
        return r

    fe.dense_tracking = capture
    cwd = os.getcwd()
    os.chdir(REF)                                                   # FE:539 relative colour map
    try:
        for win in range(case["windows"]):
            blocks, sha = synthetic.attention_q_dumps(F, h, w, C, num_blocks=3, seed=case["seed"] + 100 * win)
            rec[f"w{win}_input_sha256"] = sha
            for name, t in zip(BLOCKS, blocks):
                torch.save(torch.from_numpy(t), os.path.join(fm, f"{name}_spatial_self_attn_q_time_24.pt"))
            names = [f"{win:02d}{i:03d}" for i in range(F)]
            gt_path = None
            if case["gt"] and win == 0:
                g = np.random.Generator(np.random.PCG64(case["seed"] + 7))
                gt = (g.integers(0, 3, size=(h // 2 + 1, w // 2 + 1)) * 40 + 7).astype(np.uint8)
                gt = np.kron(gt, np.ones((4, 4), dtype=np.uint8))[: 2 * h, : 2 * w]   # 2x feature res
                gt_path = os.path.join(base, "gt.png")
                Image.fromarray(gt).save(gt_path)
                rec["gt_mask_resized"] = np.array(Image.open(gt_path).resize((w, h), Image.NEAREST)).reshape(-1)
            np.random.seed(case["seed"])                                # seed_everything, SDP:619-623
            ul, ref_mask, ref_feature_map = fe.feature_extraction_main(
                "match_gt_mask", K, 22, ",".join(BLOCKS), exp, exp, "spatial_self_attn_q", h, w, "24",
                frame_name_list=names, base_folder=base, num_frames=F, ref_mask=ref_mask,
                ref_feature_map=ref_feature_map, ref_unique_labels=ref_unique_labels, gt_mask_path=gt_path)
            if win == 0:
                ref_unique_labels = ul
            rec[f"w{win}_unique_labels"] = np.asarray(ul)
            rec[f"w{win}_match_labels"] = np.asarray(ref_mask).astype(np.int64)
            rec[f"w{win}_ref_feature_sha256"] = synthetic.sha256_of(np.asarray(ref_feature_map))
            mask_folder = os.path.join(base, exp, "match_gt_mask", "_".join(BLOCKS) + f"_spatial_self_attn_q_masks_{K}")
            _, ref_mask2, _ = fe.feature_extraction_main(
                "correct_low_res_mask", K, 22, "output_block_7", exp, exp, "spatial_self_attn_q", h, w, "24",
                frame_name_list=names, base_folder=base, num_frames=F, ref_mask=ref_mask,
                ref_feature_map=ref_feature_map, ref_unique_labels=ref_unique_labels, gt_mask_path=gt_path,
                mask_folder=mask_folder)
            rec[f"w{win}_track_h"] = captured["h"].astype(np.int16)
            rec[f"w{win}_track_w"] = captured["w"].astype(np.int16)
            rec[f"w{win}_corrected_labels"] = np.asarray(ref_mask2).astype(np.int64)
            ref_mask = ref_mask2                                       # SDP:401 overwrites before next window
            if win == 0 and case.get("kmeans_masks", True):
                np.random.seed(case["seed"])
                fe.feature_extraction_main(
                    "kmeans_masks", K, 22, "output_block_8", exp, exp, "spatial_self_attn_q", h, w, "24",
                    frame_name_list=names, base_folder=base, num_frames=F)
                folder = os.path.join(base, exp, "kmeans_masks", f"output_block_8_spatial_self_attn_q_masks_{K}")
                lab = np.stack([fe.generate_aggregate_mask(folder, 24, K, n, h, w) for n in names])
                rec["w0_kmeans_masks_labels"] = lab.reshape(F, -1).astype(np.int64)
    finally:
        os.chdir(cwd)
        fe.dense_tracking = orig_dt
        shutil.rmtree(base, ignore_errors=True)
    import sklearn
    rec["versions"] = np.array([f"sklearn {sklearn.__version__}", f"numpy {np.__version__}", f"torch {torch.__version__}"])
    np.savez_compressed(os.path.join(out_dir, f"analysis_{case['name']}.npz"), **rec)
    print(case["name"], "ok", {k: (v.shape if hasattr(v, "shape") else v) for k, v in rec.items() if "labels" in k})


def main():
    fe = import_reference()
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    only = set(sys.argv[1:])
    for case in CASES:
        if only and case["name"] not in only and case["name"].split("_")[0] not in only:
            continue
        t0 = time.time()
        run_case(fe, case, out_dir)
        print(f"  {case['name']}: {time.time() - t0:.1f} s")


if __name__ == "__main__":
    main()
