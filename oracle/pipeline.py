"""Oracle of one window of Steps 1-3b (scripts/sampling/sd_pipeline_vspw.py:336-409) on the CPU:
euler_sample (oracle/unet.py) with the step-24 Q taps of decoder blocks 6/7/8, then the analysis oracle.
Test infrastructure only; also the `cpu_baseline` ("port") leg of bench.py."""
import numpy as np
import torch

from . import analysis as A
from .unet import UNetOracle, euler_sample


def segment_window(unet: UNetOracle, latent, c_cross, uc_cross, noise, *, num_masks=20, num_steps=25, t_start=22, scale=5.0,
                   is_aggre_attn=True, is_refine_mask=False, seed=17, ref_mask=None, ref_feature_map=None):
    """Returns dict(labels [F,N] int64, q_taps {block: fp16 [2F,N,C]}, x_final)."""
    taps = {}

    def cb(x, i, t):
        if i == 24:
            for b in (6, 7, 8):
                taps[b] = t[f"output_block_{b}_spatial_self_attn_q"].numpy()

    xf = euler_sample(unet, latent, c_cross, uc_cross, num_steps=num_steps, t_start=t_start, scale=scale, noise=noise, callback=cb)
    Fn = latent.shape[0]
    fh, fw = latent.shape[2] // 2, latent.shape[3] // 2
    blocks = [taps[8], taps[7], taps[6]] if is_aggre_attn else [taps[7]]
    agg = A.aggregate_blocks(blocks) if len(blocks) > 1 else blocks[0]
    np.random.seed(seed)
    ul, labels, fm = A.match_gt_mask(agg, num_masks, np.random.mtrand._rand, ref_mask=ref_mask, ref_feature_map=ref_feature_map)
    out = dict(match_labels=labels.reshape(Fn, -1), q_taps=taps, x_final=xf, ref_feature_map=fm)
    if is_refine_mask:
        th, tw = A.dense_tracking(taps[7], Fn, fh, fw)
        corr, _ = A.correct_low_res_mask(labels.reshape(Fn, fh, fw), th, tw)
        labels = corr
    out["labels"] = labels.reshape(Fn, -1)
    return out
