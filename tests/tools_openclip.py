"""Shapes (open_clip's parameter names) and synthetic weights of CLIP towers, for the OpenCLIP oracle / GPU tests."""
import torch

from vidseg_diffusion_amd import synthetic


def clip_shapes(W, layers, hidden, pre=""):
    s = {}
    for i in range(layers):
        p = f"{pre}transformer.resblocks.{i}."
        s.update({p + "ln_1.weight": (W,), p + "ln_1.bias": (W,), p + "ln_2.weight": (W,), p + "ln_2.bias": (W,),
                  p + "attn.in_proj_weight": (3 * W, W), p + "attn.in_proj_bias": (3 * W,), p + "attn.out_proj.weight": (W, W),
                  p + "attn.out_proj.bias": (W,), p + "mlp.c_fc.weight": (hidden, W), p + "mlp.c_fc.bias": (hidden,),
                  p + "mlp.c_proj.weight": (W, hidden), p + "mlp.c_proj.bias": (W,)})
    return s


def text_shapes(vocab, ctx, W, layers, embed):
    s = clip_shapes(W, layers, 4 * W)
    s.update({"token_embedding.weight": (vocab, W), "positional_embedding": (ctx, W), "ln_final.weight": (W,), "ln_final.bias": (W,),
              "text_projection": (W, embed), "logit_scale": ()})
    return s


def visual_shapes(W, layers, patch, grid, embed):
    s = clip_shapes(W, layers, 4 * W)
    s.update({"conv1.weight": (W, 3, patch, patch), "class_embedding": (W,), "positional_embedding": (grid * grid + 1, W),
              "ln_pre.weight": (W,), "ln_pre.bias": (W,), "ln_post.weight": (W,), "ln_post.bias": (W,), "proj": (W, embed)})
    return s


def fill(shapes, seed):
    return {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=seed).items()}
