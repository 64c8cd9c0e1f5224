"""The conditioner's OpenCLIP towers at full size (ViT-H-14, synthetic weights) on one MI355X: milliseconds per call and the distance
from the CPU oracle -- what the towers cost next to a window (SD: once per clip, the empty prompt is a constant; SVD: one frame per window).

    python tools/conditioner_bench.py            # prints one JSON line
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    from oracle import openclip as O
    from tools_openclip import fill, text_shapes, visual_shapes
    from vidseg_diffusion_amd import openclip as C
    a = C.ARCHS["ViT-H-14"]
    t, v = a["text"], a["vision"]
    out = {"arch": "ViT-H-14", "weights": "synthetic (fill_state_dict)", "precision": "exact mode (fp32 activations, split-operand GEMMs)"}
    sd = fill(text_shapes(t["vocab_size"], t["context_length"], t["width"], t["layers"], a["embed_dim"]), 51)
    emb = C.FrozenOpenCLIPEmbedder(layer="penultimate", state_dict=sd)
    t0 = time.perf_counter()
    z = emb([""] * 14)
    torch.cuda.synchronize()
    out["text_first_call_ms_incl_weight_packing"] = round((time.perf_counter() - t0) * 1e3, 1)
    out["text_ms_per_call_14_empty_prompts"] = round(timed(lambda: emb([""] * 14)), 2)
    ref = O.text_encode(sd, C.tokenize([""]), t["heads"], "penultimate")
    out["text_max_err_over_max_ref"] = float((z[:1].cpu().double() - ref.double()).abs().max() / ref.double().abs().max())
    emb.model.release()
    del emb, sd
    sd = fill(visual_shapes(v["width"], v["layers"], v["patch_size"], v["image_size"] // v["patch_size"], a["embed_dim"]), 52)
    im = C.FrozenOpenCLIPImageEmbedder(state_dict={"visual." + k: x for k, x in sd.items()})
    img = torch.tanh(torch.randn(1, 3, 576, 1024, generator=torch.Generator().manual_seed(3)))
    dimg = img.cuda()
    t0 = time.perf_counter()
    z = im(dimg)
    torch.cuda.synchronize()
    out["image_first_call_ms_incl_weight_packing"] = round((time.perf_counter() - t0) * 1e3, 1)
    out["image_ms_per_call_one_576x1024_frame"] = round(timed(lambda: im(dimg)), 2)
    ref = O.image_embed(sd, img, v["width"] // v["head_width"], v["patch_size"], v["image_size"])
    out["image_max_err_over_max_ref"] = float((z.cpu().double() - ref.double()).abs().max() / ref.double().abs().max())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
