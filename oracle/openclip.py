"""CPU oracle (TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it) for the
conditioner's OpenCLIP towers -- SURVEY.md §8(f) rank 4.

What is restated.  The reference reaches the towers through sgm/modules/encoders/modules.py (MOD): FrozenOpenCLIPEmbedder MOD:498-567
(text), FrozenOpenCLIPImageEmbedder MOD:570-728 (image, with `preprocess` MOD:621-633), FrozenOpenCLIPImagePredictionEmbedder
MOD:1028-1046.  The networks are THIRD-PARTY code absent from /root/reference and from this image: open_clip_torch == 2.24.0 and
kornia == 0.7.2 (requirements/pt2.txt:5, 9).  Their published algorithms, restated in plain torch fp32 below:

  open_clip/transformer.py  ResidualAttentionBlock.forward: x = x + ls_1(attention(ln_1(x))); x = x + ls_2(mlp(ln_2(x))), ls = Identity
                            for ViT-H-14; attention = nn.MultiheadAttention(width, heads) (in_proj [3W, W] + bias, softmax(q k^T /
                            sqrt(d) + mask) v, out_proj); mlp = c_fc -> nn.GELU() (erf) -> c_proj; LayerNorm eps 1e-5.
                            VisionTransformer.forward: conv1 (P x P stride P, no bias) -> tokens, class_embedding prepended,
                            + positional_embedding, ln_pre, transformer, ln_post, pooled = token 0, @ proj.
  open_clip/model.py        build_attention_mask: -inf above the diagonal (causal).  ViT-H-14.json: text width 1024 / 16 heads / 24
                            layers / 77 tokens / vocab 49408; vision width 1280 / head width 80 / 32 layers / patch 14 / image 224.
  kornia/geometry/transform/affwarp.py  resize(antialias=True): per axis factor = in / out; if max factor > 1: gaussian_blur2d with
                            sigma = max((factor - 1) / 2, 0.001), window = int(max(4 sigma, 3)) made odd, then F.interpolate.
  kornia/filters/gaussian.py, kernels.py  separable, border_type "reflect", taps exp(-x^2 / (2 sigma^2)) normalised to sum 1.

PARITY PIN.  open_clip and kornia cannot be imported here, so the restatement is pinned against an INDEPENDENT implementation of the same
published architecture that is installed: transformers' CLIPTextModel / CLIPVisionModelWithProjection (the classes the
`laion/CLIP-ViT-H-14-laion2B-s32B-b79K` conversion of this very checkpoint loads into), on the same weights, in
tests/test_oracle_openclip.py; `interpolate(bicubic, align_corners)` IS the torch call kornia makes.  The gaussian antialias pass of
kornia.geometry.resize has no second implementation here: that piece is "parity unpinned" (restated from kornia 0.7.2's source as
published; the window / sigma rule is quoted above so a maintainer with kornia installed can check it in one line).
"""
import math

import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)        # MOD:609-614
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def resblock(x, sd, pre, heads, mask):
    """open_clip ResidualAttentionBlock on x [B, N, W] (batch first; the reference permutes to LND and back, MOD:545-547)."""
    B, N, W = x.shape
    d = W // heads
    h = F.layer_norm(x, (W,), sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"], 1e-5)
    qkv = F.linear(h, sd[pre + "attn.in_proj_weight"], sd[pre + "attn.in_proj_bias"])
    q, k, v = (t.reshape(B, N, heads, d).transpose(1, 2) for t in qkv.chunk(3, -1))
    s = q @ k.transpose(-1, -2) / math.sqrt(d)
    if mask is not None:
        s = s + mask
    a = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, N, W)
    x = x + F.linear(a, sd[pre + "attn.out_proj.weight"], sd[pre + "attn.out_proj.bias"])
    h = F.layer_norm(x, (W,), sd[pre + "ln_2.weight"], sd[pre + "ln_2.bias"], 1e-5)
    h = F.gelu(F.linear(h, sd[pre + "mlp.c_fc.weight"], sd[pre + "mlp.c_fc.bias"]))
    return x + F.linear(h, sd[pre + "mlp.c_proj.weight"], sd[pre + "mlp.c_proj.bias"])


def n_blocks(sd, pre):
    return 1 + max(int(k[len(pre):].split(".")[0]) for k in sd if k.startswith(pre))


def text_encode(sd, tokens, heads, layer="penultimate"):
    """FrozenOpenCLIPEmbedder.encode_with_transformer (MOD:544-565): sd with open_clip's CLIP names, tokens int [B, ctx]."""
    x = sd["token_embedding.weight"][tokens] + sd["positional_embedding"]
    N = x.shape[1]
    mask = torch.full((N, N), float("-inf")).triu_(1)                      # open_clip.model.CLIP.build_attention_mask
    L = n_blocks(sd, "transformer.resblocks.")
    for i in range(L - (1 if layer == "penultimate" else 0)):
        x = resblock(x, sd, f"transformer.resblocks.{i}.", heads, mask)
    return F.layer_norm(x, (x.shape[-1],), sd["ln_final.weight"], sd["ln_final.bias"], 1e-5)


def gaussian_taps(ks, sigma):
    x = torch.arange(ks, dtype=torch.float32) - ks // 2
    g = torch.exp(-x.pow(2.0) / (2.0 * float(sigma) ** 2))
    return g / g.sum()


def kornia_resize(img, size, antialias=True):
    """kornia.geometry.resize(img, (size, size), interpolation="bicubic", align_corners=True, antialias=antialias), kornia 0.7.2."""
    H, W = img.shape[-2:]
    fy, fx = H / size, W / size
    if antialias and max(fy, fx) > 1:
        sy, sx = max((fy - 1.0) / 2.0, 0.001), max((fx - 1.0) / 2.0, 0.001)
        ky, kx = int(max(2.0 * 2 * sy, 3)), int(max(2.0 * 2 * sx, 3))
        ky, kx = ky + (ky % 2 == 0), kx + (kx % 2 == 0)
        C = img.shape[1]
        wx = gaussian_taps(kx, sx).reshape(1, 1, 1, kx).expand(C, 1, 1, kx)
        wy = gaussian_taps(ky, sy).reshape(1, 1, ky, 1).expand(C, 1, ky, 1)
        img = F.conv2d(F.pad(img, (kx // 2, kx // 2, 0, 0), mode="reflect"), wx, groups=C)        # filter2d_separable: x, then y
        img = F.conv2d(F.pad(img, (0, 0, ky // 2, ky // 2), mode="reflect"), wy, groups=C)
    return F.interpolate(img, size=(size, size), mode="bicubic", align_corners=True)


def preprocess(img, size=224, antialias=True):
    """MOD:621-633."""
    x = (kornia_resize(img, size, antialias) + 1.0) / 2.0
    mean, std = torch.tensor(CLIP_MEAN).reshape(1, 3, 1, 1), torch.tensor(CLIP_STD).reshape(1, 3, 1, 1)
    return (x - mean) / std


def visual_forward(sd, x, heads, patch):
    """open_clip VisionTransformer.forward on a preprocessed image [B, 3, S, S]; sd holds the `visual.`-stripped names."""
    x = F.conv2d(x, sd["conv1.weight"], stride=patch)
    B, W = x.shape[:2]
    x = x.reshape(B, W, -1).permute(0, 2, 1)
    x = torch.cat([sd["class_embedding"].reshape(1, 1, W).expand(B, 1, W), x], 1) + sd["positional_embedding"]
    x = F.layer_norm(x, (W,), sd["ln_pre.weight"], sd["ln_pre.bias"], 1e-5)
    for i in range(n_blocks(sd, "transformer.resblocks.")):
        x = resblock(x, sd, f"transformer.resblocks.{i}.", heads, None)
    x = F.layer_norm(x, (W,), sd["ln_post.weight"], sd["ln_post.bias"], 1e-5)
    return x[:, 0] @ sd["proj"]


def image_embed(sd, img, heads, patch, size, antialias=True):
    """FrozenOpenCLIPImageEmbedder.encode_with_vision_transformer (MOD:693-726) for the inference settings (no crops / tokens)."""
    return visual_forward(sd, preprocess(img, size, antialias), heads, patch)
