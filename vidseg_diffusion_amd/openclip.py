"""The conditioner's OpenCLIP ViT-H towers on the HIP path.

Reference call sites (sgm/modules/encoders/modules.py):
    FrozenOpenCLIPEmbedder                 :498-567   SD 2.1's `txt` embedder (sd_2_1.yaml:39-43, layer "penultimate"): token + positional
                                                      embedding, the first 23 of 24 causal transformer blocks, ln_final -> [B, 77, 1024]
    FrozenOpenCLIPImageEmbedder            :570-728   kornia resize to 224 x 224 (bicubic, align_corners, antialias), (x + 1) / 2, CLIP mean /
                                                      std, open_clip's VisionTransformer -> [B, 1024]
    FrozenOpenCLIPImagePredictionEmbedder  :1028-1046 SVD's `cond_frames_without_noise` embedder (svd.yaml:43-50): the image embedder,
                                                      "(b t) d -> b t d", repeated n_copies times -> `crossattn` [B, 1, 1024]

The networks themselves are open_clip_torch 2.24.0 (requirements/pt2.txt:9; absent from this image): `open_clip.transformer.
ResidualAttentionBlock` (x + attn(ln_1 x), x + mlp(ln_2 x), nn.MultiheadAttention, erf GELU, LayerNorm eps 1e-5), `TextTransformer`'s
causal mask, `VisionTransformer` (14 x 14 stride-14 patch convolution without bias, class token, ln_pre, blocks, ln_post on the class
token, `proj`).  Module / parameter names follow open_clip's, so the released checkpoints' `conditioner.embedders.N.model.*` /
`...open_clip.model.visual.*` keys load unchanged (engine.DiffusionEngine.load_state_dict).

Arithmetic: the exact mode's operators -- fp32 activations, every projection one split-operand MFMA GEMM (exact.linear_x), LayerNorm /
GELU / attention in fp32 (csrc/exact_ops.hip, csrc/clip_ops.hip) -- because the towers run once per clip (SD: the empty prompt, a
constant; SVD: one frame), so their cost is nothing and their accuracy is the UNet's input.  No CPU path: tensors live on the HIP
device, a missing library raises.

Tokenisation: open_clip's BPE merges file is part of the absent package.  The drivers' prompt is the empty string
(sd_pipeline_vspw.py:35, 280-281), whose tokenisation is [<start_of_text> = 49406, <end_of_text> = 49407, 0 x 75]; that and
pre-tokenised int tensors need no file.  Other prompts go through `SimpleTokenizer` (the byte-level BPE restated) once the merges file's
path is given (`bpe_path=` / VIDSEG_OPENCLIP_BPE); without it they raise with that explanation.
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, List, Optional, Sequence, Union

import torch
import torch.nn as nn

from . import _lib, exact as X
from ._lib import VidsegError, call, ptr, stream
from .conditioner import AbstractEmbModel
from .util import instantiate_from_config

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
_lib.register({
    "vidseg_clip_attention_f32": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "vidseg_clip_layernorm_f32": [_P, _L, _I, _P, _P, _F, _P, _P],
    "vidseg_clip_gelu_split3": [_P, _L, _I, _P, _P],
    "vidseg_clip_blur_axis": [_P, _L, _I, _I, _P, _I, _I, _P, _P],
    "vidseg_clip_resize_patches": [_P, _I, _I, _I, _I, _I, ctypes.POINTER(_F), ctypes.POINTER(_F), _P, _I, _P],
})

F32 = torch.float32
SOT, EOT = 49406, 49407
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)                    # modules.py:609-614
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

# open_clip/model_configs/ViT-H-14.json
ARCHS = {
    "ViT-H-14": {"embed_dim": 1024,
                 "text": {"context_length": 77, "vocab_size": 49408, "width": 1024, "heads": 16, "layers": 24},
                 "vision": {"image_size": 224, "patch_size": 14, "width": 1280, "head_width": 80, "layers": 32, "mlp_ratio": 4.0}},
}


def _arch(arch):
    if isinstance(arch, dict):
        return arch
    if arch not in ARCHS:
        raise VidsegError(f"OpenCLIP arch {arch!r}: only {sorted(ARCHS)} (or an explicit config dict) is defined here")
    return ARCHS[arch]


# ----------------------------------------------------------------------------- operators (csrc/clip_ops.hip)
def attention(qkv, B, N, heads, causal):
    """softmax(q k^T / sqrt(d) [+ causal mask]) v on the rows of a fused projection: qkv fp32 [B * N, 3 W] -> [B * N, W]."""
    W = qkv.shape[-1] // 3
    d = W // heads
    out = torch.empty((B * N, W), dtype=F32, device=qkv.device)
    ld = qkv.shape[-1]
    base = qkv.data_ptr()
    call("vidseg_clip_attention_f32", base, ld, base + 4 * W, ld, base + 8 * W, ld, ptr(out), W, B, heads, N, N, d, 1.0 / math.sqrt(d),
         int(causal), stream())
    return out


def layernorm(x, gamma, beta, eps=1e-5):
    C = x.shape[-1]
    out = torch.empty_like(x)
    call("vidseg_clip_layernorm_f32", ptr(x), x.numel() // C, C, ptr(gamma), ptr(beta), eps, ptr(out), stream())
    return out


def gelu_split3(y):
    C = y.shape[-1]
    out = X._image(y.shape[:-1] + (3 * C,), y.device)
    call("vidseg_clip_gelu_split3", ptr(y), y.numel() // C, C, ptr(out), stream())
    return out


def gaussian_taps(ks, sigma):
    """kornia.filters.get_gaussian_kernel1d (0.7.2) for an odd window, fp32."""
    x = torch.arange(ks, dtype=F32) - ks // 2
    g = torch.exp(-x.pow(2.0) / (2.0 * float(sigma) ** 2))
    return g / g.sum()


def preprocess_patches(img, size, patch, k_pad, antialias=True):
    """modules.py:621-633 (`preprocess`) + the im2col of the patch convolution: img fp32 NCHW [B, 3, H, W] in [-1, 1] ->
    [B * (size / patch)^2, k_pad] fp32 (columns (c, ky, kx), zero beyond 3 * patch^2).
    kornia.geometry.resize(antialias=True) (0.7.2): when an axis shrinks, gaussian_blur2d with sigma = max((factor - 1) / 2, 0.001) and
    an odd window int(max(4 sigma, 3)) (+1 if even) per axis, separable, reflect border; then bicubic interpolation, align_corners."""
    if img.dim() != 4 or img.shape[1] != 3 or img.dtype != F32 or not img.is_cuda:
        raise VidsegError("preprocess_patches: fp32 NCHW [B, 3, H, W] on the HIP device")
    img = img.contiguous()
    B, _, H, W = img.shape
    fy, fx = H / size, W / size
    if antialias and max(fy, fx) > 1:
        sy, sx = max((fy - 1.0) / 2.0, 0.001), max((fx - 1.0) / 2.0, 0.001)
        ky, kx = int(max(2.0 * 2 * sy, 3)), int(max(2.0 * 2 * sx, 3))
        ky, kx = ky + (ky % 2 == 0), kx + (kx % 2 == 0)
        tx, ty = gaussian_taps(kx, sx).to(img.device), gaussian_taps(ky, sy).to(img.device)
        tmp = torch.empty_like(img)
        call("vidseg_clip_blur_axis", ptr(img), B * 3, H, W, ptr(tx), kx, 1, ptr(tmp), stream())          # filter2d_separable: x first, then y
        blurred = torch.empty_like(img)
        call("vidseg_clip_blur_axis", ptr(tmp), B * 3, H, W, ptr(ty), ky, 0, ptr(blurred), stream())
        img = blurred
    G = size // patch
    out = torch.zeros((B * G * G, k_pad), dtype=F32, device=img.device)
    m, s = (_F * 3)(*CLIP_MEAN), (_F * 3)(*CLIP_STD)
    call("vidseg_clip_resize_patches", ptr(img), B, H, W, size, patch, m, s, ptr(out), k_pad, stream())
    return out


# ----------------------------------------------------------------------------- parameter containers (open_clip's names)
def _param(*shape):
    """A named slot for a checkpoint tensor: on the meta device until load_state_dict(assign=True) puts the host master there."""
    return nn.Parameter(torch.empty(tuple(shape), device="meta"), requires_grad=False)


class _Affine(nn.Module):                                          # nn.LayerNorm / nn.Linear as named parameter holders
    def __init__(self, *shape, bias=True):
        super().__init__()
        self.weight = _param(*shape)
        self.bias = _param(shape[0]) if bias else None


class _MHA(nn.Module):                                             # nn.MultiheadAttention's parameters
    def __init__(self, W):
        super().__init__()
        self.in_proj_weight = _param(3 * W, W)
        self.in_proj_bias = _param(3 * W)
        self.out_proj = _Affine(W, W)


class _MLP(nn.Module):
    def __init__(self, W, hidden):
        super().__init__()
        self.c_fc = _Affine(hidden, W)
        self.c_proj = _Affine(W, hidden)


class _ResBlock(nn.Module):                                        # open_clip.transformer.ResidualAttentionBlock
    def __init__(self, W, hidden):
        super().__init__()
        self.ln_1 = _Affine(W)
        self.attn = _MHA(W)
        self.ln_2 = _Affine(W)
        self.mlp = _MLP(W, hidden)


class _Transformer(nn.Module):
    def __init__(self, W, layers, hidden):
        super().__init__()
        self.resblocks = nn.ModuleList([_ResBlock(W, hidden) for _ in range(layers)])


class _Packed:
    """Device copies of one tower's parameters in the exact mode's formats, built on first use and dropped when weights change."""

    def __init__(self):
        self.dev = None
        self.blocks = []
        self.extra = {}


def _pack_block(b: _ResBlock, dev):
    f = lambda t: t.detach().to(device=dev, dtype=F32).contiguous()    # noqa: E731
    return dict(ln1=(f(b.ln_1.weight), f(b.ln_1.bias)), ln2=(f(b.ln_2.weight), f(b.ln_2.bias)),
                w_in=X.pack_linear_x(b.attn.in_proj_weight, dev), b_in=f(b.attn.in_proj_bias),
                w_out=X.pack_linear_x(b.attn.out_proj.weight, dev), b_out=f(b.attn.out_proj.bias),
                w_fc=X.pack_linear_x(b.mlp.c_fc.weight, dev), b_fc=f(b.mlp.c_fc.bias),
                w_pr=X.pack_linear_x(b.mlp.c_proj.weight, dev), b_pr=f(b.mlp.c_proj.bias))


def run_blocks(x, packed_blocks, B, N, heads, causal):
    """x fp32 [B * N, W] through ResidualAttentionBlocks: x + out_proj(attn(in_proj(ln_1 x))); x + c_proj(gelu(c_fc(ln_2 x)))."""
    for p in packed_blocks:
        qkv = X.linear_x(X.layernorm_split3(x, *p["ln1"]), p["w_in"], p["b_in"])
        a = attention(qkv, B, N, heads, causal)
        x = X.linear_x(X.split3(a), p["w_out"], p["b_out"], residual=x)
        y = X.linear_x(X.layernorm_split3(x, *p["ln2"]), p["w_fc"], p["b_fc"])
        x = X.linear_x(gelu_split3(y), p["w_pr"], p["b_pr"], residual=x)
    return x


class _Tower(nn.Module):
    HOST_MASTERS = True          # engine.DiffusionEngine._apply leaves these modules alone (.to / .cuda / .half are no-ops)

    def __init__(self):
        super().__init__()
        self._packed = _Packed()

    def _blocks_on(self, dev, resblocks):
        if dev.type != "cuda":
            raise VidsegError("OpenCLIP towers run on the HIP device only (no CPU path)")
        unset = [n for n, p in self.named_parameters() if p.is_meta and n not in self.OPTIONAL]
        if unset:
            raise VidsegError(f"OpenCLIP tower: {len(unset)} parameters were never loaded (first: {unset[0]}); load the checkpoint's "
                              "`conditioner.embedders.N.*` keys (engine.load_state_dict) or pass state_dict= to the embedder")
        pk = self._packed
        if pk.dev != dev:
            pk.dev, pk.blocks, pk.extra = dev, [_pack_block(b, dev) for b in resblocks], {}
            torch.cuda.synchronize(dev)          # the copies outlive this call and may be read from another stream (a pipeline lane) later
        return pk

    OPTIONAL = ("text_projection", "logit_scale")                   # carried by checkpoints, read by nobody on this path

    def load_state_dict(self, state_dict, strict=False, assign=True):
        """The host masters of the parameters (fp32, CPU); the device copies are rebuilt on the next call."""
        self.release()
        sd = {k: v.detach().to(device="cpu", dtype=F32) for k, v in state_dict.items()}
        return super().load_state_dict(sd, strict=strict, assign=True)

    def _apply(self, fn, recurse=True):                             # .to(device) / .half() of an owning engine: the masters stay where they are
        return self

    def release(self):
        """Drop the device copies (they are rebuilt on the next call); waits for the kernels that may still read them."""
        if self._packed.dev is not None:
            torch.cuda.synchronize(self._packed.dev)
        self._packed = _Packed()


class TextTower(_Tower):
    """open_clip.CLIP's text half (`del model.visual`, modules.py:516): token_embedding, positional_embedding, transformer, ln_final
    (+ text_projection / logit_scale, which FrozenOpenCLIPEmbedder never reads but the checkpoints carry)."""

    def __init__(self, arch="ViT-H-14"):
        super().__init__()
        a = _arch(arch)
        t = a["text"]
        self.cfg = t
        W = t["width"]
        self.token_embedding = _Affine(t["vocab_size"], W, bias=False)
        self.positional_embedding = _param(t["context_length"], W)
        self.transformer = _Transformer(W, t["layers"], int(W * t.get("mlp_ratio", 4.0)))
        self.ln_final = _Affine(W)
        self.text_projection = _param(W, a["embed_dim"])
        self.logit_scale = _param()

    def encode(self, tokens, skip_last):
        """modules.py:544-565: [B, ctx] int tokens -> ln_final of the residual stream after all but the last `skip_last` blocks."""
        dev = tokens.device
        if not tokens.is_cuda:
            raise VidsegError("OpenCLIP text tower: tokens must be on the HIP device (no CPU path)")
        t = self.cfg
        B, N = tokens.shape
        if N != t["context_length"]:
            raise VidsegError(f"OpenCLIP text tower: {N} tokens, context length is {t['context_length']}")
        pk = self._blocks_on(dev, self.transformer.resblocks)
        if not pk.extra:
            f = lambda p: p.detach().to(device=dev, dtype=F32).contiguous()   # noqa: E731
            pk.extra = dict(tok=f(self.token_embedding.weight), pos=f(self.positional_embedding), lnf=(f(self.ln_final.weight), f(self.ln_final.bias)))
            torch.cuda.synchronize(dev)
        x = (pk.extra["tok"][tokens.long()] + pk.extra["pos"]).reshape(B * N, -1).contiguous()     # gather + add of 77 rows: plumbing
        x = run_blocks(x, pk.blocks[:len(pk.blocks) - skip_last], B, N, t["heads"], causal=True)
        return layernorm(x, *pk.extra["lnf"]).reshape(B, N, -1)


class VisualTower(_Tower):
    """open_clip.transformer.VisionTransformer (pool 'tok', no attentional pooling, no patch dropout): conv1, class_embedding,
    positional_embedding, ln_pre, transformer, ln_post, proj."""

    def __init__(self, arch="ViT-H-14"):
        super().__init__()
        a = _arch(arch)
        v = a["vision"]
        self.cfg = v
        W, P = v["width"], v["patch_size"]
        self.heads = W // v["head_width"]
        self.grid = v["image_size"] // P
        self.conv1 = _Affine(W, 3, P, P, bias=False)
        self.class_embedding = _param(W)
        self.positional_embedding = _param(self.grid ** 2 + 1, W)
        self.ln_pre = _Affine(W)
        self.transformer = _Transformer(W, v["layers"], int(W * v["mlp_ratio"]))
        self.ln_post = _Affine(W)
        self.proj = _param(W, a["embed_dim"])
        self.k_pad = -(-3 * P * P // 64) * 64                       # the patch matrix's K, padded to a GEMM width

    def forward(self, img, antialias=True):
        """img: fp32 NCHW in [-1, 1], any size -> pooled embedding [B, embed_dim] (preprocess + VisionTransformer.forward)."""
        v = self.cfg
        dev = img.device
        pk = self._blocks_on(dev, self.transformer.resblocks)
        if not pk.extra:
            f = lambda p: p.detach().to(device=dev, dtype=F32).contiguous()   # noqa: E731
            W = v["width"]
            wc = torch.zeros(W, self.k_pad)
            wc[:, :3 * v["patch_size"] ** 2] = self.conv1.weight.detach().to(F32).reshape(W, -1)
            pk.extra = dict(conv=X.pack_linear_x(wc, dev), cls=f(self.class_embedding), pos=f(self.positional_embedding),
                            pre=(f(self.ln_pre.weight), f(self.ln_pre.bias)), post=(f(self.ln_post.weight), f(self.ln_post.bias)),
                            proj=X.pack_linear_x(self.proj.detach().t().contiguous(), dev))
            torch.cuda.synchronize(dev)
        B = img.shape[0]
        G2, W = self.grid ** 2, v["width"]
        patches = preprocess_patches(img, v["image_size"], v["patch_size"], self.k_pad, antialias)
        emb = X.linear_x(X.split3(patches), pk.extra["conv"])                                       # conv1 as a GEMM: [B * G2, W]
        x = torch.cat([pk.extra["cls"].expand(B, 1, W), emb.reshape(B, G2, W)], 1) + pk.extra["pos"]   # class token + positions: plumbing
        N = G2 + 1
        x = layernorm(x.reshape(B * N, W).contiguous(), *pk.extra["pre"])
        x = run_blocks(x, pk.blocks, B, N, self.heads, causal=False)
        pooled = layernorm(x.reshape(B, N, W)[:, 0].contiguous(), *pk.extra["post"])              # ln_post is per token: token 0 is all `tok` pooling reads
        return X.linear_x(X.split3(pooled), pk.extra["proj"])


def _bytes_to_unicode():
    """The byte -> printable-character table of the CLIP byte-level BPE (open_clip/tokenizer.py: bytes_to_unicode)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\u00a1"), ord("\u00ac") + 1)) + list(range(ord("\u00ae"), ord("\u00ff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, (chr(c) for c in cs)))


class SimpleTokenizer:
    """open_clip.tokenizer.SimpleTokenizer (2.24.0; OpenAI CLIP's byte-level BPE) over a merges file in the package's format
    (`bpe_simple_vocab_16e6.txt.gz`: a header line, then one merge per line; gzip or plain text).  The file itself is not in this image:
    pass its path (`bpe_path=` of FrozenOpenCLIPEmbedder, or VIDSEG_OPENCLIP_BPE).  Vocabulary: 256 byte symbols, the same with the
    end-of-word mark, one entry per merge (the first 48894 of the file), <start_of_text>, <end_of_text> -- 49408 with the real file.
    Text cleaning: html.unescape twice, whitespace collapsed, lower-cased; ftfy's mojibake repair (absent here) is skipped.
    Pinned against transformers' CLIPTokenizer on a synthetic merges file (tests/test_oracle_openclip.py)."""

    def __init__(self, bpe_path, max_merges=49152 - 256 - 2):
        import gzip
        import regex
        with open(bpe_path, "rb") as fh:
            raw = fh.read()
        text = (gzip.decompress(raw) if raw[:2] == b"\x1f\x8b" else raw).decode("utf-8")
        merges = [tuple(m.split()) for m in text.split("\n")[1:max_merges + 1] if len(m.split()) == 2]
        self.byte_encoder = _bytes_to_unicode()
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab] + ["".join(m) for m in merges] + ["<start_of_text>", "<end_of_text>"]
        self.encoder = {t: i for i, t in enumerate(vocab)}
        self.bpe_ranks = {m: i for i, m in enumerate(merges)}
        self.cache = {"<start_of_text>": "<start_of_text>", "<end_of_text>": "<end_of_text>"}
        self.pat = regex.compile(r"""<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", regex.IGNORECASE)
        self.sot, self.eot = self.encoder["<start_of_text>"], self.encoder["<end_of_text>"]

    def bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        while len(word) > 1:
            pairs = {(a, b) for a, b in zip(word, word[1:])}
            best = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if best not in self.bpe_ranks:
                break
            a, b = best
            merged, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    merged.append(a + b)
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = tuple(merged)
        out = " ".join(word)
        self.cache[token] = out
        return out

    def encode(self, text):
        import html
        import re
        text = re.sub(r"\s+", " ", html.unescape(html.unescape(text)).strip()).strip().lower()
        ids = []
        for tok in self.pat.findall(text):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(tok).split(" "))
        return ids

    def __call__(self, texts, context_length=77):
        """open_clip.tokenize: <start_of_text> ids <end_of_text>, zero-padded; a longer text is cut and ends with <end_of_text>."""
        texts = [texts] if isinstance(texts, str) else list(texts)
        out = torch.zeros((len(texts), context_length), dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > context_length:
                ids = ids[:context_length]
                ids[-1] = self.eot
            out[i, :len(ids)] = torch.tensor(ids)
        return out


_TOKENIZERS: Dict[str, SimpleTokenizer] = {}


def tokenize(text: Union[str, Sequence[str], torch.Tensor], context_length=77, bpe_path: Optional[str] = None) -> torch.Tensor:
    """open_clip.tokenize.  Token tensors pass through; with a merges file (`bpe_path` or VIDSEG_OPENCLIP_BPE) any text is tokenised by
    SimpleTokenizer; without one only the empty string -- the drivers' prompt -- can be, as [49406, 49407, 0, ...]."""
    import os
    if isinstance(text, torch.Tensor):
        if text.dtype not in (torch.int32, torch.int64) or text.dim() != 2 or text.shape[1] != context_length:
            raise VidsegError(f"tokenize: token tensors are int [B, {context_length}]")
        return text.long()
    texts = [text] if isinstance(text, str) else list(text)
    bpe_path = bpe_path or os.environ.get("VIDSEG_OPENCLIP_BPE")
    if bpe_path:
        if bpe_path not in _TOKENIZERS:
            _TOKENIZERS[bpe_path] = SimpleTokenizer(bpe_path)
        return _TOKENIZERS[bpe_path](texts, context_length)
    if any(t.strip() != "" for t in texts):
        raise VidsegError("tokenize: open_clip's BPE vocabulary (bpe_simple_vocab_16e6.txt.gz) is part of the absent open_clip package; "
                          "pass its path (bpe_path= / VIDSEG_OPENCLIP_BPE), or use the empty prompt (the drivers' default, "
                          "sd_pipeline_vspw.py:35) or pre-tokenised int tensors")
    out = torch.zeros((len(texts), context_length), dtype=torch.long)
    out[:, 0], out[:, 1] = SOT, EOT
    return out


class FrozenOpenCLIPEmbedder(AbstractEmbModel):
    """modules.py:498-567.  `state_dict` (open_clip names, optional) fills the text tower at construction; the released checkpoints'
    `conditioner.embedders.N.model.*` keys arrive through load_state_dict."""
    LAYERS = ["last", "penultimate"]

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True, layer="last",
                 state_dict: Optional[Dict[str, torch.Tensor]] = None, bpe_path: Optional[str] = None):
        super().__init__()
        if layer not in self.LAYERS:
            raise AssertionError(layer)
        self.bpe_path = bpe_path
        self.model = TextTower(arch)
        self.device, self.max_length, self.layer = device, max_length, layer
        self.layer_idx = 0 if layer == "last" else 1
        if state_dict is not None:
            self.model.load_state_dict({k: v for k, v in state_dict.items() if not k.startswith("visual.")}, strict=False)

    def freeze(self):
        return self

    @torch.no_grad()
    def forward(self, text):
        if isinstance(text, torch.Tensor) and text.is_floating_point():
            return text                                               # an embedding computed elsewhere ([B, 77, W]) passes through
        tokens = tokenize(text, self.max_length, self.bpe_path)
        return self.encode_with_transformer(tokens.to(self.device))

    def encode_with_transformer(self, text):
        # the drivers hand one prompt per frame ([prompt] * num_frames, sd_pipeline_vspw.py:536): evaluate each distinct row once
        rows, inverse = torch.unique(text, dim=0, return_inverse=True)
        z = self.model.encode(rows.contiguous(), self.layer_idx)
        return z if rows.shape[0] == text.shape[0] and bool((inverse == torch.arange(text.shape[0], device=text.device)).all()) else z[inverse]

    def encode(self, text):
        return self(text)

    def load_state_dict(self, state_dict, strict=False, assign=False):
        sub = {k[len("model."):]: v for k, v in state_dict.items() if k.startswith("model.") and not k.startswith("model.visual.")
               and k != "model.attn_mask"}
        r = self.model.load_state_dict(sub, strict=False)
        unexpected = [k for k in state_dict if not k.startswith("model.")] + ["model." + k for k in r.unexpected_keys]
        return ["model." + k for k in r.missing_keys], unexpected


class _VisualModel(nn.Module):
    def __init__(self, arch):
        super().__init__()
        self.visual = VisualTower(arch)


class FrozenOpenCLIPImageEmbedder(AbstractEmbModel):
    """modules.py:570-728 (inference subset: no crops, no token output, ucg_rate 0)."""

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True, antialias=True, ucg_rate=0.0,
                 unsqueeze_dim=False, repeat_to_max_len=False, num_image_crops=0, output_tokens=False, init_device=None,
                 state_dict: Optional[Dict[str, torch.Tensor]] = None):
        super().__init__()
        if num_image_crops or output_tokens or ucg_rate:
            raise NotImplementedError("FrozenOpenCLIPImageEmbedder: crops / token output / ucg dropout are not on the inference path")
        self.model = _VisualModel(arch)
        self.device, self.max_length, self.antialias = device, max_length, antialias
        self.unsqueeze_dim, self.repeat_to_max_len = unsqueeze_dim, repeat_to_max_len
        self.ucg_rate = ucg_rate
        if state_dict is not None:
            self.model.visual.load_state_dict({k[len("visual."):]: v for k, v in state_dict.items() if k.startswith("visual.")}, strict=False)

    def freeze(self):
        return self

    @torch.no_grad()
    def forward(self, image, no_dropout=False):
        z = self.encode_with_vision_transformer(image).to(image.dtype)
        if self.unsqueeze_dim:
            z = z[:, None, :]
        if self.repeat_to_max_len:
            z_ = z[:, None, :] if z.dim() == 2 else z
            return z_.expand(-1, self.max_length, -1).contiguous(), z
        return z

    def encode_with_vision_transformer(self, img):
        return self.model.visual(img.to(F32), antialias=self.antialias)

    def encode(self, text):
        return self(text)

    def load_state_dict(self, state_dict, strict=False, assign=False):
        sub = {k[len("model.visual."):]: v for k, v in state_dict.items() if k.startswith("model.visual.")}
        r = self.model.visual.load_state_dict(sub, strict=False)
        # the image embedder's checkpoints still carry the text half's small tensors (positional_embedding, text_projection, ..): not ours
        return ["model.visual." + k for k in r.missing_keys], ["model.visual." + k for k in r.unexpected_keys]


class FrozenOpenCLIPImagePredictionEmbedder(AbstractEmbModel):
    """modules.py:1028-1046."""

    def __init__(self, open_clip_embedding_config: Optional[Dict] = None, n_cond_frames: int = 1, n_copies: int = 1):
        super().__init__()
        self.n_cond_frames, self.n_copies = n_cond_frames, n_copies
        cfg = open_clip_embedding_config or {"target": "sgm.modules.encoders.modules.FrozenOpenCLIPImageEmbedder"}
        if isinstance(cfg, nn.Module):
            self.open_clip = cfg
        elif cfg.get("target", "").endswith("FrozenOpenCLIPImageEmbedder"):
            self.open_clip = FrozenOpenCLIPImageEmbedder(**cfg.get("params", {}))
        else:
            self.open_clip = instantiate_from_config(cfg)

    @torch.no_grad()
    def forward(self, vid):
        if vid.dim() == 3:
            return vid                                                # an embedding computed elsewhere ([B, t, d]) passes through
        z = self.open_clip(vid)
        z = z.reshape(-1, self.n_cond_frames, z.shape[-1])                           # "(b t) d -> b t d"
        return z[:, None].expand(-1, self.n_copies, -1, -1).reshape(-1, self.n_cond_frames, z.shape[-1]).contiguous()   # "b t d -> (b s) t d"

    def load_state_dict(self, state_dict, strict=False, assign=False):
        sub = {k[len("open_clip."):]: v for k, v in state_dict.items() if k.startswith("open_clip.")}
        miss, unex = self.open_clip.load_state_dict(sub, strict=False)
        return ["open_clip." + k for k in miss], ["open_clip." + k for k in unex] + [k for k in state_dict if not k.startswith("open_clip.")]
