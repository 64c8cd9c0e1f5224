"""Deterministic synthetic inputs for the hot path (tests, golden fixtures, bench).

The reference ships no test data for this path, so every workload is
synthetic.  Everything here is built from numpy's PCG64 stream and the four IEEE
basic operations only (no exp/sin), so the bytes are identical on every host; each
generator also returns a sha256 so fixtures can detect drift.

Shapes follow the reference's dump format (scripts/sampling/sd_pipeline_vspw.py:103-120):
a dumped attention tensor is ``[2F, N, C]`` with the unconditional half first
(sgm/modules/diffusionmodules/guiders.py:33-42).
"""
from __future__ import annotations

import hashlib

import numpy as np


def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def sha256_of(*arrays: np.ndarray) -> str:
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def blob_weights(num_frames: int, h: int, w: int, num_blobs: int, seed: int) -> np.ndarray:
    """Soft membership of every cell in `num_blobs` drifting blobs + background.

    Returns float64 [F, h*w, num_blobs+1], rows sum to 1.  Rational falloff so that
    only + - * / are used.
    """
    g = _rng(seed)
    cy = g.uniform(0.15, 0.85, size=num_blobs) * h
    cx = g.uniform(0.15, 0.85, size=num_blobs) * w
    vy = g.uniform(-0.25, 0.25, size=num_blobs)
    vx = g.uniform(-0.25, 0.25, size=num_blobs)
    rad = g.uniform(0.12, 0.28, size=num_blobs) * min(h, w)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    out = np.empty((num_frames, h * w, num_blobs + 1), dtype=np.float64)
    for t in range(num_frames):
        wts = [np.full(h * w, 0.35)]
        for b in range(num_blobs):
            d2 = (yy - (cy[b] + vy[b] * t)) ** 2 + (xx - (cx[b] + vx[b] * t)) ** 2
            q = d2 / (rad[b] * rad[b])
            wts.append((1.0 / (1.0 + q * q * q)).reshape(-1))
        wt = np.stack(wts, axis=-1)
        out[t] = wt / wt.sum(axis=-1, keepdims=True)
    return out


def attention_q_dumps(
    num_frames: int,
    h: int,
    w: int,
    channels: int,
    num_blocks: int = 3,
    num_blobs: int = 6,
    seed: int = 1,
    noise: float = 0.15,
    scale: float = 3.0,
):
    """Synthetic stand-ins for the dumped decoder-block self-attention queries.

    Returns (list of `num_blocks` float16 arrays [2F, h*w, C], sha256).  The
    conditional half (rows F:) carries cluster + position structure so that
    K-means, KNN and dense tracking are all non-degenerate; the unconditional
    half is an independent draw (it must be ignored by the analysis path,
    feature_extraction.py:550-551).
    """
    g = _rng(seed)
    wts = blob_weights(num_frames, h, w, num_blobs, seed + 1000)          # [F, N, G+1]
    protos = g.standard_normal((num_blobs + 1, channels))
    pos_basis = g.standard_normal((3, channels)) * 0.6
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64) / h, np.arange(w, dtype=np.float64) / w, indexing="ij")
    pos = np.stack([yy.reshape(-1), xx.reshape(-1), (yy * xx).reshape(-1)], axis=-1) @ pos_basis  # [N, C]
    base = wts @ protos + pos[None]                                                              # [F, N, C]
    blocks = []
    for _ in range(num_blocks):
        gain = 1.0 + 0.1 * g.standard_normal(channels)
        cond = (base * gain + noise * g.standard_normal(base.shape)) * scale
        uncond = g.standard_normal(base.shape) * scale
        blocks.append(np.concatenate([uncond, cond], axis=0).astype(np.float16))
    return blocks, sha256_of(*blocks)


def latent_clip(num_frames: int, h: int, w: int, seed: int = 1, channels: int = 4) -> np.ndarray:
    """Seeded smooth latent field with the blob motion, float32 [F, C, h, w] (VAE excluded,
    SURVEY.md §8(d)): 0.18215-scaled like encode_first_stage output (sd_pipeline_vspw.py:258-260)."""
    g = _rng(seed)
    wts = blob_weights(num_frames, h, w, 6, seed + 2000)                   # [F, N, 7]
    protos = g.standard_normal((7, channels)) * 4.0
    lat = (wts @ protos).reshape(num_frames, h, w, channels)
    lat = lat + 0.3 * g.standard_normal(lat.shape)
    return np.ascontiguousarray((lat * 0.18215 * 4.0).transpose(0, 3, 1, 2)).astype(np.float32)


def region_labels(num_frames: int, h: int, w: int, num_regions: int, seed: int) -> np.ndarray:
    """Ground-truth partition of every frame into `num_regions` drifting Voronoi cells, int64 [F, h, w].
    Sites start on a jittered gy x gx grid (balanced cell areas) and drift by at most half a cell over the clip."""
    g = _rng(seed)
    gy = int(np.floor(np.sqrt(num_regions * h / w) + 0.5))
    gy = max(1, min(gy, num_regions))
    while num_regions % gy:
        gy -= 1
    gx = num_regions // gy
    jy, jx = g.uniform(-0.2, 0.2, size=num_regions), g.uniform(-0.2, 0.2, size=num_regions)
    vy, vx = g.uniform(-0.5, 0.5, size=num_regions), g.uniform(-0.5, 0.5, size=num_regions)
    r = np.arange(num_regions)
    cy = ((r // gx) + 0.5 + jy) * (h / gy)
    cx = ((r % gx) + 0.5 + jx) * (w / gx)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64) + 0.5, np.arange(w, dtype=np.float64) + 0.5, indexing="ij")
    out = np.empty((num_frames, h, w), dtype=np.int64)
    for t in range(num_frames):
        f = t / max(num_frames, 1)
        d2 = (yy[..., None] - (cy + vy * f * (h / gy))) ** 2 + (xx[..., None] - (cx + vx * f * (w / gx))) ** 2
        out[t] = np.argmin(d2, axis=-1)
    return out


def block_labels(num_frames: int, h: int, w: int, num_regions: int, seed: int, align: int = 4) -> np.ndarray:
    """Partition into a gy x gx grid of rectangles whose edges sit on multiples of `align` latent pixels (seeded jitter of the
    cut positions), the whole pattern sliding right by `align` pixels every fourth frame (wrapping): int64 [F, h, w].  With an
    even `align` no 2 x 2-pixel token of the UNet's first attention level ever straddles two regions, so no token is an exact
    half-and-half mixture of two prototypes -- the tokens whose cluster a rounding error decides."""
    g = _rng(seed)
    gy = int(np.floor(np.sqrt(num_regions * h / w) + 0.5))
    gy = max(1, min(gy, num_regions))
    while num_regions % gy:
        gy -= 1
    gx = num_regions // gy

    def cuts(n, parts):
        base = np.round(np.arange(1, parts) * (n / parts) / align).astype(np.int64)
        jit = g.integers(-1, 2, size=parts - 1)
        c = np.clip(base + jit, 1, n // align - 1) * align
        return np.concatenate([[0], np.maximum.accumulate(c), [n]])

    cy, cx = cuts(h, gy), cuts(w, gx)
    row = np.searchsorted(cy, np.arange(h), side="right") - 1
    col = np.searchsorted(cx, np.arange(w), side="right") - 1
    base = (np.clip(row, 0, gy - 1)[:, None] * gx + np.clip(col, 0, gx - 1)[None, :]).astype(np.int64)
    return np.stack([np.roll(base, align * (t // 4), axis=1) for t in range(num_frames)])


def region_prototypes(num_regions: int, channels: int, seed: int, min_dist: float = 1.2, kind: str = "random") -> np.ndarray:
    """`num_regions` prototype vectors, float64 [R, channels].  kind="random": unit-scale normal draws with pairwise Euclidean
    distance >= min_dist (rejection sampling on the seeded stream).  kind="lattice" (channels = 4, R <= 24): a seeded choice
    among the 16 vertices (+-1, +-1, +-1, +-1) and the 8 axis points +-2 e_i -- every point has norm 2 (rms 1 per channel) and
    every pair is at least 2 apart, the densest such set in four dimensions (the 24-cell)."""
    g = _rng(seed)
    if kind == "lattice":
        if channels != 4 or num_regions > 24:
            raise ValueError("lattice prototypes: 4 channels, at most 24 regions")
        pts = [[a, b, c, d] for a in (-1, 1) for b in (-1, 1) for c in (-1, 1) for d in (-1, 1)]
        pts += [[2 * sgn if i == ax else 0 for i in range(4)] for ax in range(4) for sgn in (-1, 1)]
        return np.asarray(pts, dtype=np.float64)[g.permutation(24)[:num_regions]]
    protos = []
    while len(protos) < num_regions:
        p = g.standard_normal(channels)
        if all(np.sqrt(((p - q) ** 2).sum()) >= min_dist for q in protos):
            protos.append(p)
    return np.stack(protos)


def region_clip(num_frames: int, h: int, w: int, num_regions: int = 20, seed: int = 1, channels: int = 4, amp: float = 1.5,
                noise: float = 0.05, protos: str = "random", layout: str = "voronoi") -> np.ndarray:
    """Seeded piecewise-constant latent: `num_regions` drifting Voronoi cells, each with its own well-separated prototype
    vector, + white noise; float32 [F, C, h, w] at the scale of an encode_first_stage output (std ~ 1).  The headline
    workload (bench.py, tests at BASELINE configs[1]) uses it with num_regions = the number of masks, so that the K-means
    problem Steps 3-3b solve has K natural clusters (a clip of K objects) instead of over-segmenting a 6-blob scene, whose
    partition is decided by the last bits of the features."""
    lab = (block_labels if layout == "blocks" else region_labels)(num_frames, h, w, num_regions, seed + 4000)
    pv = region_prototypes(num_regions, channels, seed + 5000, kind=protos)
    g = _rng(seed)
    lat = pv[lab] * amp + noise * g.standard_normal((num_frames, h, w, channels))
    return np.ascontiguousarray(lat.transpose(0, 3, 1, 2)).astype(np.float32)


# The headline workload (bench.py, tests/golden/c2_window.npz, tests at BASELINE configs[1]): a clip of K = 20 objects -- every
# frame shows the same 20 drifting regions, so the frame-0-anchored clustering of Steps 3-3b (K-means over the window, labels
# propagated from frame 0, feature_extraction.py:546-643) has K natural clusters -- through a network near the reference's own
# initialisation (zero_gain): with the generic random network the K = 20 partition of the 6-blob clip was decided by the last
# bits of the taps (16 % of the tokens moved under an fp16-rounding-level perturbation; tools/lab/cond_probe.py measures this).
HEADLINE = dict(num_regions=20, amp=2.0, noise=0.05, zero_gain=0.3, protos="lattice")


def headline_latent(num_frames: int, h: int, w: int, window_id: int = 0) -> np.ndarray:
    return region_clip(num_frames, h, w, num_regions=HEADLINE["num_regions"], seed=1 + window_id, amp=HEADLINE["amp"],
                       noise=HEADLINE["noise"], protos=HEADLINE["protos"])


def headline_partition(num_frames: int, h: int, w: int, window_id: int = 0) -> np.ndarray:
    """The generating partition of `headline_latent` on the token grid (latent / 2): int64 [F, (h/2)*(w/2)]."""
    lab = region_labels(num_frames, h, w, HEADLINE["num_regions"], 1 + window_id + 4000)
    return lab[:, ::2, ::2].reshape(num_frames, -1)


def scene_labels(num_frames: int, h: int, w: int, num_objects: int, cells: int, seed: int) -> np.ndarray:
    """Object id of every latent pixel, int64 [F, h, w]: each frame is split into `cells` drifting Voronoi cells (see
    `region_labels`); cell j shows object j * (num_objects // cells) + (f * (num_objects // cells)) // F, i.e. every cell cycles
    through its own num_objects / cells objects over the clip, each staying a few consecutive frames."""
    if num_objects % cells:
        raise ValueError("num_objects must be a multiple of cells")
    per = num_objects // cells
    cell = region_labels(num_frames, h, w, cells, seed)
    f = np.arange(num_frames)[:, None, None]
    return cell * per + (f * per) // max(num_frames, 1)


def scene_clip(num_frames: int, h: int, w: int, num_objects: int = 20, cells: int = 4, seed: int = 1, channels: int = 4,
               amp: float = 1.5, noise: float = 0.05) -> np.ndarray:
    """Piecewise-constant latent of a scene with `num_objects` objects of which `cells` are visible per frame as large regions
    (`scene_labels`), each object with its own well-separated prototype vector, + white noise; float32 [F, C, h, w]."""
    lab = scene_labels(num_frames, h, w, num_objects, cells, seed + 4000)
    protos = region_prototypes(num_objects, channels, seed + 5000)
    g = _rng(seed)
    lat = protos[lab] * amp + noise * g.standard_normal((num_frames, h, w, channels))
    return np.ascontiguousarray(lat.transpose(0, 3, 1, 2)).astype(np.float32)


def sd_conditioning(num_frames: int, context_dim: int = 1024, seq: int = 77, seed: int = 1):
    """`c["crossattn"]` ~ N(0,1) [F, seq, ctx], `uc` zeros (force_uc_zero_embeddings,
    sd_pipeline_vspw.py:299-305); float32."""
    g = _rng(seed + 3000)
    c = g.standard_normal((1, seq, context_dim)).astype(np.float32)
    c = np.repeat(c, num_frames, axis=0)
    return c, np.zeros_like(c)


ZERO_INIT_MARKERS = (".out_layers.3.", ".proj_out.")      # openaimodel.py:306-314 (ResBlock out conv), :406 / attention.py:880-886


def fill_state_dict(shapes: dict, seed: int = 1234, gain: float = 1.0, zero_gain: float = 1.0) -> dict:
    """Deterministic synthetic weights for a state dict given {name: shape}.

    Every tensor gets its own PCG64 stream keyed by (seed, sha256(name)), so the values do not depend
    on dict order or on which other tensors exist.  Matrices / conv kernels ~ N(0, gain/sqrt(fan_in))
    -- including the reference's zero-initialised modules (openaimodel.py:306-314, :828;
    attention.py:880-886), which would otherwise silence every residual branch -- 1-D ".weight"
    (norm scales) ~ 1 + 0.1 N(0,1), biases ~ 0.05 N(0,1).  `zero_gain` scales the weights AND biases of exactly those
    zero-initialised modules (the output conv of every ResBlock and `proj_out` of every transformer): 1.0 is the generic
    random network, a small value is a network near the reference's own initialisation, whose residual branches perturb
    the skip path instead of replacing it.  Same random numbers for every zero_gain.  Returns float32 numpy arrays.
    """
    out = {}
    for name in sorted(shapes):
        shape = tuple(int(s) for s in shapes[name])
        key = int.from_bytes(hashlib.sha256(name.encode()).digest()[:8], "little")
        g = np.random.Generator(np.random.PCG64([seed, key]))
        n = g.standard_normal(shape, dtype=np.float32)
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            n *= np.float32(gain / np.sqrt(fan_in))
        elif name.endswith("weight"):
            n = np.float32(1.0) + np.float32(0.1) * n
        else:
            n *= np.float32(0.05)
        if zero_gain != 1.0 and any(m in "." + name for m in ZERO_INIT_MARKERS):
            n *= np.float32(zero_gain)
        out[name] = n
    return out


def state_dict_signature(shapes: dict) -> str:
    h = hashlib.sha256()
    for name in sorted(shapes):
        h.update(name.encode())
        h.update(str(tuple(int(s) for s in shapes[name])).encode())
    return h.hexdigest()


SD21_NARROW = dict(in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1], num_res_blocks=2,
                   channel_mult=[1, 2, 4, 4], num_head_channels=64, use_linear_in_transformer=True, transformer_depth=1,
                   context_dim=64)
SD21_FULL = dict(in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
                 channel_mult=[1, 2, 4, 4], num_head_channels=64, use_linear_in_transformer=True, transformer_depth=1,
                 context_dim=1024)                                     # configs/inference/sd_2_1.yaml:19-30
SVD_NARROW = dict(in_channels=8, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1], num_res_blocks=2,
                  channel_mult=[1, 2, 4, 4], num_head_channels=64, use_linear_in_transformer=True, transformer_depth=1,
                  context_dim=64, adm_in_channels=64, num_classes="sequential", extra_ff_mix_layer=True, use_spatial_context=True,
                  merge_strategy="learned_with_images", video_kernel_size=[3, 1, 1])
SVD_FULL = dict(in_channels=8, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
                channel_mult=[1, 2, 4, 4], num_head_channels=64, use_linear_in_transformer=True, transformer_depth=1,
                context_dim=1024, adm_in_channels=768, num_classes="sequential", extra_ff_mix_layer=True, use_spatial_context=True,
                merge_strategy="learned_with_images", video_kernel_size=[3, 1, 1])   # configs/inference/svd.yaml:16-34


# ResBlocks whose in_layers_features / out_layers_features (openaimodel.py:349-350, 367-368) the goldens hold (tools/gen_golden_unet.py)
RESBLOCK_FEATURE_PROBES = ("input_blocks.1.0", "input_blocks.4.0", "middle_block.0", "middle_block.2", "output_blocks.2.0", "output_blocks.11.0")
