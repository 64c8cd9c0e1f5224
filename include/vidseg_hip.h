/* libvidseg_hip.so -- C ABI of the MI355X (gfx950) VidSeg hot path.
 *
 * The reference (QianWangX/VidSeg_diffusion) is 100 % Python and has no FFI; its replaceable seam is the
 * `instantiate_from_config` plug-in mechanism (sgm/util.py:168-185).  This header is therefore the NEW boundary
 * the Python host mirror (vidseg_diffusion_amd/*.py, same class/function names and signatures as the reference)
 * binds with ctypes; each entry point cites the reference code whose arithmetic it replaces
 * (paths relative to the reference root; FE = scripts/sampling/feature_extraction.py,
 * OAI = sgm/modules/diffusionmodules/openaimodel.py, ATT = sgm/modules/attention.py,
 * SAM = sgm/modules/diffusionmodules/sampling.py, DU = sgm/modules/diffusionmodules/util.py).
 *
 * Conventions: plain pointers and sizes only (device pointers unless said otherwise); every buffer is
 * caller-owned, nothing is allocated or freed; calls are asynchronous on `stream`; the return value is 0 or a
 * negative error code (VIDSEG_ERR_*), with a message available from vidseg_last_error() (thread-local);
 * no exceptions cross the boundary; no global mutable state except the opt-in GEMM profiler.
 * Layouts: activations NHWC in the 16-bit activation type "a16" (IEEE fp16 in the default build, see vidseg_act_is_fp16; tokens [B][H*W][C] == images [B][H][W][C]); dumped features fp16 [2F][N][C]
 * with the unconditional half first (sgm/modules/diffusionmodules/guiders.py:33-42); labels int32.
 */
#ifndef VIDSEG_HIP_H
#define VIDSEG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* vidseg_stream_t; /* hipStream_t */

#define VIDSEG_OK 0
#define VIDSEG_ERR_ARG (-1)
#define VIDSEG_ERR_HIP (-2)
#define VIDSEG_ERR_UNSUPPORTED (-3)

int vidseg_version(void);
const char* vidseg_last_error(void);
/* 16-bit storage format of every `*_a16` entry point's activation / weight arguments ("a16" = the build's 16-bit activation
 * type): 1 = IEEE fp16 (default build; the reference's CUDA autocast dtype), 0 = bfloat16 (-DVIDSEG_ACT_BF16).  Parameter names
 * and comments that say "a16" mean that 16-bit type. */
int vidseg_act_dtype(void);

/* ------------------------------------------------------------------------------------------------------
 * Post-UNet analysis
 * ---------------------------------------------------------------------------------------------------- */

/* FE:739-748 torch.mean(torch.stack(blocks)) in fp16 (fp32 accumulate, one rounding) fused with
 * FE:550-555 / FE:38-45: keep rows [row0,row0+rows) (the conditional half) and divide every token by its
 * max |.| over channels in fp16.  `blocks` is a HOST array of nblk device pointers to fp16 [*][C]. */
int vidseg_mean_normalize_f16(const void* const* blocks, int nblk, int64_t row0, int64_t rows, int C, void* out_mean /*opt*/,
                              void* out_norm, vidseg_stream_t stream);

/* sklearn KMeans.fit preamble (cluster/_kmeans.py:1478-1485, :279-287) on fp16 x [n][C] up-cast to float64:
 * column mean (numpy axis-0 order), squared row norms of the centred data, per-feature variances. */
int vidseg_kmeans_prepare(const void* x16, int64_t n, int C, double* mean, double* xsq, double* colvar /*opt*/,
                          double* scratch /* 2*ceil(n/256)*C */, vidseg_stream_t stream);
int vidseg_row_sqnorm_f64(const void* x16, int64_t n, int C, double* xsq, vidseg_stream_t stream);

/* One k-means++ round (cluster/_kmeans.py:174-274) for all R restarts at once; the uniforms are drawn on the
 * host from numpy's RandomState in sklearn's order (FE:562 relies on np.random.seed, sd_pipeline_vspw.py:619-623).
 * c = 0 evaluates the first centres (cand[r] preset, closest = +inf); c = 1..K-1 finalises centre c-1 (first
 * minimum potential over the trials) and draws/evaluates Tnext candidates; c = K only finalises.
 * cand: int32 [2][R * Tmax] -- round c reads the candidates of round c - 1 in half (c - 1) & 1 (round 0: the caller's first picks
 * cand[r] in half 0) and writes its own into half c & 1 (one buffer would let a restart's block overwrite candidates another
 * restart's block has yet to read).  cand_len = the int32 count the caller allocated, checked against 2 * R * Tmax: the
 * _v2 name + this argument replace vidseg_kpp_round, whose callers allocated one half only. */
int vidseg_kpp_round_v2(const void* x16, const double* mean, const double* xsq, int64_t n, int C, int R, int K, int c, int Tprev,
                        int Tnext, int Tmax, const double* u, int ustride, double* closest, double* dcand, double* part,
                        double* pot, int32_t* cand, int64_t cand_len, int32_t* center_ids, vidseg_stream_t stream);
int vidseg_gather_rows_f64(const void* x16, const double* mean, int C, const int32_t* ids, int J, double* out,
                           vidseg_stream_t stream);

/* One Lloyd iteration (cluster/_k_means_lloyd.pyx lloyd_iter_chunked_dense) for the restarts in bit mask
 * `active`: E-step argmin_j(|c_j|^2 - 2 x.c_j) (labels in place, changed[r] += #changed), M-step with
 * fixed-order partial sums, centres in place, shift2[r][k] = |c_new - c_old|^2, counts[r][k]. */
int vidseg_lloyd_iter(const void* x16, const double* mean /*opt*/, int64_t n, int C, int R, int K, const unsigned* d_active,
                      const int32_t* slots, int nslots, const int32_t* colrow, int update_centers, double* centers, double* cnorm, int32_t* labels, int32_t* changed, double* psum,
                      int32_t* pcnt, int chunk, double* shift2, int32_t* counts, vidseg_stream_t stream);
/* _kmeans.py:715-734 on the device: state = {active mask, strict mask, error flags, n_iter[R]} (uint32) */
int vidseg_lloyd_status(int R, int K, int it, double tol, int32_t* changed, const double* shift2, const int32_t* counts,
                        unsigned* state, vidseg_stream_t stream);
/* One ACCELERATED Lloyd iteration + the convergence bookkeeping of vidseg_lloyd_status (same labels as
 * vidseg_lloyd_iter by construction): triangle-inequality bound filter (ub/lb per restart and sample), E-step on the
 * gathered list of unsettled samples, M-step on exact incrementally-maintained raw sums (float64 adds of fp16 values
 * are exact), centre = (sum - count*mean)/count.  Replaces _k_means_lloyd.pyx:lloyd_iter_chunked_dense + _kmeans.py:715-734.
 * Empty clusters are relocated on the device (_k_means_common.pyx:167-211, numpy's argpartition replayed).
 * Before it = 0 the host sets labels = -1, sums = 0, counts = 0, list[r] = 0..n-1, nlist[r] = n, changed = 0.
 * chg: int32 [R][n][2] change list, delta: [R][K], dtop: [R][3]; relocation scratch reloc_d f64 [R][n],
 * reloc_t i32 [R][n], reloc i32 [R][64][3], nreloc i32 [R]. */
int vidseg_lloyd_step(const void* x16, const double* mean, const double* xsq, int64_t n, int C, int R, int K, int it, double tol,
                      unsigned* state, const int32_t* slots, int nslots, double* centers, double* cnorm, double* sums, int32_t* counts,
                      int32_t* labels, double* ub, double* lb, int32_t* list, int32_t* nlist, int32_t* chg, int32_t* changed,
                      double* shift2, double* delta, double* dtop, double* reloc_d, int32_t* reloc_t, int32_t* reloc, int32_t* nreloc,
                      vidseg_stream_t stream);
int vidseg_kmeans_inertia(const void* x16, const double* mean, int64_t n, int C, int R, int K, const double* centers,
                          const int32_t* labels, double* part, double* inertia, vidseg_stream_t stream);
int vidseg_add_mean_f64(double* centers, const double* mean, int K, int C, vidseg_stream_t stream);

/* FE:608-613 KNeighborsClassifier(4).fit(ref).predict(q): brute-force float64 |q|^2 - 2 q.y + |y|^2 (what
 * sklearn does for fp16 storage), 4 smallest, mode of their labels (smallest label on ties).
 * knn_top4 + vote4 are the same classifier split at the label dependency (multi-GPU chain). */
int vidseg_knn_vote(const void* q16, int64_t nq, const void* ref16, int64_t nref, int C, const double* qq, const double* yy,
                    const int32_t* ref_labels, int32_t* out, vidseg_stream_t stream);
int vidseg_knn_top4(const void* q16, int64_t nq, const void* ref16, int64_t nref, int C, const double* qq, const double* yy,
                    int32_t* out_idx, vidseg_stream_t stream);
int vidseg_vote4(const int32_t* idx, const int32_t* ref_labels, int64_t nq, int32_t* out, vidseg_stream_t stream);

/* FE:272-274: x / torch.norm(x) on fp16 rows, applied 1..nb times (the reference re-normalises target/aux maps
 * inside its 500-query batch loop): out[b][row][:] = row normalised b+1 times. */
int vidseg_track_normalize(const void* x16, int64_t rows, int C, int nb, void* out, vidseg_stream_t stream);
/* FE:218-300 for one frame pair f -> f+1: cosine maps against frame f+1 and frame 0 (exact dot, one fp16
 * rounding), fp16 blend f/(f+1)*cos + 1/(f+1)*cos_aux (FE:290-291), row arg-max with numpy's
 * argpartition(-1) tie behaviour (FE:293-296).  cur/next: flat cell index per track. */
int vidseg_track_step(const void* normed, int F, int N, int w, int C, int f, int batch, const int32_t* cur, int use_aux,
                      void* blend, int32_t* next, int32_t* tie_rows /*opt*/, vidseg_stream_t stream);
/* FE:392-421: signed-jump spatial filter, Counter.most_common(1) vote along each trajectory, write-back with
 * the last writer (largest point index) winning. */
int vidseg_trajectory_vote(const int32_t* idx, const int32_t* labels, int F, int N, int w, int spatial_filter, int32_t* common,
                           int32_t* winner, int32_t* out, vidseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * UNet operators
 * ---------------------------------------------------------------------------------------------------- */

/* nn.Linear / 1x1 conv (ATT:274-280, 862, 886; OAI:317-324) as a 16-bit (a16) MFMA GEMM: out = act(cat(a0,a1) @ w^T +
 * bias + rowvec[sample]) + residual.  act: 0 none, 1 SiLU (OAI:605-609 time_embed), 2 GEGLU (ATT:89-96; w/bias
 * packed in 32-row value|gate groups).  tap/tap2: fp16 copies of output columns [0,tap_cols) / [tap_cols,2*tap_cols)
 * = the q / k dumps of ATT:330-331. */
int vidseg_linear_a16(const void* a0, const void* a1, int C0, int C1, long long M, const void* w, int N, const float* bias,
                       const float* rowvec, int rv_stride, int rows_per_sample, const void* residual, int ldr, void* out,
                       float* out_f32, int ldo, void* tap, void* tap2, int tap_cols, int tap_ld,
                       const float* rowadd /* per-row scalar: modulation lambda*mask[:,None], ATT:646-663, 697-719 */, int act,
                       vidseg_stream_t stream);
/* 3x3 conv, padding 1 (OAI:267-271, 302-315 ResBlock convs; OAI:202-217 Downsample stride 2; OAI:149-167
 * Upsample = nearest x2 folded into the addressing) over the channel concat of x0 and x1 (skip connection,
 * OAI:912), + bias + per-sample emb vector (OAI:353-365) + residual (OAI:369).  pad = 1, or 0 for the first stage's
 * (0,1,0,1)-padded stride-2 Downsample (sgm/modules/diffusionmodules/model.py:84-91); out_f32 (optional) receives an fp32
 * copy of the result (the VAE's conv_out moments).  Weight layout: [Cout][c/64][kh*3+kw][c%64] (channel-chunk-major K order:
 * the 9 taps of a 64-channel chunk are consecutive K-tiles; C0 and C1 are multiples of 64). */
int vidseg_conv3x3_a16(const void* x0, const void* x1, int C0, int C1, int B, int Hin, int Win, int stride, int up,
                        const void* w, int Cout, const float* bias, const float* rowvec, int rv_stride, const void* residual,
                        void* out, int pad, float* out_f32 /*opt*/, vidseg_stream_t stream);
/* The same convolution with an fp16 [M][Cout] copy of the value inside the epilogue: tap_early = 1 after conv + bias, before the
 * per-sample emb vector = ResBlock.in_layers_features (OAI:349-350); tap_early = 0 after them, before the residual =
 * ResBlock.out_layers_features (OAI:367-368).  The reference stashes both on every ResBlock forward ("mid-block spatial features"). */
int vidseg_conv3x3_a16_tap(const void* x0, const void* x1, int C0, int C1, int B, int Hin, int Win, int stride, int up,
                            const void* w, int Cout, const float* bias, const float* rowvec, int rv_stride, const void* residual,
                            void* out, int pad, void* tap_f16, int tap_early, vidseg_stream_t stream);
/* First stage (VAE encoder, model.py:487-600): row softmax of fp32 logits (the single-head dim-512 mid attention runs as
 * GEMM -> softmax -> GEMM, model.py:161-202) and DiagonalGaussianDistribution.sample * scale_factor
 * (distributions.py:24-41, sgm/models/diffusion.py:138-151); moments NHWC [B][HW][2Z] fp32, noise / out NCHW [B][Z][HW]. */
int vidseg_softmax_rows_a16(const float* x, long long rows, int cols, float scale, void* out_a16, vidseg_stream_t stream);
int vidseg_gaussian_sample(const float* moments_nhwc, const float* noise_nchw, int B, int HW, int Z, float scale, float* out_nchw,
                           vidseg_stream_t stream);
/* OAI:638-644 input conv (Cin 4/8): x fp32 NHWC, w fp32 [3][3][Cin][Cout] -> a16 NHWC. */
int vidseg_conv_in(const float* x, const float* w, const float* bias, int B, int H, int W, int Cin, int Cout,
                   void* out_a16_nhwc, vidseg_stream_t stream);
/* OAI:825-829 output conv (Cout 4): x a16 NHWC, w a16 [4][3][3][Cin] -> fp32 NCHW. */
int vidseg_conv_out4(const void* x, const void* w, const float* bias, int B, int H, int W, int Cin, float* out_f32_nchw,
                     vidseg_stream_t stream);
/* GroupNorm32 (DU:276-278, fp32 statistics; eps 1e-5) / ATT:127 Normalize (eps 1e-6), optional SiLU.
   rows_per_chunk = rows of one sample a block reduces / normalises (the host picks it so that B*ceil(HW/rows_per_chunk) fills the
   chip).  Scratch: part >= B*ceil(HW/rows_per_chunk)*2*C floats (per-chunk partial sums), stats >= B*2*C floats (scale/shift). */
int vidseg_groupnorm_nhwc_a16(const void* x0, const void* x1, int C0, int C1, int B, int HW, int G, const float* gamma,
                               const float* beta, float eps, int silu, int rows_per_chunk, float* part, int part_floats, float* stats,
                               int stats_floats, void* out, vidseg_stream_t stream);
int vidseg_layernorm_a16(const void* x, long long M, int C, const float* gamma, const float* beta, float eps, void* out,
                          vidseg_stream_t stream);
/* ATT:352-356 F.scaled_dot_product_attention per 64-wide head; q/k/v/o are column slices with leading dims. */
int vidseg_attention_a16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, int B,
                          int H, int Nq, int Nk, int head_dim, vidseg_stream_t stream);
/* BASELINE configs[4] (fp8 attention path): the same operator (ATT:352-356) with q, k, v and the probabilities in OCP e4m3
 * (v_mfma_f32_32x32x16_fp8_fp8), fp32 softmax/accumulation, output in the activation dtype; strides in bytes = elements.
 * vidseg_quant_fp8: saturating round-to-nearest-even conversion of n activations (n % 8 == 0) to e4m3 bytes. */
int vidseg_quant_fp8(const void* x, long long n, void* out_fp8, vidseg_stream_t stream);
int vidseg_attention_fp8(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, int B, int H,
                         int Nq, int Nk, int head_dim, vidseg_stream_t stream);
/* AE3DConv.time_mix_conv of the video first stage (sgm/modules/autoencoding/temporal_ae.py:84-107): Conv3d(C -> C, [3,1,1],
 * padding [1,0,0]) over the T frames of each video; x fp32 [(b t)][xC][HW] (first C channels used), w fp32 [C][C][3]. */
int vidseg_time_mix3_f32(const float* x, int BT, int xC, int C, long long HW, int T, const float* w, const float* bias, float* out,
                         vidseg_stream_t stream);
/* Step 5 (scripts/sampling/process_output.py) on decoded frames resident in HBM.
 * vidseg_seg_difference: compute_difference (PO:8-29) for one mask: pos/neg fp32 NCHW [F][3][H][W] decoded frames -> uint8 frames
 *   (SDP:152-168: clamp((x+1)/2,0,1)*255 truncated), wrapped-uint8 channel distance (PO:13), 5x5 Gaussian sigma 3 with reflect-101
 *   borders (PO:15), "L" image (clip + truncate, PO:18) in out_u8 [F][H][W] and its per-frame maximum in fmax_u32 [F].  The
 *   reference's JPEG save / re-load of that image (PO:19, 119) is NOT reproduced (lossy codec).
 * vidseg_seg_argmax: get_seg_map_main (PO:119-161): maps_u8 [K][F][H][W] / (max_u32[K][F] + 1e-5), optional
 *   filter_difference_map (PO:31-40) with weight_u8 [K][F][H][W] (the label's mask image resized to the frame, 0..255) and
 *   filter_s, arg-max over the K masks (first maximum), seg_u8 [F][H][W] = labels[arg]. */
int vidseg_seg_difference(const float* pos, const float* neg, int F, int H, int W, void* out_u8, void* fmax_u32, vidseg_stream_t stream);
/* the same on the uint8 HWC [F][H][W][3] images process_output.py:9-10 reads back from the PNG files of Step 4 */
int vidseg_seg_difference_u8(const void* pos, const void* neg, int F, int H, int W, void* out_u8, void* fmax_u32, vidseg_stream_t stream);
int vidseg_seg_argmax(const void* maps_u8, const void* max_u32, const void* weight_u8, double filter_s, const int* labels, int K, int F,
                      int H, int W, void* seg_u8, vidseg_stream_t stream);
/* SVD (video) operators -- video_model.py:15-89 VideoResBlock, video_attention.py:18-489 */
/* Conv3d kernel [3,1,1], padding [1,0,0] over frames; x NHWC [(b t)][HW][C], w [Cout][c/64][dt][c%64] (video_model.py:45-58) */
int vidseg_conv_temporal3_a16(const void* x, int C, int BT, int HW, int T, const void* w, int Cout, const float* bias,
                               const float* rowvec, int rv_stride, const void* residual, void* out, vidseg_stream_t stream);
/* bias-free projection with fp16 taps in the reference's temporal layout [(b s), t, c] (video_attention.py:152, ATT:330) */
int vidseg_linear_a16_ttap(const void* a0, long long M, int C0, const void* w, int N, void* out, int ldo, void* tap, void* tap2,
                            int tap_cols, int tap_ld, int tap_T, int tap_S, vidseg_stream_t stream);
/* attention across the T frames of each (sample, location), tokens kept in spatial order (video_attention.py:166-195) */
int vidseg_temporal_attention_a16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo,
                                   int Bv, int T, int S, int H, int head_dim, vidseg_stream_t stream);
/* AlphaBlender 'learned_with_images', image_only_indicator == 0 (diffusionmodules/util.py:343-380) */
int vidseg_alpha_blend_a16(const void* x_spatial, const void* x_temporal, const float* mix_factor, long long n, void* out,
                            vidseg_stream_t stream);
/* tokens + frame-index embedding (video_attention.py:417-431) */
int vidseg_add_rowvec_a16(const void* x, const void* vec, long long rows, int C, int rows_per_sample, int nvec, void* out,
                           vidseg_stream_t stream);
/* DU:209-233 */
int vidseg_timestep_embedding(const float* t, int B, int dim, float max_period, void* out, vidseg_stream_t stream);
int vidseg_silu_a16(const void* x, long long n, void* out, vidseg_stream_t stream);
int vidseg_f32_to_a16(const float* x, long long n, void* out, vidseg_stream_t stream);
int vidseg_f16_to_a16(const void* x, long long n, void* out, vidseg_stream_t stream); /* injected fp16 dumps -> a16 operands (a copy in the fp16 build) */

/* ------------------------------------------------------------------------------------------------------
 * Sampler arithmetic on fp32 latents (SURVEY.md rows a2-a6, a17 latent blending)
 * ---------------------------------------------------------------------------------------------------- */
/* denoiser.py:23-46: out = a*sa[row] (+ b*sb[row])  -- x*c_in and net*c_out + x*c_skip */
int vidseg_rows_axpby(const float* a, const float* sa, const float* b, const float* sb, long long n, long long inner,
                      float* out, vidseg_stream_t stream);
/* guiders.py:28-31 / :82-91: x_u + s (x_c - x_u), s constant or per frame */
int vidseg_cfg_combine(const float* x, long long half, long long inner, const float* frame_scale, int num_frames, float scale,
                       float* out, vidseg_stream_t stream);
/* sampling_utils.py:34 + SAM:88, 125-131: x + (x - denoised)/sigma * (sigma_next - sigma) */
int vidseg_euler_update(const float* x, const float* den, const float* sigma, const float* sigma_next, long long n,
                        long long inner, float* out, vidseg_stream_t stream);
/* SAM:133-144 add_noise: (x + e*s) * post */
int vidseg_axpy_f32(const float* x, const float* e, long long n, float s, float post, float* out, vidseg_stream_t stream);
int vidseg_scale_f32(float* x, long long n, float s, vidseg_stream_t stream);
int vidseg_blend_f32(const float* x, const float* y, const float* m, long long n, float* out, vidseg_stream_t stream);
/* SAM:229-250: x = x*m + xt*(1-m), m nearest-upsampled from [F][fh][fw] */
int vidseg_latent_blend(float* x, const float* xt, const float* mask, int F, int C, int h, int w, int fh, int fw,
                        vidseg_stream_t stream);
/* fused forms of the three steps above for a fixed sigma per launch */
int vidseg_prepare_net_input(const float* x, const float* concat_u, const float* concat_c, int F, int C, int Cc, int HW,
                             float c_in, float* out_nhwc, vidseg_stream_t stream);
int vidseg_cfg_euler_step(float* x, const float* net_out, int F, int C, int HW, float c_out, float c_skip,
                          const float* frame_scale, float scale, float sigma, float sigma_next, float* denoised_out,
                          vidseg_stream_t stream);
int vidseg_add_noise(float* x, const float* eps, long long n, float sigma, float inv_scale, vidseg_stream_t stream);

/* binds a caller-owned fp32 device scratch for split-K partials to the launches on `stream` of the current device (optional: a
   stream without a binding runs without split-K; a null pointer unbinds; two streams must not share partials).  The reference has no
   counterpart -- cuBLAS owns its workspaces, sgm/modules/attention.py and openaimodel.py call F.linear / conv2d.  There is no
   process-wide default. */
int vidseg_bind_workspace(vidseg_stream_t stream, float* ws, long long floats);

/* opt-in HIP-event timing of the conv/linear MFMA kernel family (bench.py roofline).  The profiler is a caller-owned handle: between
   _begin(h) and _end(h) the launches of the CALLING THREAD are timed into h; _end: out = {ms, flops, launches} (host) */
int vidseg_gemm_profiler_create(void** handle_out);
int vidseg_gemm_profiler_destroy(void* handle);
int vidseg_gemm_profile_begin(void* handle);
int vidseg_gemm_profile_end(void* handle, double* out);
/* per-kernel split of the region _end closed: out[24] = {ms, flops, launches} x {128x128 LDS-DMA, 256-row big tile, mid tile, 256x64,
   224-row big tile, weight-stationary streaming, 224x320 split-operand tile, 224x256 split-operand GEGLU tile} */
int vidseg_gemm_profile_kinds(void* handle, double* out);
/* algorithmic HBM bytes of the same region per kernel, out[8] (every operand and result once; no split-K partials) */
int vidseg_gemm_profile_bytes(void* handle, double* out);

/* ---- "exact" mode of the UNet path (fp16 build; csrc/exact_ops.hip): every value is carried in fp32 and handed to the 16-bit MFMA
 * GEMM / conv entry points above as the operand image [hi | lo | hi] (hi = fp16(x), lo = fp16(x - hi)) against weights packed as
 * [w_hi | w_hi | w_lo], i.e. one ordinary GEMM over a three-fold K axis with fp32 accumulation = a_hi w_hi + a_lo w_hi + a_hi w_lo.
 * IMAGE FORMAT: rows of 3 C fp16 values, planes [hi | lo | third] at offsets 0, C, 2 C.  For C % 64 == 0 -- every width the GEMM / conv
 * entry points accept -- the third plane is NEVER READ (the split tiles address planes 0, 1; every other GEMM kernel folds the last
 * third of its walk back onto plane 0) and the producers below leave it UNWRITTEN: 4 bytes per value instead of 6 on HBM-bound
 * passes.  For other widths they write [hi | lo | hi].  A caller that wants a full [hi | lo | hi] image copies plane 0 itself.
 * Replaces, at fp32 accuracy, the same reference operators as the 16-bit entry points: GroupNorm32 (+SiLU) DU:276-278 /
 * OAI:267-271, 302-315; LayerNorm ATT:609-759; GEGLU ATT:89-96; scaled-dot-product attention ATT:352-356.  The bf16 build exports
 * the symbols and returns VS_ERR_UNSUPPORTED. */
int vidseg_x_split3(const float* x, long long M, int C, int silu, void* out_f16 /* [M][3C] */, vidseg_stream_t stream);
int vidseg_x_split3_cat(const float* x0 /* [M][C0] */, const float* x1 /* [M][C1] */, long long M, int C0, int C1,
                        void* out_f16 /* [M][3 (C0 + C1)]: split3 of the channel concat */, vidseg_stream_t stream);
int vidseg_x_geglu_split3(const float* y /* [M][2*inner]: value | gate */, long long M, int inner, void* out_f16 /* [M][3*inner] */,
                          vidseg_stream_t stream);
int vidseg_x_groupnorm_split3(const float* x0, const float* x1 /* opt: channel concat */, int C0, int C1, int B, int HW, int G,
                              const float* gamma, const float* beta, float eps, int silu, float* stats /* scratch [B][2][C] */,
                              int stats_floats, double* part /* scratch [B][ceil(HW / rows_per_chunk)][2][C] */, long long part_doubles,
                              void* out_f16 /* [B][HW][3C] */, vidseg_stream_t stream);
int vidseg_x_groupnorm_rows_per_chunk(int HW); /* rows per block of the statistics pass (sizes `part`) */
int vidseg_x_layernorm_split3(const float* x, long long M, int C, const float* gamma, const float* beta, float eps,
                              void* out_f16 /* [M][3C] */, vidseg_stream_t stream);
/* LayerNorm of x[row] + vec[(row / rows_per_sample) % nvec]: the time stack's frame-index embedding (VA:417-431) added on the way into
 * norm_in (VA:155); x_sum (optional) receives the fp32 sum -- the block's residual stream -- so the separate add pass is gone */
int vidseg_x_layernorm_rowvec_split3(const float* x, const float* vec, long long M, int C, int rows_per_sample, int nvec, const float* gamma,
                                     const float* beta, float eps, float* x_sum /* [M][C] or NULL */, void* out_f16 /* [M][3C] */,
                                     vidseg_stream_t stream);
int vidseg_x_attention_f32(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo, int B, int H,
                           int Nq, int Nk, float scale, vidseg_stream_t stream);
/* linear / 3x3 conv of the exact mode with the fp32 residual added in the epilogue (ATT:636-757, 921-927; OAI:369): split operand
 * images in, fp32 out, `residual_f32` [M][ldr] / NHWC [B][Ho][Wo][Cout] or NULL.  These entry points (and
 * vidseg_conv_temporal3_a16_f32) REQUIRE split images -- activation rows [a_hi | a_lo | a_hi] (K = 3 x the layer's width), weight rows
 * [w_hi | w_hi | w_lo] in the usual K order over those 3 x Cin channels: where the 224 x 320 tile is chosen the kernel stages each
 * plane once per 64 original channels and forms a_lo w_hi + a_hi w_hi + a_hi w_lo from them (k_gemm_p7x), i.e. it reads the FIRST
 * TWO planes of the activation image and the FIRST and THIRD of the weight image.  Plain 16-bit operands belong to vidseg_linear_a16
 * / vidseg_conv3x3_a16. */
int vidseg_linear_a16_rf32(const void* a, int K, long long M, const void* w, int N, const float* bias, const float* rowvec, int rv_stride,
                           int rows_per_sample, const float* residual_f32, int ldr, float* out_f32, int ldo, void* tap_f16, void* tap2_f16,
                           int tap_cols, int tap_ld, int act,
                           const float* rowadd /* opt: per-row scalar added to every column before the residual -- Step 4's lambda * mask on
                                                  the attention / feed-forward outputs (ATT:646-663, 697-719, 733-755; VA:197-277) */,
                           vidseg_stream_t stream);
/* the same linear writing the NEXT GEMM's operand image [hi | lo | hi] of its fp32 result instead of the fp32 tensor: for a result whose
 * only consumer is another split-operand GEMM -- the FF output projection feeding proj_out when the transformer has one block
 * (ATT:757 -> :921-927), which otherwise costs an fp32 round trip and a vidseg_x_split3 pass.  Same bits as that pair. */
int vidseg_linear_a16_rf32_x3(const void* a, int K, long long M, const void* w, int N, const float* bias, const float* residual_f32, int ldr,
                              void* out_split3_f16 /* [M][3 N] */, vidseg_stream_t stream);
/* the fused q | k | v projection of a self-attention (ATT:636-650): columns [0, plane_col0) = q leave as fp32 [M][ldo], columns
 * [plane_col0, N) = k | v as vidseg_x_attention_mfma's operand planes (hi = fp16(x), lo = fp16(x - hi), fp16 [M][plane_ld] each) -- the
 * bits vidseg_x_split_planes makes of the fp32 k | v, without their fp32 round trip; q / k taps as in vidseg_linear_a16_rf32 */
int vidseg_linear_a16_qkv_planes(const void* a, int K, long long M, const void* w, int N, const float* bias, float* q_f32, int ldo,
                                 void* kv_hi_f16, void* kv_lo_f16, int plane_col0, int plane_ld, void* tap_f16, void* tap2_f16,
                                 int tap_cols, int tap_ld, vidseg_stream_t stream);
/* GEGLU projection of the exact mode with the product value * gelu_erf(gate) formed in fp32 inside the epilogue and written as the FF
 * output projection's split operand image (ATT:89-96); w / bias interleaved like the 16-bit GEGLU weights */
int vidseg_linear_a16_geglu_x3(const void* a, int K, long long M, const void* w, int N, const float* bias,
                               void* out_split3_f16 /* [M][3 * (N / 2)] */, vidseg_stream_t stream);
/* the same projection on the 224 x 256 split tile (each plane of the operands staged once, k_gemm_p7x<4, true>): w / bias interleaved in
 * 16-ROW value | gate groups instead of 32-row ones; K % 192 == 0 (K = 3 x the layer's width), N % 256 == 0 */
int vidseg_linear_a16_geglu_x3g16(const void* a, int K, long long M, const void* w, int N, const float* bias,
                                  void* out_split3_f16 /* [M][3 * (N / 2)] */, vidseg_stream_t stream);
int vidseg_conv3x3_a16_rf32(const void* x, int C, int B, int Hin, int Win, int stride, int up, const void* w, int Cout, const float* bias,
                            const float* rowvec, int rv_stride, const float* residual_f32, float* out_f32, vidseg_stream_t stream);
/* the same attention on the matrix pipe: three fp16 MFMA products of split operands per contraction (fp32 accuracy).  q fp32; k / v as
 * fp16 hi / lo planes with one row stride (vidseg_x_split_planes makes them from fp32 column blocks) */
int vidseg_x_split_planes(const float* x, int ld, long long rows, int cols, void* hi_f16 /* [rows][cols] */, void* lo_f16 /* [rows][cols] */,
                          vidseg_stream_t stream);
int vidseg_x_attention_mfma(const float* q, int ldq, const void* k_hi, const void* k_lo, const void* v_hi, const void* v_lo, int ldkv,
                            float* out /* fp32 [B][Nq][ldo], or NULL */, void* out_split3 /* f16 [B][Nq][3 ldo] = [hi | lo | hi], or NULL */,
                            int ldo, int B, int H, int Nq, int Nk, float scale, vidseg_stream_t stream);
/* temporal self-attention of the VideoUNet's time stack at fp32 accuracy (sgm/modules/video_attention.py:171-199: rearrange
 * `(b t) s c -> (b s) t c`, attention over the T <= 16 frames of every (video, location), rearrange back; ATT:352-356 per head of 64),
 * read and written in the spatial row order: qkv fp32 rows (b T + t) S + s of `ld` floats holding q | k | v at columns 0 / 64 H / 128 H;
 * result as fp32 [(b t) s][64 H] or as the output projection's split operand image [(b t) s][3 * 64 H]; optional fp16 taps of q / k in the
 * reference's [(b s)][t][c] layout (ATT:330-331).  The permuted copies of the projection and of the result never exist. */
int vidseg_x_temporal_attention(const float* qkv, int ld, int nvid, int T, int S, int H, float scale, float* out_f32 /* or NULL */,
                                void* out_split3_f16 /* or NULL */, void* tap_q_f16 /* opt */, void* tap_k_f16 /* opt */,
                                vidseg_stream_t stream);
/* x + vec[sample % nvec] per row (the frame-index embedding add of SpatialVideoTransformer, VA:417-431), fp32 */
int vidseg_x_add_rowvec_f32(const float* x, const float* vec, long long M, int C, int rows_per_sample, int nvec, float* out,
                            vidseg_stream_t stream);
/* vidseg_conv_temporal3_a16 with the fp32 accumulators (+ bias + per-(b t) vector) stored as they are */
int vidseg_conv_temporal3_a16_f32(const void* x, int C, int BT, int HW, int T, const void* w, int Cout, const float* bias,
                                   const float* rowvec, int rv_stride, float* out_f32, vidseg_stream_t stream);
/* vidseg_linear_a16_rf32 / vidseg_conv_temporal3_a16_f32 followed inside the epilogue by the VideoUNet's AlphaBlender
 * (sgm/modules/diffusionmodules/util.py:343-380, image_only_indicator = 0): out = alpha * blend + (1 - alpha) * (result + residual).
 * The last linear of a time-stack transformer block (video_attention.py:281 -> :470-476) and the second temporal convolution of a
 * VideoResBlock (video_model.py:66-89) write the mixed stream themselves. */
int vidseg_linear_a16_rf32_blend(const void* a, int K, long long M, const void* w, int N, const float* bias, const float* residual_f32, int ldr,
                                 const float* blend_f32 /* [M][N] */, float alpha, float* out_f32 /* [M][N] */, vidseg_stream_t stream);
int vidseg_conv_temporal3_a16_f32_blend(const void* x, int C, int BT, int HW, int T, const void* w, int Cout, const float* bias,
                                         const float* residual_f32, const float* blend_f32, float alpha, float* out_f32, vidseg_stream_t stream);
/* input conv (OAI:638-644) with the fp32 accumulators stored as they are: fp32 NHWC [B][H][W][Cin] -> fp32 NHWC [B][H][W][Cout] */
int vidseg_conv_in_f32(const float* x, const float* w, const float* bias, int B, int H, int W, int Cin, int Cout, float* out_f32_nhwc,
                       vidseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Conditioner: the OpenCLIP ViT-H towers (sgm/modules/encoders/modules.py = MOD; csrc/clip_ops.hip)
 * open_clip_torch 2.24.0's ResidualAttentionBlock / TextTransformer / VisionTransformer, reached through MOD:498-567 (text),
 * MOD:570-728 (image), MOD:1028-1046 (SVD's image prediction embedder).  The towers' projections are vidseg_linear_a16_rf32 on split
 * operand images; these entry points are the rest of their arithmetic, all fp32, a few hundred KB per call (once per clip).
 * ---------------------------------------------------------------------------------------------------- */
/* nn.MultiheadAttention's core: softmax(q k^T * scale [+ causal mask]) v per head of width d (<= 128, d % 4 == 0; ViT-H: 64 text,
 * 80 image); q / k / v fp32 column slices with row strides ld*, head h at columns [d h, d h + d); Nk <= 1024; causal (Nq == Nk):
 * keys above the diagonal are masked (open_clip's build_attention_mask, MOD:546 `attn_mask=self.model.attn_mask`). */
int vidseg_clip_attention_f32(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo, int B, int H,
                              int Nq, int Nk, int d, float scale, int causal, vidseg_stream_t stream);
/* torch.nn.LayerNorm, fp32 in / out (ln_pre, ln_post, ln_final: MOD:548); C % 4 == 0, C <= 2048 */
int vidseg_clip_layernorm_f32(const float* x, long long M, int C, const float* gamma, const float* beta, float eps, float* out,
                              vidseg_stream_t stream);
/* erf GELU of the block's hidden layer (open_clip: mlp.gelu = nn.GELU) written as the next GEMM's operand image [M][3C] (fp16 build) */
int vidseg_clip_gelu_split3(const float* x, long long M, int C, void* out_f16, vidseg_stream_t stream);
/* one axis (1 = x, 0 = y) of kornia.filters.gaussian_blur2d(separable, border_type "reflect") on fp32 [planes][H][W]: the antialias
 * pass of kornia.geometry.resize (kornia 0.7.2, MOD:623-629); taps: device fp32 [ks], ks odd */
int vidseg_clip_blur_axis(const float* x, long long planes, int H, int W, const float* taps, int ks, int axis, float* out, vidseg_stream_t stream);
/* bicubic (align_corners, A = -0.75) resize of fp32 NCHW [B][3][H][W] to S x S, (x + 1) / 2, (x - mean) / std (MOD:621-633;
 * mean3 / std3: HOST pointers to 3 floats), stored as the patch matrix of the P x P stride-P convolution: row (b, py, px),
 * column (c, ky, kx), row stride ldo >= 3 P P (columns beyond 3 P P are left as the caller set them) */
int vidseg_clip_resize_patches(const float* x, int B, int H, int W, int S, int P, const float* mean3, const float* std3, float* out, int ldo,
                               vidseg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VIDSEG_HIP_H */
