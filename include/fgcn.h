/*
 * fgcn.h — C ABI of libfgcn.so: the MI355X (gfx950) kernels behind fusion-gcn's AGCN / ST-GCN block.
 *
 * The reference (mduhme/fusion-gcn) has no FFI of its own: its hot path is the op sequence of
 * torch_src/models/mmargcn/agcn.py (SpatialGraphConv :96-115, TemporalConv :49-51, SpatialTemporalConv
 * :134-136), dispatched to ATen.  Each entry point below names the reference lines whose arithmetic it
 * replaces.  The Python host (fusion_gcn_amd/ops.py) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain C types only: device pointers, sizes, a hipStream_t passed as void*.
 *   - activations are float32, channels-last: element (n, t, v, c) of a tensor with row stride `ld`
 *     lives at base[((n*T + t)*V + v)*ld + c].  `ld` and every channel offset are multiples of 4 floats
 *     and every base pointer is 16-byte aligned.
 *   - the caller owns every buffer (outputs, workspaces, partial-sum scratch); the library allocates nothing,
 *     never synchronises and only enqueues work on `stream`.
 *   - return value: 0 = ok, <0 = error (FGCN_E_*); text via fgcn_last_error() (thread-local).  Never throws
 *     or exits.  Shapes are validated on the host before any launch.
 */
#ifndef FGCN_H
#define FGCN_H

#ifdef __cplusplus
extern "C" {
#endif

#define FGCN_OK 0
#define FGCN_E_BADARG (-1)   /* null pointer / inconsistent or unsupported shape */
#define FGCN_E_ALIGN (-2)    /* pointer or stride not aligned as required */
#define FGCN_E_LAUNCH (-3)   /* hipLaunchKernel reported an error */
#define FGCN_E_ARCH (-4)     /* device is not gfx950 */

#define FGCN_MAX_V 32        /* joints per skeleton graph (reference graphs: 18..27) */
#define FGCN_MIX_MAX_ITEMS 24
#define FGCN_GRAM_MAX_ITEMS 4

int fgcn_version(void);
const char* fgcn_last_error(void);
/* 0 if the current HIP device is gfx950, FGCN_E_ARCH otherwise (message names the arch found). */
int fgcn_check_device(void);

/* Contexts: where the settings live that the launchers read on the host at call time -- the math mode, the product form and the
 * kernel-variant table below.  Every thread has a CURRENT context; a thread that never made one current reads (and, through the
 * setters below, writes) the process-wide defaults.  fgcn_ctx_create copies the calling thread's current settings into a new
 * context; fgcn_ctx_set_current(ctx) makes it current for the CALLING THREAD ONLY (NULL: back to the process-wide defaults);
 * fgcn_set_math_mode / fgcn_set_products / fgcn_set_tuning then change that context and nothing else.  Two threads with two contexts
 * -- two models in two math modes on two streams -- therefore never see each other's settings (tests/test_context_gpu.py).  A caller
 * whose work continues on another thread (an autograd backward) makes the same context current there for the duration of its calls:
 * fusion_gcn_amd.ops.context_bound does that for every autograd Function of the host.  A context must not be destroyed while it is
 * current on the destroying thread, and not while another thread still uses it. */
typedef struct fgcn_ctx fgcn_ctx;
int fgcn_ctx_create(fgcn_ctx** out);
int fgcn_ctx_destroy(fgcn_ctx* ctx);
int fgcn_ctx_set_current(fgcn_ctx* ctx);
fgcn_ctx* fgcn_ctx_get_current(void);

/* Select a kernel variant in the calling thread's current context (for tuning tools and same-box A/B runs -- value 0 of every key is the
 * measured-best default, every other form computes the same result in another summation order at most).  Keys 0..31 (others: FGCN_E_BADARG):
 *   0  row-GEMM tile when N <= 64: 0 = 128x(32*nt) rows per workgroup, 1 = 256-row tile (default), 2 = 256-row, double-buffered LDS
 *   1  row GEMM, wider N: 0 = two barriers per K chunk, 1 = double-buffered LDS
 *   4  f32 halo conv: 0 = three workgroups per CU, 1 = two
 *   5  workgroup orders (bits): 1 row GEMM XCD-aware order on; 8 row GEMM column-tile-fastest off; 4 generic weight gradient (rows_wgrad) XCD-aware off;
 *      16 spatial forward XCD-aware off; 32 split-bf16 halo conv XCD-aware off; 2 f32 halo conv XCD-aware on; 64 split weight gradient XCD-aware off
 *   6  (bits) 1: 1x1 weight gradients of the bf16 modes on the 256-thread kernel that splits fragments as it reads them;
 *      8: joint_dagg at two workgroups per CU; 16: joint_dagg's gram on the f32 MFMA in every mode;
 *      32: 1x1 weight gradients on ONE 8-wave workgroup per CU (default: two 4-wave ones); 64: all-taps kernels above 64 output
 *      columns on 8 waves (default 4); 256: at 64 columns on 4 waves (default 8); 128: all-taps weight gradient without the circular tap window;
 *      512: fgcn_spatial_wgrad on exact-f32 MFMAs in math mode bf16x3 (default there: the split-bf16 form)
 *      1024: joint_dagg with its subset count / accumulate flag as run-time values (the form before the end of round 3); 2048: both at compile time,
 *      three workgroups per CU, one tile register set (default: one set per subset refilled a chunk ahead, two workgroups per CU)
 *   7  split-bf16 kernels (bits) 1: spatial forward, one frame per wave (older form); 2: halo conv on the 32x32x16 MFMA shape;
 *      4: the same for N <= 64 only; 8: split-bf16 halo conv at <= 64 output columns as 2 x 2 waves over 128-row tiles (default: 4 x 1 waves
 *      over 192-row tiles from 1536 tiles on); 16: the 4 x 1 form with a 128-column tile for every N > 64 (default: 64 < N <= 128 only); 32: never; 64: the 4 x 1 form at <= 64 columns whatever the tile count (tests)
 *   10 output stores of the activation-writing kernels: 0 = non-temporal (streamed past L2) when the call writes 96 MiB or more, plain below;
 *      1 = always plain; 2 = always non-temporal (same results in every setting: tests/test_block_model_gpu.py)
 *   11 fgcn_spatial_bwd_tile: 2 = two four-wave workgroups per CU (8-22 % slower; parity-tested); calls with gated addends always run the
 *      eight-wave form
 *   12 joint gram of three equal-width items: 1 = the generic kernel (default: joint_gram3_kernel)
 *   13 fgcn_spatial_wgrad: workgroups to aim for (0 = 512 up to 32 samples, 1024 above)
 *   15 fgcn_spatial_bwd_tile: workgroups to aim for (0 = 256, one per CU; sets the segment count, i.e. the shape of `partial`)
 *   16 fgcn_spatial_wgrad_tile: workgroups to aim for (0 = 256; sets the slab count)
 *   17 fgcn_emb_wgrad_tile: workgroups to aim for (0 = 256; sets the slab count)
 *   18 fgcn_emb_dx_tile: 1 = 128-column tiles with a two-slot weight ring (default four)
 *   19 fgcn_emb_wgrad_tile: 1 = emb values requested one frame slot ahead (default: two)
 *   21 fgcn_spatial_wgrad_tile / fgcn_emb_wgrad_tile: 2 = 64 x 64 tiles (default: the widest tiles the channels allow)
 *   22 fgcn_emb_fwd_tile: resident workgroups to aim for (0 = 512; sets the segment count)
 *   24 fgcn_rows_gemm*: 1 = the large-problem tiles (256 / 128 rows x the widest column tile) also where they leave fewer than 256
 *      workgroups (small problems otherwise take the smallest tiles that pad no extra column) */
int fgcn_set_tuning(int key, int value);
int fgcn_get_tuning(int key);

/* Arithmetic of the convolution / GEMM kernels, a setting of the calling thread's current context (the reference's counterpart is its
 * mixed-precision step, session/procedures/step.py:55-78, which wraps model(x) in autocast).  The process-wide default is
 * FGCN_MATH_BF16X3: the arithmetic bench.py reports (float32-accurate, 1.57x the step rate of FGCN_MATH_F32 on MI355X).
 *   FGCN_MATH_F32  v_mfma_f32_32x32x2_f32 on float32 operands -- the exact-f32 parity path (<= 1e-3 rel);
 *   FGCN_MATH_BF16 (BASELINE config 5) operands rounded to bfloat16 (round-to-nearest-even) once -- as a tile is staged,
 *                  as a fragment is formed, or (weights of fgcn_tconv_halo) by fgcn_pack_split3 -- bf16 MFMAs with float32
 *                  accumulation; tensors in HBM, BatchNorm statistics, softmax, the joint mixing and every reduction
 *                  stay float32.  Different tolerance contract: logits <= 1e-2 rel, gradient cosine >= 0.98 against the
 *                  f32 path.
 *   FGCN_MATH_BF16X3 float32-accurate products on the bf16 matrix pipe: both operand fragments are split exactly into
 *                  three bfloat16 terms (x = x_h + x_m + x_l, 24 significand bits) and the six partial products down
 *                  to 2^-16 of the leading one are accumulated in float32 (the dropped ones are below 2^-23 |a.b|, the
 *                  rounding of a float32 product).  Six bf16 MFMAs replace four f32 MFMAs (2.67x the f32 matrix
 *                  rate); same tolerance contract as FGCN_MATH_F32 (the parity tests run in both).  The default. */
#define FGCN_MATH_F32 0
#define FGCN_MATH_BF16 1
#define FGCN_MATH_BF16X3 2
int fgcn_set_math_mode(int mode);
int fgcn_get_math_mode(void);
/* Product form of the convolution / 1x1 kernels inside FGCN_MATH_BF16X3 (every other kernel is unaffected):
 *   FGCN_PRODUCTS_BF16X3  six bf16 partial products from exact three-way bf16 splits (above);
 *   FGCN_PRODUCTS_F16X2   three f16 products from two-way f16 splits: x 2^s = h + l (11 + 11 significand bits + the sign of l: within
 *                         2^-24 |x| while l is a normal f16), l.l dropped (<= 2^-24 |a.b|) -- half the matrix work at float32-class
 *                         accuracy.  f16 has 5 exponent bits, so every operand BLOCK is scaled by an exact power of two that puts its
 *                         largest magnitude into [2^14, 2^15): activations per staged (row tile, channel chunk) inside the kernels
 *                         (the accumulator carries the scale, only ever towards larger magnitudes), weights per packed form
 *                         (FGCN_PACK_SPLIT2H).  Error model per operand element: max(2^-24 |x|, 2^-40 * block maximum).  Weights must
 *                         then be FGCN_PACK_SPLIT2H forms. */
#define FGCN_PRODUCTS_BF16X3 0
#define FGCN_PRODUCTS_F16X2 1
int fgcn_set_products(int products);
int fgcn_get_products(void);

/* Temporal index map shared by the row GEMMs: for output frame `to` and tap `j`
 *     num = to*ta + j*tb + tc ;  valid iff num >= 0, num % td == 0 and num/td < T_in ;  ti = num/td.
 * forward conv (kernel kt, stride s, pad p): ta=s tb=1 tc=-p td=1 ; its data gradient: ta=1 tb=-1 tc=p td=s. */
typedef struct {
    int taps, ta, tb, tc, td;
} fgcn_tmap;

/* out[(n,to,v), 0:N] (+)= bias + sum_j sum_k in[(n,ti(to,j),v), k] * w[j][k][n]      (implicit GEMM over rows)
 *   replaces nn.Conv2d 1x1 / (kt,1) forward and data-gradient: agcn.py:41-42 (tcn conv), :71-73 (conv_a/b/d),
 *   :77 (down), :125-132 (residual conv) and their autograd backward w.r.t. the input.
 *   w is packed [taps][K][N] (N contiguous, N % 4 == 0).  bias may be NULL.
 *   stat_partials (may be NULL): float[ceil(M/128)][2][N] receives per-row-tile sum and sum of squares of the
 *   values written (the BatchNorm batch statistics of agcn.py:44,78,83 come from these).  accumulate: out += .
 *   At most 2^29 rows (B * T_out * V) per call. */
int fgcn_rows_gemm(const float* in, float* out, const float* w, const float* bias, float* stat_partials,
                   int B, int T_in, int T_out, int V, int K, int N, int ld_in, int ld_out,
                   fgcn_tmap map, int accumulate, void* stream);
/* `batch` independent problems out_b[rows x N] (+)= in_b[rows x K] . w_b[K x N] in one launch (one per blockIdx.z): problem b
 * reads in + b*in_bstride, w + b*w_bstride and writes out + b*out_bstride (strides in floats, multiples of 4; row strides
 * ld_in / ld_out as above).  The per-sample V x V products of AGCNGraphConvolution on IMU graphs
 * (torch_src/models/mmargcn/graph_convolution.py:96-101 and their backward): one sample's matrix is the weight. */
int fgcn_rows_gemm_batched(const float* in, float* out, const float* w, int batch, long long in_bstride,
                           long long out_bstride, long long w_bstride, int rows, int K, int N, int ld_in, int ld_out,
                           int accumulate, void* stream);
/* The same with two batch levels, batch * inner problems per launch: problem (o, i) reads in + o*in_bstride + i*in_bstride2 (w, out
 * alike) -- the three subsets of one sample in one launch (o = sample, i = subset) where a tensor is laid out (sample, subset, ...)
 * and another (subset, sample, ...). */
int fgcn_rows_gemm_batched2(const float* in, float* out, const float* w, int batch, long long in_bstride,
                            long long out_bstride, long long w_bstride, int inner, long long in_bstride2,
                            long long out_bstride2, long long w_bstride2, int rows, int K, int N, int ld_in, int ld_out,
                            int accumulate, void* stream);
/* number of row tiles = leading dimension of stat_partials for M = B*T_out*V rows */
int fgcn_rows_gemm_tiles(long long M);

/* Temporal (kt x 1) convolution / its data gradient as a halo-tile implicit GEMM (agcn.py:41-42 and its backward):
 *   out[(n, fo(th), v), 0:N] (+)= bias + sum_j sum_k in[(n, fi(th + d_j), v), k] * W[j][k][n],   d_j = j*tb + tc,
 * over "virtual" frames th in [0, Th): fi(t) = t*in_s + in_o (valid for t in [0, Th_in)), fo(t) = t*out_s + out_o.
 * A stride-1 conv is one call with identity views; a stride-2 conv (or its data gradient) is two calls, one per frame
 * parity of the strided side, so no tap meets a structurally empty row.  The input tile plus its temporal halo is
 * staged in LDS once per 32 input channels and all taps run from it.
 *   w4 (FGCN_MATH_F32): k-interleaved packed weights float[taps][K/4][N][4]  (w4[j][k/4][n][k%4] = W[j][k][n]);
 *   w4 (FGCN_MATH_BF16X3 and FGCN_MATH_BF16): the fgcn_pack_split3 form (the bf16 mode reads its part 0 only);  K % 32 == 0.
 *   stat_partials: float[fgcn_tconv_halo_tiles(B, Th, Th_in, V)][2][N] or NULL (sums of the values written, after
 *   accumulation).  Tensors must be smaller than 2 GiB (32-bit buffer offsets). */
int fgcn_tconv_halo_tiles(int B, int Th_out, int Th_in, int V);

/* ---- INFERENCE forms of the two north-star kernels (round 6): with eval-mode BatchNorm the statistics are constants, so BatchNorm +
 * shortcut + ReLU are the producing kernel's epilogue and a block is two kernels + the attention, as SURVEY.md Appendix A.3 states it.
 * (A TRAINING step cannot do this: train-mode BatchNorm needs the statistics of the whole batch before it can be applied, so there the
 * kernels emit the partial sums and fgcn_bn_act applies them.)  Split kernels only: FGCN_MATH_BF16X3 with the bf16x3 products, or
 * FGCN_MATH_BF16.  bn_vec / res_vec: float[4][N] of fgcn_bn_eval_coeffs; res: the shortcut operand or NULL; nothing is kept for a backward.
 *
 * fgcn_tconv_halo_bn_relu -- "temporal 9x1 conv + BN + ReLU" (reference torch_src/models/mmargcn/agcn.py:49-51,134-136):
 *   out = relu(BN(conv_{taps x 1, stride 1}(in) + bias) + [res | BN_res(res)]), res laid out like out.
 * fgcn_spatial_fwd_tile_bn_relu -- aggregation + 1x1 feature contraction + BN + shortcut + ReLU (agcn.py:103-115):
 *   g = relu(BN(sum_k conv_d[k](x . A^_k) + bias_sum) + [res | BN_res(res)]), res rows of ld_res floats (x, or the down conv's output). */
int fgcn_tconv_halo_bn_relu(const float* in, float* out, const float* w4, const float* bias, const float* bn_vec,
                            const float* res, const float* res_vec, int B, int T, int V, int K, int N, int ld_in, int ld_out,
                            int taps, int tb, int tc, void* stream);
int fgcn_spatial_fwd_tile_bn_relu(const float* x, const float* a_hat, const void* w3, const float* bias_sum, float* g,
                                  const float* bn_vec, const float* res, int ld_res, const float* res_vec,
                                  int B, int T, int V, int Cin, int Cout, int ld_x, int ld_g, int a_hat_batched, void* stream);

/* ---- half-precision STORAGE of the temporal convolution's operands (math mode FGCN_MATH_BF16 only; round 6) -----------------------------
 * The reference's MixedPrecisionStep (torch_src/session/procedures/step.py:55-78, autocast) keeps conv inputs in half precision.  In
 * FGCN_MATH_BF16 the matrix kernels round their f32 inputs to bfloat16 (to nearest even) as they stage them; for the two tensors that
 * ONLY such staging reads -- G, the temporal conv's input (fgcn_bn_act), and dU, the gradient of its output (fgcn_bn_act_bwd_apply) --
 * the producer can write the bfloat16 values directly: the consumers copy instead of convert and move half the bytes, and every result
 * is BIT-IDENTICAL to the f32-storage form (one rounding per value either way).  `_h` entry points take / write `unsigned short`
 * (bfloat16 bit patterns), contiguous (rows, C); everything else is as in the entry point without the suffix. */
int fgcn_bn_act_h(const float* a, const float* vec_a, const float* b, const float* vec_b, unsigned short* out_h,
                  unsigned char* sign_mask, long long rows, int C, int res_mode, int relu, void* stream);
/* grp_rows >= 0 (0: dout is (rows, C); > 0: one row per group, as fgcn_bn_act_bwd_apply_g) */
int fgcn_bn_act_bwd_apply_h(const float* dout, int grp_rows, const float* out, const unsigned char* sign_mask,
                            const float* a, const float* vec_a, const float* b, const float* vec_b,
                            const float* sums, unsigned short* da_h, float* db,
                            long long rows, int C, int res_mode, int relu, int train, int db_accumulate, void* stream);
/* fgcn_tconv_halo with a bfloat16 input tensor (ld_in in elements; the tap form, no fused input stage) */
int fgcn_tconv_halo_h(const unsigned short* in_h, float* out, const float* w4, const float* bias, float* stat_partials,
                      int B, int Th, int V, int K, int N, int ld_in, int ld_out,
                      int T_in_full, int in_s, int in_o, int Th_in,
                      int T_out_full, int out_s, int out_o,
                      int taps, int tb, int tc, int accumulate, const float* bn_a, const unsigned char* bn_mask,
                      const float* bn_vec, void* stream);
/* fgcn_tconv_wgrad with bfloat16 tensors a and g (ld_a / ld_g in elements) */
int fgcn_tconv_wgrad_h(const unsigned short* a_h, const unsigned short* g_h, float* partial, int B, int T_g, int V, int K, int N,
                       int ld_a, int ld_g, int T_a_full, int a_s, int a_o, int Th_a,
                       int ntaps, int shift0, int tap0, int tap_step, int taps_total, int nsplit, void* stream);
/* fgcn_pw_wgrad (the 1x1 weight gradient, channel chunks) with bfloat16 tensors a and g */
int fgcn_pw_wgrad_h(const unsigned short* a_h, const unsigned short* g_h, float* partial, int B, int T_g, int V, int K, int N,
                    int ld_a, int ld_g, int T_a_full, int a_s, int a_o, int nsplit, void* stream);
/* ... and of dY, the gradient of the spatial stage's output (written by fgcn_bn_act_bwd_apply_h; read only by the staging of the two tile
 * kernels of the spatial backward): fgcn_spatial_bwd_tile / _g (extra1_group = 0: the plain form) and fgcn_spatial_wgrad_tile with dy as
 * bfloat16 (ld_dy in elements) */
int fgcn_spatial_bwd_tile_h(const unsigned short* dy_h, const float* x, const float* a_hat, const void* w3, float* dx, float* partial,
                            int B, int T, int V, int Cin, int Cout, int ld_dy, int ld_x, int ld_dx, int a_hat_batched, int accumulate,
                            const float* extra1, int extra1_group, const unsigned char* mask1, const float* extra2,
                            const unsigned char* mask2, void* stream);
int fgcn_spatial_wgrad_tile_h(const float* x, const unsigned short* dy_h, const float* a_hat, float* partial, int B, int T, int V,
                              int Cin, int Cout, int ld_x, int ld_dy, int a_hat_batched, void* stream);
/* ... and of emb, the attention embeddings (1.5 activations wide; written by fgcn_emb_fwd_tile, whose own gram reads the tile on chip; read
 * only by the operand fetches of the two tile kernels of the embedding backward): ld_e in elements */
int fgcn_emb_fwd_tile_h(const float* x, const void* w3, const float* bias, unsigned short* emb_h, float* partial, int B, int T, int V,
                        int Cin, int ic, int ld_x, int ld_e, void* stream);
int fgcn_emb_dx_tile_h(const unsigned short* emb_h, const float* d_s, const void* w3, float* dx, void* workspace, int B, int T, int V,
                       int ic, int Cx, int ld_e, int ld_dx, int d_s_batched, int accumulate, void* stream);
int fgcn_emb_wgrad_tile_h(const unsigned short* emb_h, const float* x, const float* d_s, float* partial, float* bias_partial, int B,
                          int T, int V, int ic, int Cx, int ld_e, int ld_x, int d_s_batched, void* stream);
/* ---- half-precision ACTIVATION storage (math mode FGCN_MATH_BF16 only; round 6): the typed entry points `_t` ---------------------------
 * The `_h` entry points above keep results bit-identical (only tensors that bf16 MFMA staging alone reads are bfloat16).  The reference's
 * MixedPrecisionStep (torch_src/session/procedures/step.py:55-78: torch.cuda.amp.autocast around model(x)) goes further: under autocast every
 * convolution / matmul OUTPUT is a half-precision tensor and BatchNorm / ReLU / the residual adds read and write half precision too; only
 * statistics, softmax and accumulators are float32.  The `_t` entry points give the hot path that storage format: each takes `half_mask`,
 * one bit per activation-sized tensor argument in argument order (set = the tensor is bfloat16, `unsigned short` bit patterns, strides in
 * elements); accumulation, BatchNorm statistics (summed from the float32 accumulators, before the rounding of the stored value) and every
 * small tensor stay float32.  A bfloat16 INPUT of a matrix kernel changes nothing (the FGCN_MATH_BF16 kernels round their operands to
 * bfloat16 anyway: same staged bytes); a bfloat16 OUTPUT is the float32 result rounded to nearest even once.  Masks a kernel is not built for
 * are refused with FGCN_E_BADARG.  Contract of the mode: SURVEY.md section 7 (logits <= 1e-2, loss <= 1e-3 abs ..., gradient cosine >= 0.98).
 *   fgcn_bn_act_t             bit 0 a, 1 b (shortcut), 2 out
 *   fgcn_bn_act_pool_t        bit 0 a, 1 b
 *   fgcn_bn_act_bwd_reduce_t  bit 0 dout, 1 a, 2 b          (grp_rows > 0: dout is the float32 per-group form; the ReLU gate is the sign image)
 *   fgcn_bn_act_bwd_apply_t   bit 0 dout, 1 a, 2 b, 3 da, 4 db  (a bfloat16 db: C % 8 == 0, not accumulating)
 *   fgcn_tconv_halo_t         bit 0 in, 1 out               masks 0, 1, 3; plain store epilogue (no accumulation / BatchNorm-backward sums)
 *   fgcn_spatial_fwd_tile_t   bit 0 x, 1 y                  masks 0, 2, 3
 *   fgcn_emb_fwd_tile_t       bit 0 x, 1 emb
 *   fgcn_spatial_bwd_tile_t   bit 0 dy, 1 x, 2 dx + gated addends (a per-group extra1 stays float32)     masks 0, 1, 3, 7
 *   fgcn_spatial_wgrad_tile_t bit 0 x, 1 dy                 masks 0, 2, 3
 *   fgcn_emb_dx_tile_t        bit 0 emb, 1 dx               masks 0, 1, 3
 *   fgcn_emb_wgrad_tile_t     bit 0 emb, 1 x                masks 0, 1, 3
 *   fgcn_rows_gemm_t          bit 0 in, 1 out               (a bfloat16 out: no accumulation; a single problem)
 *   fgcn_pw_gemm_t            bit 0 in, 1 out               (a bfloat16 out: no accumulation) */
int fgcn_bn_act_t(const void* a, const float* vec_a, const void* b, const float* vec_b, void* out, unsigned char* sign_mask,
                  long long rows, int C, int res_mode, int relu, int half_mask, void* stream);
int fgcn_bn_act_pool_t(const void* a, const float* vec_a, const void* b, const float* vec_b, unsigned char* sign_mask,
                       float* partial, float* pooled, int groups, int grp_rows, int C, int res_mode, int half_mask, void* stream);
int fgcn_bn_act_bwd_reduce_t(const void* dout, int grp_rows, const float* out, const unsigned char* sign_mask, const void* a,
                             const float* vec_a, const void* b, const float* vec_b, float* partials, int n_tiles, long long rows,
                             int C, int res_mode, int relu, int half_mask, void* stream);
int fgcn_bn_act_bwd_apply_t(const void* dout, int grp_rows, const float* out, const unsigned char* sign_mask, const void* a,
                            const float* vec_a, const void* b, const float* vec_b, const float* sums, void* da, void* db,
                            long long rows, int C, int res_mode, int relu, int train, int db_accumulate, int half_mask, void* stream);
int fgcn_tconv_halo_t(const void* in, void* out, const float* w4, const float* bias, float* stat_partials,
                      int B, int Th, int V, int K, int N, int ld_in, int ld_out,
                      int T_in_full, int in_s, int in_o, int Th_in,
                      int T_out_full, int out_s, int out_o,
                      int taps, int tb, int tc, int half_mask, void* stream);
int fgcn_spatial_fwd_tile_t(const void* x, const float* a_hat, const void* w3, const float* bias_sum, void* y,
                            float* stat_partials, int B, int T, int V, int Cin, int Cout, int ld_x, int ld_y,
                            int a_hat_batched, int half_mask, void* stream);
int fgcn_emb_fwd_tile_t(const void* x, const void* w3, const float* bias, void* emb, float* partial, int B, int T, int V, int Cin,
                        int ic, int ld_x, int ld_e, int half_mask, void* stream);
int fgcn_spatial_bwd_tile_t(const void* dy, const void* x, const float* a_hat, const void* w3, void* dx, float* partial,
                            int B, int T, int V, int Cin, int Cout, int ld_dy, int ld_x, int ld_dx, int a_hat_batched, int accumulate,
                            const void* extra1, int extra1_group, const unsigned char* mask1, const void* extra2,
                            const unsigned char* mask2, int half_mask, void* stream);
int fgcn_spatial_wgrad_tile_t(const void* x, const void* dy, const float* a_hat, float* partial, int B, int T, int V, int Cin,
                              int Cout, int ld_x, int ld_dy, int a_hat_batched, int half_mask, void* stream);
/* dx_old (mask 3, accumulate; or NULL): the float32 tensor that holds the values to add to -- dx = bfloat16(dx_old + term) is then only written */
int fgcn_emb_dx_tile_t(const void* emb, const float* d_s, const void* w3, void* dx, void* workspace, int B, int T, int V,
                       int ic, int Cx, int ld_e, int ld_dx, int d_s_batched, int accumulate, const float* dx_old, int half_mask,
                       void* stream);
int fgcn_emb_wgrad_tile_t(const void* emb, const void* x, const float* d_s, float* partial, float* bias_partial, int B,
                          int T, int V, int ic, int Cx, int ld_e, int ld_x, int d_s_batched, int half_mask, void* stream);
int fgcn_rows_gemm_t(const void* in, void* out, const float* w, const float* bias, float* stat_partials,
                     int B, int T_in, int T_out, int V, int K, int N, int ld_in, int ld_out,
                     fgcn_tmap map, int accumulate, int half_mask, void* stream);
int fgcn_pw_gemm_t(const void* in, void* out, const void* w3, const float* bias, float* stat_partials, long long rows,
                   int K, int N, int ld_in, int ld_out, int accumulate, int half_mask, void* stream);
/* bn_a / bn_mask / bn_vec (all NULL, or all given where fgcn_tconv_halo_bn_sums() == 1: the split-bf16 kernel of the bf16 math
 * modes): the call is the data gradient that produces dG, the gradient of G = relu(BatchNorm(a) + shortcut) (agcn.py:113-115), and
 * stat_partials receives the BatchNorm-backward sums instead of the forward moments -- per row tile and channel
 *   [0] sum dp,  [1] sum dp * (a - mean) * rstd,   dp = (value written) * [bit of bn_mask],
 * bn_a = the BatchNorm's input (same shape as out, contiguous: ld_out == N), bn_mask = fgcn_bn_act's sign image of G, bn_vec =
 * fgcn_bn_finalize's vector (mean at [0, N), rstd at [N, 2N)): what fgcn_bn_act_bwd_reduce computes in a pass of its own over dG
 * and a (two activation reads and two launches less per identity block). */
int fgcn_tconv_halo_bn_sums(void);
/* fin_vec / fin_res / fin_out / fin_mask (all NULL, or all given where fgcn_tconv_halo_bn_sums() == 1; taps > 1, a plain contiguous
 * input view with ld_in == K, V <= 32): the INPUT STAGE of the temporal conv fused into its image fill -- the north star's "temporal
 * 9x1 conv + BN + ReLU as one kernel", consumer side.  `in` is then y, the input of the graph convolution's BatchNorm, and the conv
 * runs on G = relu(y * scale + shift + fin_res) (agcn.py:113-115; fin_vec = fgcn_bn_finalize's float[4][K], fin_res = the identity
 * shortcut x, laid out like `in`), formed as the rows are staged; G (fin_out, like `in`) and its sign image (fin_mask, fgcn_bn_act's
 * layout, rows*K/8 bytes) are written once as by-products (the backward's weight gradient and ReLU gate read them): what an
 * fgcn_bn_act(res_mode = 1, relu = 1) pass in front of this call computes, without that pass. */
int fgcn_tconv_halo(const float* in, float* out, const float* w4, const float* bias, float* stat_partials,
                    int B, int Th, int V, int K, int N, int ld_in, int ld_out,
                    int T_in_full, int in_s, int in_o, int Th_in,
                    int T_out_full, int out_s, int out_o,
                    int taps, int tb, int tc, int accumulate, const float* bn_a, const unsigned char* bn_mask,
                    const float* bn_vec, const float* fin_vec, const float* fin_res, float* fin_out, unsigned char* fin_mask,
                    unsigned* in_amax, void* stream);
/* in_amax (may be NULL; FGCN_PRODUCTS_F16X2 only): a device word that receives, by integer atomic maximum (order-independent), the
 * float bits of max |in| over everything this call stages -- zero it before the first call that should count; the weight gradient of
 * the same tensor takes it as its operand scale (fgcn_tconv_wgrad). */

/* partial[s][j][k][n] = sum over the s-th slice of rows m=(n,tg,v) of a[(n,ti(tg,j),v), k] * g[m, n]
 *   (weight gradient of the same convolutions; autograd backward of agcn.py:41-42,71-73,77).
 *   partial: float[nsplit][taps][K][N]; reduce with fgcn_reduce_sum.  nsplit >= 1. */
int fgcn_rows_wgrad(const float* a, const float* g, float* partial,
                    int B, int T_a, int T_g, int V, int K, int N, int ld_a, int ld_g,
                    fgcn_tmap map, int nsplit, void* stream);

/* Weight gradient of the (taps x 1) temporal convolution, all taps of one call in a single pass over the rows
 * (backward of agcn.py:41-51 w.r.t. conv.weight):
 *     partial[slab][tap0 + i*tap_step][k][n] = sum over the slab's rows (b, t, v) of
 *                                              a[(b, (t + shift0 + i)*a_s + a_o, v), k] * g[(b, t, v), n],  i < ntaps
 *   a: float[B][T_a_full][V][ld_a] seen through the frame view f -> f*a_s + a_o (Th_a frames; frames outside the view
 *   contribute zeros), g: float[B][T_g][V][ld_g].  A stride-s convolution is s calls (one per residue of the tap offset).
 *   partial: float[fgcn_tconv_wgrad_slabs(N, nsplit)][taps_total][K][N]; every slab of the taps of this call is written;
 *   the caller sums the slabs (fgcn_reduce_sum).  ntaps in {1..5, 9}; K, N, ld_a, ld_g multiples of 4; tensors < 2 GiB. */
int fgcn_tconv_wgrad_slabs(int N, int nsplit);
/* slabs of fgcn_pw_wgrad's partial buffer, and the number of workgroups of a fgcn_tconv_wgrad launch that are resident
 * at once in the current math mode (choose nsplit so that tiles * nsplit stays within it) */
int fgcn_pw_wgrad_slabs(int N, int nsplit);
int fgcn_tconv_wgrad_resident(int N);
int fgcn_pw_wgrad_resident(int N);
int fgcn_tconv_wgrad(const float* a, const float* g, float* partial, int B, int T_g, int V, int K, int N,
                     int ld_a, int ld_g, int T_a_full, int a_s, int a_o, int Th_a,
                     int ntaps, int shift0, int tap0, int tap_step, int taps_total, int nsplit,
                     const unsigned* a_amax, const unsigned* g_amax, void* stream);

/* Weight gradient of a 1x1 convolution (theta|phi embedding, conv_d on the stacked agg, down, residual): the same
 * kernel with one accumulator per 32-channel chunk of a instead of per tap (each g fragment feeds up to 6 MFMAs):
 *     partial[slab][k][n] = sum over the slab's rows (b, t, v) of a[(b, t*a_s + a_o, v), k] * g[(b, t, v), n]
 *   partial: float[fgcn_tconv_wgrad_slabs(N, nsplit)][K][N].  Same alignment / size rules as fgcn_tconv_wgrad. */
int fgcn_pw_wgrad_chunks(int K, int N);
int fgcn_pw_wgrad(const float* a, const float* g, float* partial, int B, int T_g, int V, int K, int N,
                  int ld_a, int ld_g, int T_a_full, int a_s, int a_o, int nsplit,
                  const unsigned* a_amax, const unsigned* g_amax, void* stream);
/* a_amax / g_amax (both weight-gradient entry points; may be NULL): device words holding the float bits of max |a| / max |g| over
 * the whole tensors, as fgcn_tconv_halo / fgcn_pw_gemm leave them in `in_amax`.  With both given and FGCN_PRODUCTS_F16X2 selected the
 * split kernel runs its f16x2 form (operands scaled by the exact powers of two that put those maxima into [2^14, 2^15)); without
 * them it runs the bf16x3 form. */

/* dst[i] (+)= sum_s src[s*count + i]   (deterministic tree-free column sum; also bias / adj_b gradients) */
int fgcn_reduce_sum(float* dst, const float* src, int S, long long count, int accumulate, void* stream);

/* The same slab sum written through strides: element (tap, k, n) of the [taps][K][N] slabs lands at
 * dst[tap*st_tap + k*st_k + n*st_n] for k < K_dst (input channels beyond K_dst are padding) -- i.e. directly in a conv
 * parameter's own (out, in, taps, 1) layout, so the gradient needs no transposing copy. */
int fgcn_reduce_sum_strided(float* dst, const float* src, int S, int taps, int K, int N, int K_dst,
                            long long st_tap, long long st_k, long long st_n, int accumulate, void* stream);

/* Up to FGCN_REDUCE_MAX_ITEMS of those slab sums in one launch (the leaf reductions of a block's backward -- weight-gradient
 * slabs, adj_b, bias partials -- collected and issued together; same arithmetic and summation order as the single calls;
 * a plain fgcn_reduce_sum over `count` elements is the item taps = K = K_dst = 1, N = count, st_n = 1). */
#define FGCN_REDUCE_MAX_ITEMS 8
typedef struct {
    float* dst;
    const float* src;
    long long st_tap, st_k, st_n;
    int S, taps, K, N, K_dst, accumulate;
} fgcn_reduce_item;
int fgcn_reduce_multi(const fgcn_reduce_item* items, int n_items, void* stream);

/* FGCN_MATH_BF16X3 form of a packed (taps, K, N) weight: dst = unsigned short[3][taps][ceil(K/8)][N][8], part 0/1/2 the
 * high / middle / low bfloat16 term of the exact split w = w_h + w_m + w_l (channels beyond K are zeros; part 0 alone is the
 * round-to-nearest-even bfloat16 of w).  This is what fgcn_tconv_halo takes as `w4` in both bf16 math modes.  acc_order = 1 (fgcn_spatial_fwd's `wd` in that
 * mode; ceil(K/16)*2 groups): group 2*k16 + h holds k = 16*k16 + 4h + (j & 3) + 8*(j >> 2), the order in which a 32x32
 * MFMA accumulator enumerates its rows. */
int fgcn_pack_split3(unsigned short* dst, const float* src, int taps, int K, int N, int acc_order, void* stream);

/* All packed / split weight forms of a model in ONE launch (the step after an optimizer update re-lays-out ~250 small
 * matrices; as separate launches they sit on the critical path of a small-batch step).  An item describes one form: the
 * logical matrix W[tap][k][n] (taps x K x N) as the SUM of up to FGCN_PACK_MAX_SEG source segments -- segment s covers
 * taps [t0, t0+tlen), rows [k0, k0+klen), columns [n0, n0+nlen) and reads
 *     src[((tap-t0)*tap_step + tap0)*st_tap + (k-k0)*st_k + (n-n0)*st_n]
 * (disjoint segments = concatenation of parameters along k or n; overlapping = their sum, e.g. conv_d's three biases; an
 * uncovered range = zero padding of the channel count) -- and the layout written to dst:
 *   FGCN_PACK_PLAIN       float[taps][K][N]
 *   FGCN_PACK_K4          float[taps][ceil(K/4)][N][4]        (k-interleaved: fgcn_tconv_halo / fgcn_spatial_fwd, f32 mode)
 *   FGCN_PACK_SPLIT3      the fgcn_pack_split3 form, acc_order 0   (unsigned short[3][taps][ceil(K/8)][N][8])
 *   FGCN_PACK_SPLIT3_ACC  the fgcn_pack_split3 form, acc_order 1
 * `kgroups` must equal fgcn_pack_kgroups(mode, K).  items_dev: the item table in DEVICE memory; blockmap_dev: int[2] per
 * workgroup = (item index, index of the workgroup's first 256 units within the item), fgcn_pack_units(...) units per item
 * (one unit = one element, float4, or 8-value split group).  The reference's counterpart is nothing: its Conv2d weights are
 * consumed in place by ATen (agcn.py:41-42,71-73,77). */
#define FGCN_PACK_MAX_SEG 6
#define FGCN_PACK_PLAIN 0
#define FGCN_PACK_K4 1
#define FGCN_PACK_SPLIT3 2
#define FGCN_PACK_SPLIT3_ACC 3
/*   FGCN_PACK_SPLIT2H     FGCN_PRODUCTS_F16X2 weights: a 16-byte header whose first word holds the float bits of max |W| over the
 *                         form, then _Float16[2][taps][ceil(K/8)][N][8] = the high / low f16 parts of W * 2^s, s = 141 - biased
 *                         exponent of that maximum (scaled maximum in [2^14, 2^15); s = 0 for an all-zero form).  Needs
 *                         fgcn_pack_run_scaled (the maximum is a pass of its own). */
#define FGCN_PACK_SPLIT2H 4
/*   FGCN_PACK_SPLIT2H_ACC the same with the k order of FGCN_PACK_SPLIT3_ACC (fgcn_spatial_fwd's weights with FGCN_PRODUCTS_F16X2) */
#define FGCN_PACK_SPLIT2H_ACC 5
typedef struct {
    const float* src;
    long long st_tap, st_k, st_n;
    int t0, tlen, k0, klen, n0, nlen, tap0, tap_step;
} fgcn_pack_seg;
typedef struct {
    void* dst;
    int mode, taps, K, N, kgroups, nseg;
    fgcn_pack_seg seg[FGCN_PACK_MAX_SEG];
} fgcn_pack_item;
int fgcn_pack_kgroups(int mode, int K);
long long fgcn_pack_units(int mode, int taps, int K, int N);
int fgcn_pack_run(const fgcn_pack_item* items_dev, const int* blockmap_dev, int n_workgroups, void* stream);
/* the same for tables that hold FGCN_PACK_SPLIT2H items: three launches -- headers zeroed, per-form maxima (integer atomic max over
 * the float bits: order-independent, so reproducible), then the pack proper */
int fgcn_pack_run_scaled(const fgcn_pack_item* items_dev, const int* blockmap_dev, int n_workgroups, int n_items, void* stream);

/* dst[j][k][n] = src[n*st_n + k*st_k + jj*st_tap], jj = flip ? taps-1-j : j ; n >= N_src zero-filled up to N_dst
 * (weight re-layout into the packed [taps][K][N] form; N_dst % 4 == 0). */
int fgcn_pack_weight(float* dst, const float* src, int taps, int K, int N_src, int N_dst,
                     long long st_tap, long long st_k, long long st_n, int flip, void* stream);

/* Global average pooling in front of the classifier (agcn.py:196-197: mean over frames and joints, then persons):
 * out[g][c] = mean over the `rows` consecutive rows of group g of x[(g*rows + r)*ld + c].  Two fixed-order stages;
 * partial: float[groups * fgcn_group_mean_splits(groups, rows)][C]. */
int fgcn_group_mean_splits(int groups, int rows);
int fgcn_group_mean(const float* x, float* partial, float* out, int groups, int rows, int C, int ld, void* stream);

/* ---- joint-mixing kernels (the "graph" part: A.X over the K=3 partition adjacencies) ------------------ */
typedef struct {
    short mat;        /* which V x V matrix of the sample (0..n_mats-1) */
    short transpose;  /* 0: out_joint = row index of mat;  1: out_joint = column index */
    short in_c_lo;    /* input channel of lane 0  (lanes 0..15 read in_c_lo + lane) */
    short in_c_hi;    /* input channel of lane 16 (lanes 16..31 read in_c_hi + lane-16) */
    short mask;       /* bit0: lanes 0..15 take this term, bit1: lanes 16..31 take it */
} fgcn_mix_term;
typedef struct {
    short out_c;      /* first of the (up to 32) output channels of this tile */
    short width;      /* lanes (channels) of the tile that exist: 1..32 */
    short nterms;     /* 1..3 */
    fgcn_mix_term term[3];
} fgcn_mix_item;

/* For every sample n, frame t and item: out[(n,t,u), out_c + l] (+)= sum_terms sum_v M[u][v] * in[(n,t,v), in_c(l)]
 *   with M = mats[n][term.mat] (or its transpose).  One kernel serves
 *     agg_k = x . A^_k                (agcn.py:109-110, torch.matmul(A2, A1))        and the two backward mixes
 *     dx += sum_k dagg_k . A^_k^T ,   dtheta_k = dS_k . phi_k ,  dphi_k = dS_k^T . theta_k.
 *   mats: float[B or 1][n_mats][V][V]; mats_batched = 0 shares one set across the batch (static-adjacency ST-GCN).
 *   Channels >= in_channels / out_channels are treated as absent. */
int fgcn_joint_mix(const float* in, float* out, const float* mats, int B, int T, int V,
                   int ld_in, int ld_out, int in_channels, int out_channels, int n_mats, int mats_batched,
                   const fgcn_mix_item* items, int n_items, int accumulate, void* stream);

/* Channel-group variant (no 16-lane masks): lane j of a group owns `vw` (1 or 2) consecutive channels, one item
 * covers up to 32*vw channels with 4/8-byte buffer loads and stores; all items of a call have the same `nch`.
 * Same formula as fgcn_joint_mix; used for the agg recompute, dx and the embedding gradients of the backward pass.
 * Both tensors must be smaller than 2 GiB. */
typedef struct {
    short mat;
    short transpose;
    short in_c;       /* first input channel of the group (multiple of vw) */
} fgcn_mixv_term;
typedef struct {
    short out_c;      /* first output channel of the group (multiple of vw) */
    short nch;        /* channels in the group: vw .. 32*vw, multiple of vw */
    short nterms;     /* 1..3 */
    fgcn_mixv_term term[3];
} fgcn_mixv_item;
int fgcn_joint_mix_vec(const float* in, float* out, const float* mats, int B, int T, int V,
                       int ld_in, int ld_out, int n_mats, int mats_batched,
                       const fgcn_mixv_item* items, int n_items, int vw, int accumulate,
                       float* colsum_partial, unsigned* out_amax, void* stream);
/* colsum_partial (optional, only without accumulation): float[B * fgcn_joint_mix_chunks(B, T)][ld_out]; row i receives the
 * column sums of everything workgroup i wrote (the theta|phi bias gradient falls out of the embedding-gradient mix
 * instead of a separate pass over its output); channels no item writes receive 0.
 * out_amax (optional): a device word that receives, by integer atomic maximum, the float bits of the largest magnitude written -- the
 * operand scale of the f16x2 weight gradient that reads `out` (fgcn_pw_wgrad); zero it before the launches that should count. */
int fgcn_joint_mix_chunks(int B, int T);

typedef struct {
    short c1;     /* first channel of in1 */
    short c2;     /* first channel of in2 */
    short width;  /* channels contracted */
    short mat;    /* output matrix index */
} fgcn_gram_item;

/* partial[n][chunk][mat][u][w] = sum_{t in chunk} sum_c in1[(n,t,u), c1+c] * in2[(n,t,w), c2+c]   (32x32 padded)
 *   joint affinity  theta^T phi  (agcn.py:104-106, torch.matmul(A1, A2)) and dA^_k = x^T dagg_k in backward.
 *   partial: float[B][nchunk][n_mats][32][32], nchunk = ceil(T / t_chunk). */
int fgcn_joint_gram(const float* in1, const float* in2, float* partial, int B, int T, int V,
                    int ld1, int ld2, int t_chunk, int n_mats, const fgcn_gram_item* items, int n_items,
                    void* stream);

/* Weight gradient of conv_d with the aggregation recomputed on chip (backward of agcn.py:109-110 w.r.t. conv_d[k].weight):
 *     partial[slab][k*Cin + c][o] = sum over the slab's frames of (n, t) and joints w of
 *                                   (sum_v x[(n,t,v), c] * A^_k[n][v][w]) * dy[(n,t,w), o]
 *   replaces fgcn_joint_mix_vec (agg = x . A^, 3 activations wide, written to HBM) + the row weight-gradient GEMM over it.
 *   partial: float[B * fgcn_spatial_wgrad_chunks(B, T, Cin, Cout)][n_subsets * Cin][Cout]; the caller sums the slabs.
 *   Honors fgcn_set_math_mode for the Cin x Cout contraction (the joint mixing stays f32). */
int fgcn_spatial_wgrad_chunks(int B, int T, int Cin, int Cout);
int fgcn_spatial_wgrad(const float* x, const float* dy, const float* mats, float* partial,
                       int B, int T, int V, int Cin, int Cout, int ld_x, int ld_dy, int n_subsets,
                       int mats_batched, void* stream);

/* Both consumers of dagg = dY . Wd in one pass over it (backward of agcn.py:109-110):
 *     dx[(n,t,v), c]        (+)= sum_k sum_w A^_k[n][v][w] * dagg[(n,t,w), k*C + c]
 *     partial[n][chunk][k][v][w] = sum_{t in chunk} sum_c x[(n,t,v), c] * dagg[(n,t,w), k*C + c]      (32x32 padded)
 *   i.e. fgcn_joint_mix_vec (dx) and fgcn_joint_gram (dA^_k = x^T dagg_k) without reading the 3*C-wide dagg twice.
 *   mats: float[B or 1][n_subsets][V][V]; partial: float[B][ceil(T/t_chunk)][n_subsets][32][32].  C % 4 == 0.
 *   extra1/mask1, extra2/mask2 (NULL or both of a pair): gated addends of dx,
 *     dx[(n,t,v), c] += extra_i[(n,t,v), c] * [bit ((n,t,v)*C + c) of mask_i]
 *   -- the gradients that reach x through the block's identity shortcuts (y += x before the ReLU of the graph convolution,
 *   agcn.py:114, and + residual(x) before the block's output ReLU, agcn.py:135): extra = incoming gradient of that ReLU,
 *   mask = fgcn_bn_act's sign image of its output; contiguous float[B][T][V][C], C % 8 == 0.  Writing them here saves the
 *   BatchNorm-backward kernels a read-modify-write of dx each. */
int fgcn_joint_dagg(const float* x, const float* dagg, const float* mats, float* dx, float* partial,
                    int B, int T, int V, int C, int ld_x, int ld_dagg, int ld_dx, int n_subsets,
                    int mats_batched, int t_chunk, int accumulate, const float* extra1, const unsigned char* mask1,
                    const float* extra2, const unsigned char* mask2, void* stream);

/* S = scale * sum_chunks partial ; C[n,k,:,w] = softmax over v ; a_hat = C + adj_a[k] + adj_b[k]   (agcn.py:84,100,106-108)
 *   c_out, a_hat: float[B][K][V][V]; adj_a (constant partition adjacency), adj_b (learned, may be NULL): float[K][V][V].
 *   use_softmax = 0 gives the static-adjacency case a_hat = adj_a + adj_b (c_out untouched). */
int fgcn_adj_softmax_fwd(const float* partial, int nchunk, float scale, const float* adj_a, const float* adj_b,
                         float* c_out, float* a_hat, int B, int K, int V, int use_softmax, void* stream);
/* d_a_hat[n,k] = sum_chunks partial ; dS = scale * C .* (dC - colsum(C .* dC))  with dC = d_a_hat (softmax backward) */
int fgcn_adj_softmax_bwd(const float* partial, int nchunk, float scale, const float* c_in,
                         float* d_a_hat, float* d_s, int B, int K, int V, void* stream);

/* ---- BatchNorm / activation epilogues ------------------------------------------------------------------ */
/* mean/var from row-tile partials -> scale = gamma*rstd, shift = beta - mean*scale, and the running-stat update
 * (momentum, unbiased variance) of nn.BatchNorm2d in train mode (agcn.py:44,78,83; torch defaults eps 1e-5, 0.1).
 * out_vec: float[4][C] = {mean, rstd, scale, shift}.  running_* may be NULL. */
int fgcn_bn_finalize(const float* partials, int n_partials, long long count, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps,
                     float* out_vec, int C, void* stream);
/* eval mode: scale/shift from running statistics; out_vec as above (mean = running_mean, rstd = 1/sqrt(var+eps)) */
int fgcn_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float eps, float* out_vec, int C, void* stream);

/* out = act( a*scale_a + shift_a + r ),  r = 0 | b | b*scale_b + shift_b          (agcn.py:113-115, :135-136)
 *   res_mode: 0 none, 1 identity, 2 batch-normalised.  vec_a / vec_b: the float[4][C] of fgcn_bn_finalize.
 *   rows x C elements, all tensors share row stride ld (== C). relu: 0/1. */
/*   sign_mask (may be NULL; needs rows*C % 8 == 0): also stores bit e%8 of byte e/8 = [out[e] > 0] -- rows*C/8 bytes the
 *   backward passes can read instead of the whole of `out` (the ReLU gate of the reference's autograd). */
int fgcn_bn_act(const float* a, const float* vec_a, const float* b, const float* vec_b, float* out,
                unsigned char* sign_mask, long long rows, int C, int res_mode, int relu, void* stream);

/* The last block's epilogue and the pooling behind it in one pass (agcn.py:135-136 then :196-197):
 *     pooled[g][c] = mean over the grp_rows consecutive rows r of group g of relu( a*scale_a + shift_a + r-term )[(g*grp_rows + r), c]
 *   -- fgcn_bn_act (relu = 1) followed by fgcn_group_mean, but `out` is never written: only its sign image (sign_mask, rows*C/8 bytes,
 *   what the backward reads) and the sums.  C % 8 == 0; partial: float[groups * fgcn_bn_act_pool_splits(groups, grp_rows)][C].
 *   Fixed summation order (another one than fgcn_group_mean's). */
int fgcn_bn_act_pool_splits(int groups, int grp_rows);
int fgcn_bn_act_pool(const float* a, const float* vec_a, const float* b, const float* vec_b, unsigned char* sign_mask,
                     float* partial, float* pooled, int groups, int grp_rows, int C, int res_mode, void* stream);

/* Backward of the above, pass 1 (reductions): with dP = dout .* [out > 0] (or dout when relu = 0)
 *   partials[tile][0][c] = sum dP, [1] = sum dP * a_hat, [2] = sum dP * b_hat   (a_hat = (a-mean_a)*rstd_a).
 *   The gate comes from sign_mask when it is given (then `out` may be NULL), else from `out`. */
int fgcn_bn_act_bwd_reduce(const float* dout, const float* out, const unsigned char* sign_mask, const float* a,
                           const float* vec_a, const float* b, const float* vec_b, float* partials, int n_tiles,
                           long long rows, int C, int res_mode, int relu, void* stream);
/* pass 2: da = scale_a * (dP - s1/m - a_hat*s2a/m) (train) or scale_a*dP (eval);
 *         db = dP (identity) or scale_b*(dP - s1/m - b_hat*s2b/m);  sums: float[3][C] reduced partials.
 *   db may be NULL (res_mode 0); db_accumulate adds into db instead of storing. */
int fgcn_bn_act_bwd_apply(const float* dout, const float* out, const unsigned char* sign_mask, const float* a,
                          const float* vec_a, const float* b, const float* vec_b, const float* sums, float* da, float* db,
                          long long rows, int C, int res_mode, int relu, int train, int db_accumulate,
                          void* stream);
/* Both passes with the gradient of a POOLED output (fgcn_bn_act_pool): dout_g is float[rows / grp_rows][C], one row per group of grp_rows consecutive
 * rows (the pooled gradient already divided by the group size); every row reads its group's row instead of a rows x C broadcast of it. */
int fgcn_bn_act_bwd_reduce_g(const float* dout_g, int grp_rows, const float* out, const unsigned char* sign_mask, const float* a,
                             const float* vec_a, const float* b, const float* vec_b, float* partials, int n_tiles,
                             long long rows, int C, int res_mode, int relu, void* stream);
int fgcn_bn_act_bwd_apply_g(const float* dout_g, int grp_rows, const float* out, const unsigned char* sign_mask, const float* a,
                            const float* vec_a, const float* b, const float* vec_b, const float* sums, float* da, float* db,
                            long long rows, int C, int res_mode, int relu, int train, int db_accumulate, void* stream);
/* number of row tiles the reduce kernel uses for `rows` rows (leading dim of its partials) */
int fgcn_elem_tiles(long long rows);
/* The same three passes for a plain BatchNorm (no residual, no activation) whose result is a CHANNEL WINDOW of a wider tensor: one of the
 * six branches of MS-G3D's multi-scale temporal convolution, concatenated on the channel axis (reference torch_src/models/msg3d/ms_tcn.py:88-109).
 *   fgcn_bn_apply_ld:      out[r * ld_out + c] = a[r * C + c] * scale[c] + shift[c]     (out = window base; ld_out >= C, both multiples of 4)
 *   fgcn_bn_bwd_reduce_ld: partials as fgcn_bn_act_bwd_reduce with res_mode = relu = 0, dout read as dout[r * ld_dout + c] (dout = window base)
 *   fgcn_bn_bwd_apply_ld:  da (contiguous, float[rows][C]) as fgcn_bn_act_bwd_apply with res_mode = relu = 0, dout read the same way
 * -- no torch.cat of the branches and no contiguous copies of its backward slices. */
int fgcn_bn_apply_ld(const float* a, const float* vec_a, float* out, long long rows, int C, int ld_out, void* stream);
int fgcn_bn_bwd_reduce_ld(const float* dout, int ld_dout, const float* a, const float* vec_a, float* partials, int n_tiles,
                          long long rows, int C, void* stream);
int fgcn_bn_bwd_apply_ld(const float* dout, int ld_dout, const float* a, const float* vec_a, const float* sums, float* da,
                         long long rows, int C, int train, void* stream);

/* partials[tile][c] = sum over the tile's rows of x[row][c]   (bias gradients); n_tiles = fgcn_elem_tiles(rows) */
int fgcn_col_sum(const float* x, float* partials, long long rows, int C, int ld, void* stream);

/* ---- fused spatial block forward (north-star kernel 1) -------------------------------------------------- */
/* y[(n,t,w), o] = sum_k bd_k[o] + sum_c Wd_k[o][c] * sum_v x[(n,t,v), c] * a_hat[n][k][v][w]
 *   = conv_d[k](x . A^_k) summed over the K subsets (agcn.py:103-111) in ONE kernel: A^ staged in LDS, the joint
 *   aggregation and the Cin x Cout contraction chained on MFMA without materialising agg.
 *   wd: k-interleaved packed float[K*Cin/4][Cout][4] with wd[(k*Cin+c)/4][o][(k*Cin+c)%4] = Wd_k[o][c] (Cin % 4 == 0);
 *   bias_sum: float[Cout] (= sum_k bd_k) or NULL.
 *   stat_partials: float[fgcn_spatial_tiles(B,T)][2][Cout] or NULL. */
int fgcn_spatial_fwd(const float* x, const float* a_hat, const float* wd, const float* bias_sum, float* y,
                     float* stat_partials, int B, int T, int V, int Cin, int Cout, int ld_x, int ld_y,
                     int n_subsets, int a_hat_batched, void* stream);
int fgcn_spatial_tiles(int B, int T);
/* The same product in its TILE form (split-bf16 math mode, bf16x3 products; fgcn_spatial_tile.hip): a workgroup owns 128 / V whole
 *   frames of one sample (125 of 128 matrix rows at V = 25 instead of 25 of 32 columns per frame) times 64 / 128 output columns; the
 *   aggregation x . A^_k of a (32-channel tile, subset) pair is formed once per workgroup on the matrix pipe and written to an LDS
 *   image, from which the feature contraction runs like one tap of fgcn_tconv_halo.
 *   w3: fgcn_pack_split3 form (acc_order 0) of the (3 Cin) x Cout matrix [k * Cin + c][o] = Wd_k[o][c] (one tap, K = 3 Cin); three
 *   subsets; Cin % 64 == 0; 16 <= V <= 32.  stat_partials: float[fgcn_spatial_fwd_tile_tiles(B, T, V)][2][Cout] or NULL.
 *   fgcn_spatial_fwd_tile_available: 1 when the current math mode / products and these sizes run on this kernel. */
int fgcn_spatial_fwd_tile(const float* x, const float* a_hat, const void* w3, const float* bias_sum, float* y,
                          float* stat_partials, int B, int T, int V, int Cin, int Cout, int ld_x, int ld_y,
                          int a_hat_batched, void* stream);
int fgcn_spatial_fwd_tile_tiles(int B, int T, int V);
int fgcn_spatial_fwd_tile_available(int V, int Cin, int Cout);

/* The backward of the same stage in tile form (fgcn_spatial_bwd_tile.hip; reference: the autograd backward of agcn.py:103-111,
 * SURVEY.md Appendix A.2) -- ONE launch for what fgcn_pw_gemm (dagg = dy . Wd) + fgcn_joint_dagg did through a three-activation-wide
 * tensor in HBM:
 *     dagg_k = dy . Wd_k (on chip only);  dx (+)= sum_k dagg_k . A^_k^T;  partial = per-workgroup sums of dA^_k = x^T . dagg_k.
 *   dy (B,T,V,>=Cout), x (B,T,V,>=Cin), a_hat (B or 1, 3, V, V), dx (B,T,V,>=Cin);
 *   w3: fgcn_pack_split3 form (acc_order 0) of the Cout x (3 Cin) matrix [o][k * Cin + c] = Wd_k[o][c] (one tap, K = Cout);
 *   partial: float[B][fgcn_spatial_bwd_tile_segments(B, T, V)][3][32][32] (rows v, columns w; entries beyond V are zero) -- the layout
 *   fgcn_adj_softmax_bwd sums.  Three subsets; Cin % 64 == 0, Cout % 64 == 0; 16 <= V <= 32; math mode bf16x3 with either product form (the
 *   kernel always multiplies three-way bf16 splits: fgcn_spatial_bwd_tile_available).  accumulate != 0: dx += (load, add, store); every sum has a fixed order.
 *   extra1 / mask1, extra2 / mask2 (all four or none; not with accumulate; ld_x == Cin): dx = ... + extra_i * [bit of mask_i] -- contiguous
 *   (B, T, V, Cin) tensors with fgcn_bn_act's one-bit sign images: the ReLU-gated gradients of the block's two identity shortcuts
 *   (agcn.py:114,135), as in fgcn_joint_dagg. */
int fgcn_spatial_bwd_tile(const float* dy, const float* x, const float* a_hat, const void* w3, float* dx, float* partial, int B,
                          int T, int V, int Cin, int Cout, int ld_dy, int ld_x, int ld_dx, int a_hat_batched, int accumulate,
                          const float* extra1, const unsigned char* mask1, const float* extra2, const unsigned char* mask2,
                          void* stream);
/* ... with the first gated addend given per GROUP of extra1_group consecutive samples: extra1 = float[B / extra1_group][Cin], every row of a group's
 * samples adds its group's row (gated by mask1's bits as before) -- the gradient of the pooled output of the model's last block (fgcn_bn_act_pool)
 * without its (B, T, V, Cin) broadcast.  Not accumulating. */
int fgcn_spatial_bwd_tile_g(const float* dy, const float* x, const float* a_hat, const void* w3, float* dx, float* partial, int B,
                            int T, int V, int Cin, int Cout, int ld_dy, int ld_x, int ld_dx, int a_hat_batched,
                            const float* extra1, int extra1_group, const unsigned char* mask1, const float* extra2,
                            const unsigned char* mask2, void* stream);
int fgcn_spatial_bwd_tile_segments(int B, int T, int V);
int fgcn_spatial_bwd_tile_available(int V, int Cin, int Cout);

/* conv_d's weight gradient of the same stage in tile form (fgcn_spatial_wgrad_tile.hip; reference: the autograd backward of
 * agcn.py:103-111 with respect to conv_d[k].weight): fgcn_spatial_wgrad's sum,
 *     partial[slab][k*Cin + c][o] = sum over the slab's (n, t) and joints w of (sum_v x[(n,t,v), c] * A^_k[n][v][w]) * dy[(n,t,w), o],
 *   for every channel count in 64s: a workgroup owns a (64 or 128) x (64 or 128) tile of all three subsets and walks whole frame
 *   tiles, so x and dy are read Cout/128 and Cin/128 times instead of Cout/64 and Cin/32 times, and the 3-wide aggregation that
 *   wider layers wrote with fgcn_joint_mix_vec and read back in the row weight-gradient GEMM never exists.
 *   a_hat: float[B or 1][3][V][V] (a_hat_batched: one per sample).  partial: float[fgcn_spatial_wgrad_tile_slabs(B, T, V, Cin, Cout)]
 *   [3 * Cin][Cout]; the caller sums the slabs (fgcn_reduce_multi).  Math mode FGCN_MATH_BF16X3 only (either product form; the kernel
 *   always multiplies three-way bf16 splits, the mixing included): fgcn_spatial_wgrad_tile_available.  Every sum has a fixed order.
 *   Tuning key 16: workgroups to aim for (0 = 256, one per CU; sets the slab count). */
int fgcn_spatial_wgrad_tile(const float* x, const float* dy, const float* a_hat, float* partial, int B, int T, int V, int Cin,
                            int Cout, int ld_x, int ld_dy, int a_hat_batched, void* stream);
int fgcn_spatial_wgrad_tile_slabs(int B, int T, int V, int Cin, int Cout);
int fgcn_spatial_wgrad_tile_available(int V, int Cin, int Cout);

/* Forward of the attention embeddings with the affinity gram on chip (fgcn_emb_fwd_tile.hip; reference: A1 = conv_a[k](x), A2 = conv_b[k](x),
 * torch.matmul(A1, A2) of SpatialGraphConv.forward, torch_src/models/mmargcn/agcn.py:104-106):
 *     emb[(n,t,v), j] = sum_c x[(n,t,v), c] Wemb[c][j] + bias[j]     (B, T, V, ld_e) rows [th0 ph0 th1 ph1 th2 ph2], each `ic` wide: written
 *     partial[n][s][k][v][w] = sum over the frames t of row segment s and the channels e of emb[(n,t,v), th_k + e] emb[(n,t,w), ph_k + e]
 * w3 = fgcn_pack_split3 of the (1, Cin, 6 ic) matrix, bias: float[6 ic].  partial: float[B][fgcn_emb_fwd_tile_segments(B, T, V, ic)][3][32][32]
 * (rows / columns >= V are zeros), the input format of fgcn_adj_softmax_fwd (nchunk = segments), which applies the 1 / (ic T) scale.
 * Replaces fgcn_pw_gemm / fgcn_rows_gemm (emb) + fgcn_joint_gram and the gram's read of the 1.5-activation-wide emb.  Sizes: 16 <= V <=
 * FGCN_MAX_V, ic 16 / 32 / 64, Cin a multiple of 32; math modes FGCN_MATH_BF16X3 (either product form: exact three-way bf16 splits) and
 * FGCN_MATH_BF16: fgcn_emb_fwd_tile_available.  Tuning key 22: resident workgroups to aim for (0 = 512; sets the segment count).
 * emb == NULL (inference: only the backward reads the embeddings): emb is not written, `partial` is all the call produces. */
int fgcn_emb_fwd_tile(const float* x, const void* w3, const float* bias, float* emb, float* partial, int B, int T, int V, int Cin, int ic,
                      int ld_x, int ld_e, void* stream);
int fgcn_emb_fwd_tile_segments(int B, int T, int V, int ic);
int fgcn_emb_fwd_tile_available(int V, int ic, int Cin);

/* Backward of the attention embeddings with the embedding gradient on chip (fgcn_emb_tile.hip; reference: the autograd backward of
 * SpatialGraphConv.forward through A1 = conv_a[k](x), A2 = conv_b[k](x), softmax(A1^T A2 / (ic T)), torch_src/models/mmargcn/agcn.py:104-106).
 *   emb: (B, T, V, ld_e) rows [th0 ph0 th1 ph1 th2 ph2], each `ic` channels wide (the 6 ic outputs of the stacked 1x1 embedding);
 *   d_s: float[B or 1][3][V][V], the gradient in front of the column softmax (fgcn_adj_softmax_bwd's dS, scale folded in);
 *   demb[(n,t,v), th_k + e] = sum_w dS_k[v][w] emb[(n,t,w), ph_k + e],  demb[(n,t,w), ph_k + e] = sum_v dS_k[v][w] emb[(n,t,v), th_k + e]
 * is formed per frame on the matrix pipe inside both kernels and never written:
 *   fgcn_emb_dx_tile     dx[(n,t,v), 0:Cx] (+)= demb[(n,t,v), :] . Wemb^T, w3 = fgcn_pack_split3 of the (1, 6 ic, Cx) matrix [j][c] = Wemb[j][c]
 *                        (accumulate: dx += ...);
 *   fgcn_emb_wgrad_tile  partial[s][j][c] = sum over the rows of slab s of demb[row, j] x[row, c]  (float[slabs][6 ic][Cx]: already the
 *                        parameters' (out, in) order) and bias_partial[s][j] = sum of demb[row, j] (float[slabs][6 ic]); the caller adds the
 *                        fgcn_emb_wgrad_tile_slabs(B, T, V, ic, Cx) slabs (fgcn_reduce_multi).
 * Sizes: 16 <= V <= FGCN_MAX_V, ic a multiple of 16, Cx a multiple of 64; math modes FGCN_MATH_BF16X3 (either product form: the kernels
 * always multiply exact three-way bf16 splits, the mixing included) and FGCN_MATH_BF16 (operands rounded to bfloat16 once):
 * fgcn_emb_tile_available.  Replaces fgcn_joint_mix_vec(demb) + fgcn_pw_gemm / fgcn_rows_gemm(demb . W) + fgcn_pw_wgrad(x, demb) and the
 * 1.5-activation-wide demb tensor between them.  Every sum has a fixed order.
 * Tuning key 17: fgcn_emb_wgrad_tile workgroups to aim for (0 = 256; sets the slab count). */
int fgcn_emb_dx_tile(const float* emb, const float* d_s, const void* w3, float* dx, void* workspace, int B, int T, int V, int ic, int Cx, int ld_e,
                     int ld_dx, int d_s_batched, int accumulate, void* stream);
/* bytes of fgcn_emb_dx_tile's caller-owned workspace (16-byte aligned; the split bf16 planes of dS and dS^T of every sample, written by a
 * small first launch and read by every workgroup of the main one) in the current math mode */
long long fgcn_emb_dx_tile_workspace(int B, int d_s_batched);
int fgcn_emb_wgrad_tile(const float* emb, const float* x, const float* d_s, float* partial, float* bias_partial, int B, int T, int V, int ic,
                        int Cx, int ld_e, int ld_x, int d_s_batched, void* stream);
int fgcn_emb_wgrad_tile_slabs(int B, int T, int V, int ic, int Cx);
int fgcn_emb_tile_available(int V, int ic, int Cx);

/* ---- 1-D graph convolutions on IMU graphs --------------------------------------------------- */
/* Batched transpose between the node-major (B, V, F) and feature-major (B, F, V) images of an activation:
 *     out[b][c][r] = in[b][r][c]  (r < R, c < C);  out rows have stride ld_out >= R, their columns [R, ld_out) are zero-filled.
 * STGCNGraphConvolution (torch_src/models/mmargcn/graph_convolution.py:12-52) is then three existing entry points: the 1x1
 * Conv1d = fgcn_rows_gemm over node-major rows, `torch.matmul(support, adj.t())` = fgcn_rows_gemm / fgcn_tconv_halo (taps = 1)
 * over feature-major rows with the V x V adjacency as the shared weight, residual + ReLU = fgcn_bn_act. */
int fgcn_transpose(const float* in, float* out, int B, int R, int C, int ld_in, int ld_out, void* stream);

/* Softmax of AGCNGraphConvolution's attention (graph_convolution.py:95-100: softmax over dim -2 of theta^T phi / ic, then
 * + adj[i]) on the TRANSPOSED scores st[row = (b, k, w)][v], so that it runs along the contiguous axis (V up to thousands):
 *     c = softmax_v(scale * st[row][0:V]);   a = c + adj_t[row % KV][0:V]   (adj_t: float[K*V][ld] = (adj_a + adj_b)^T per subset)
 *     backward: ds = scale * c .* (da - sum_v c .* da).
 * Rows have stride ld >= V; the columns [V, ld) of every output row are zero-filled. */
int fgcn_row_softmax_fwd(const float* st, const float* adj_t, float* c_out, float* a_out, long long rows, int V, int ld,
                         int KV, float scale, void* stream);
int fgcn_row_softmax_bwd(const float* da, const float* c, float* ds, long long rows, int V, int ld, float scale, void* stream);

/* ---- the step after the path: parameter update over flat buffers ------------------- */
/* One launch applies torch.optim's update to every trainable value of the model (reference: create_optimizer,
 * torch_src/session_helper.py:80-84, optimizer.step() in session/session.py:176-183):
 *   FGCN_OPT_SGD    d = g + wd*p;  buf = first step ? d : momentum*buf + (1-dampening)*d;  d = nesterov ? d + momentum*buf : buf;
 *                   p -= lr*d                                                        (state1 = buf, NULL when momentum == 0)
 *   FGCN_OPT_ADAM   g += wd*p;  m += (g-m)(1-beta1);  v = beta2*v + (1-beta2) g*g;
 *                   p -= lr/(1-beta1^step) * m / (sqrt(v)/sqrt(1-beta2^step) + eps)   (state1 = exp_avg, state2 = exp_avg_sq)
 *   FGCN_OPT_ADAMW  p *= 1 - lr*wd first, g untouched, then the same moments and update.
 * g is read as grad_scale * grads (the 1/world of the data-parallel average).  params, grads, state*: float[n], 16-byte
 * aligned, n % 4 == 0 (padding elements must hold zeros in all buffers); step counts from 1.  amsgrad / maximize: not built. */
#define FGCN_OPT_SGD 0
#define FGCN_OPT_ADAM 1
#define FGCN_OPT_ADAMW 2
int fgcn_optim_step(float* params, const float* grads, float* state1, float* state2, long long n, int kind,
                    float lr, float weight_decay, float grad_scale, float beta1, float beta2, float eps,
                    float momentum, float dampening, int nesterov, long long step, void* stream);

/* ---- MS-G3D data movement ------------------------------------------------------------------
 * (3 x 1) temporal max pooling with padding 1 and stride `stride` (nn.MaxPool2d((3,1), (stride,1), (1,0)) of
 * MultiScale_TemporalConv's pooling branch, models/msg3d/ms_tcn.py:72-78):
 *   out[(b, to, v), c] = max_j in[(b, to*stride + j - 1, v), c], j = 0..2 inside [0, T_in); idx = the winning tap (first maximum,
 *   as torch keeps it).  `in` may be a channel window of a wider tensor (row stride ld_in); out / idx / dout are contiguous
 *   (B, T_out, V, C), T_out = (T_in - 1) / stride + 1.  The backward is a gather: din[(b, ti, v), c] (+)= the dout of every
 *   window that holds ti and whose winning tap is ti. */
int fgcn_tmaxpool3_fwd(const float* in, float* out, unsigned char* idx, int B, int T_in, int T_out, int V, int C, int ld_in,
                       int stride, void* stream);
int fgcn_tmaxpool3_bwd(const float* dout, const unsigned char* idx, float* din, int B, int T_in, int T_out, int V, int C,
                       int ld_in, int stride, int accumulate, void* stream);

/* Temporal-window unfold (UnfoldTemporalWindows.forward, models/msg3d/ms_gtcn.py:37-45): the `window` frames around every
 * stride-th frame become `window * V` nodes of one spatial-temporal graph,
 *   out[(b, to, j*V + v), c] = x[(b, to*stride + j*dilation - pad, v), c]   (zeros outside [0, T)),
 *   pad = (window + (window-1)*(dilation-1) - 1) / 2,  T_out = (T + 2 pad - dilation (window-1) - 1) / stride + 1.
 * backward != 0: `in` is d(out) (B, T_out, window*V, C) and `out` receives dx (B, T, V, C) (a gather over the windows). */
int fgcn_unfold_windows(const float* in, float* out, int B, int T, int T_out, int V, int C, int window, int stride,
                        int dilation, int backward, void* stream);

/* ---- 1x1 convolutions in the split-bf16 math modes: persistent row GEMM (fgcn_pw.hip) ---------------------------------------------
 * out[m][n] (+)= sum_k in[m][k] * W[k][n] + bias[n] over `rows` contiguous rows (row strides ld_in / ld_out): the theta|phi embedding,
 * dY.Wd, down and dEmb.W products of the block (agcn.py:71-73,77,104-111) where the row GEMM above has no temporal map.  w3 = the
 * fgcn_pack_split3 form of the (1, K, N) matrix; K % 32 == 0, N % 4 == 0; stat_partials: float[fgcn_pw_gemm_tiles(rows)][2][N] or NULL
 * (sum and sum of squares of the values written, per 128-row tile).  A workgroup walks a list of (128 rows x 64 | 128 columns) tiles
 * and requests the next chunk's -- or the next tile's -- rows before the current MFMAs: the tiles of a 1x1 convolution are too short
 * (K = 64..384) to hide their own staging latency and store tail.  fgcn_pw_gemm_available() = 1 in FGCN_MATH_BF16X3 / FGCN_MATH_BF16. */
int fgcn_pw_gemm_available(void);
int fgcn_pw_gemm_tiles(long long rows);
int fgcn_pw_gemm(const float* in, float* out, const void* w3, const float* bias, float* stat_partials, long long rows,
                 int K, int N, int ld_in, int ld_out, int accumulate, unsigned* in_amax, void* stream);
/* (w3: the FGCN_PACK_SPLIT2H form with FGCN_PRODUCTS_F16X2 selected; in_amax as for fgcn_tconv_halo) */

/* ---- the two ends of the step: input BatchNorm and loss (fgcn_head.hip) ------------------------------------------------------
 * `data_bn` = nn.BatchNorm1d(M*V*C) over the network input x (N, M, T, V, C) viewed as (N, M*V*C, T)
 * (torch_src/models/mmargcn/agcn.py:150,186-188; msg3d.py:93,152-154): channel ch = (m*V + v)*C + c, statistics over (n, t).
 *   stats:      partials float[fgcn_data_bn_tiles(N, T)][2][M*V*C] (sum x, sum x^2 per tile) -> fgcn_bn_finalize(count = N*T)
 *               gives vec float[4][M*V*C] and updates the running statistics;
 *   apply:      out (N*M, T, V, Cp) = x * scale[ch] + shift[ch], channels [C, Cp) zero -- the blocks' input layout, written directly;
 *   bwd_reduce: partials float[tiles][2][M*V*C] = (sum dout, sum dout * xhat) -> fgcn_reduce_sum -> (d beta, d gamma);
 *   bwd_apply:  dx (N, M, T, V, C) = scale * (dout - sum0/m - xhat * sum1/m), m = N*T (train) | scale * dout (eval). */
int fgcn_data_bn_tiles(int N, int T);
int fgcn_data_bn_stats(const float* x, float* partials, int N, int M, int T, int V, int C, void* stream);
int fgcn_data_bn_apply(const float* x, const float* vec, float* out, int N, int M, int T, int V, int C, int Cp, void* stream);
int fgcn_data_bn_bwd_reduce(const float* dout, const float* x, const float* vec, float* partials, int N, int M, int T, int V,
                            int C, int Cp, void* stream);
int fgcn_data_bn_bwd_apply(const float* dout, const float* x, const float* vec, const float* sums, float* dx, int N, int M, int T,
                           int V, int C, int Cp, int train, void* stream);
/* nn.CrossEntropyLoss() (mean reduction; session/session.py:53, procedures/step.py:38-46) over logits (rows, classes) with row
 * stride ld and int64 labels: probs float[rows][classes] = softmax (kept for the backward), row_loss float[rows], loss float[2] =
 * {mean over the rows whose label is not -100 (torch's ignore_index rows do not count), that row count}; any OTHER label outside
 * [0, classes) -- a device assert in torch -- makes the loss and, in the backward, its row's gradient NaN; one workgroup, fixed
 * summation order.  bwd: dlogits (rows, ld_out) = (probs - onehot) * dloss[0] / loss[1], columns [classes, ld_out) zero. */
int fgcn_cross_entropy_fwd(const float* logits, const long long* labels, float* probs, float* row_loss, float* loss, int rows,
                           int classes, int ld, void* stream);
int fgcn_cross_entropy_bwd(const float* probs, const long long* labels, const float* loss, const float* dloss, float* dlogits,
                           int rows, int classes, int ld_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FGCN_H */
