/*
 * vivid_hip.h — C ABI of libvivid_hip.so: MI355X (gfx950) kernels for the VIVID
 * pose-conditioned UNet denoiser.
 *
 * The reference (danielcodelavin/vivid) has no native layer: its device boundary
 * is the set of stock PyTorch op call sites inside training/models.py
 * (SURVEY.md 2.3).  Each entry point below replaces one group of those call
 * sites; the reference lines are cited per function.  All tensors are raw device
 * pointers to fp32 data; activations are NHWC ([rows, H, W, C], C fastest) and
 * every pointer passed as an NHWC source must be 16-byte aligned with C % 4 == 0.
 * Nothing here allocates device memory or synchronises; work is enqueued on the
 * stream given to vh_ctx_create / vh_ctx_set_stream.
 *
 * Every call returns VH_OK (0) or a negative VH_E* code; vh_last_error() gives the
 * message.  Arguments are validated on the host before any launch.
 *
 * Record / replay: between vh_plan_begin() and vh_plan_end() the op entry points
 * append to a plan instead of launching; vh_plan_run() replays the whole recorded
 * denoiser evaluation with no per-op host work (shapes and buffers of one
 * denoiser call are static across the sampler's 2N-1 calls).
 */
#ifndef VIVID_HIP_H
#define VIVID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VH_OK 0
#define VH_EINVAL -1   /* bad argument / shape */
#define VH_EHIP -2     /* HIP runtime error */
#define VH_ESTATE -3   /* call not valid in the current record/replay state */

typedef struct vh_ctx vh_ctx;
typedef struct vh_plan vh_plan;

#define VH_ABI_VERSION 5
int vh_abi_version(void);                                /* == VH_ABI_VERSION of the header the library was built from */
const char* vh_last_error(void);

/* stream: a hipStream_t (0 = default stream). */
int vh_ctx_create(void* stream, vh_ctx** out);
int vh_ctx_destroy(vh_ctx* ctx);
int vh_ctx_set_stream(vh_ctx* ctx, void* stream);

/* Process-wide diagnostic knobs, read when an op is issued or recorded; none changes results.  "attn_xcd": 1 (default) places
 * every (batch, head) of the bf16x3 attention on one XCD, 0 keeps the plain workgroup order (A/B timing in one process);
 * "attn_m16": 1 (default) runs 64-channel bounded-logit attention on the 16x16x32 MFMA kernel, 0 on the 32x32x16 one;
 * "dbg_lo" / "dbg_hi": the two halves of a device pointer that receives the clock stamps of the diagnostic builds
 * (-DVH_CLOCK, tools/clock_probe.py) - the product build never writes to it;
 * "conv_korder": -1 (default) K order of the 3x3 bf16x3 convolutions by input size / vh_conv_args.korder, 0 tap-major, 1 chunk-major;
 * "conv_stagger": -1 (default) vh_conv_args.stagger decides, 0 never, 1 always;
 * "attn_pipe": 1 (default) software-pipelined bf16x3 attention kernels for long sequences, 0 the plain ones;
 * "attn_nomax": 1 (default) bounded-logit attention keeps no running maximum, 0 keeps it;
 * "conv_slim2": -1 (default) Cout <= 64 layers at large M take the 256x64 tile (two workgroups per CU), 0 the 512x64 one;
 * "conv_korder_mb": input size in MB above which 3x3 convolutions take the chunk-major K order (default: see conv_x3.hip);
 * "conv_ksplit": > 0 forces that many K slices where split-K is possible (default 0: the dispatcher's rule);
 * "conv_patch": -1 (default) eligible 3x3 Cout == 64 layers take the patch-resident kernel by the size rule, 0 never, 1 whenever eligible;
 * "fuse_concat": whole-network walks (vh_net_*, and vivid_amd.engine through the same knob): the halves of a decoder block's concat input written by
 *   the convolutions that produce them (vh_s8_sink) - 2 both halves, 1 the x half only, 0 (default) never (a vh_split pass over the fp32 tensors);
 * "conv_patch96": 1 (default) Cout = 192 layers run the patch-resident kernel as two 96-channel blocks, 0: the 256x192 tile (A/B);
 * "conv_patch_tail": tail segment of the patch-resident kernel - 2 (default) staged per wave through registers, 1 through two LDS-DMA stages (A/B);
 * "conv_tail_f32": vh_conv_args.tail_f32 launches (and the whole-network walks, which ask vh_conv_takes_patch per decoder block) - 1 (default) wherever the
 *   patch-resident kernel takes a tail segment plus Cout = 256 from 64x64 up, 0 never (the walks fall back to vh_split's raw S8 form), 2 also Cout = 512 (A/B);
 * "conv_src_f32": vh_conv_args.src_f32 launches (and the walks' conv_res0 of decoder blocks, which ask vh_conv_takes_patch) - 2 (default) every block width
 *   under the patch-resident kernel's size rule, 1 only Cout = 64 and 96-channel blocks (Cout = 192), 0 never (a vh_split pass and the S8 convolution instead);
 * "conv_patch_delay": start delay, in units of 2048 shader cycles, of every CU's second workgroup in the first round of a patch-kernel launch.
 * The library reads no environment variables.  Returns VH_EINVAL for an unknown name. */
int vh_set_knob(const char* name, int value);

/* 0 for a product build.  Non-zero: some translation unit was compiled with -DVH_DIAG (clock stamps or timing ablations that
 * compute WRONG results on purpose; `make variant`) - the Python binding refuses to load such a library as libvivid_hip.so. */
int vh_diag_flags(void);

/* Per-kernel timing with HIP events on the launch stream (bench.py's roofline line).
 * While enabled, every launch (direct or replayed) is bracketed by two events and carries its
 * algorithmic FLOPs / HBM bytes; vh_profile_read synchronises the stream, sums them per
 * kernel family (VH_TAG_*) into the caller's arrays of length ntags, and clears the records. */
enum { VH_TAG_CONV3 = 0, VH_TAG_CONV1 = 1, VH_TAG_ATTN = 2, VH_TAG_PIXNORM = 3, VH_TAG_QKVSPLIT = 4,
       VH_TAG_EMBED = 5, VH_TAG_ASSEMBLE = 6, VH_TAG_SAMPLER = 7, VH_TAG_PREP = 8, VH_TAG_WARP = 9, VH_TAG_SPLIT = 10,
       VH_NUM_TAGS = 11 };
int vh_profile_enable(vh_ctx* ctx, int on);
int vh_profile_read(vh_ctx* ctx, int ntags, double* ms, double* flops, double* bytes, long long* launches);
/* Same records, one entry per launch in launch order (up to max_n); clears them. */
int vh_profile_read_list(vh_ctx* ctx, int max_n, int* tags, double* ms, double* flops, double* bytes, int* n_out);

int vh_plan_begin(vh_ctx* ctx);
int vh_plan_end(vh_ctx* ctx, vh_plan** out);
int vh_plan_abort(vh_ctx* ctx);                          /* drop the plan being recorded (an op failed validation); no-op when not recording */
int vh_plan_capture_graph(vh_ctx* ctx, vh_plan* plan);   /* optional: replay through one hipGraphLaunch (launch-bound shapes) */
int vh_plan_run(vh_ctx* ctx, const vh_plan* plan);
int vh_plan_num_ops(const vh_plan* plan);
int vh_plan_destroy(vh_plan* plan);

/* ---- K1: one-time weight preparation ------------------------------------
 * MPConv.forward's weight path, training/models.py:115-120 (normalize :37-42):
 *   w_hat[o] = w[o] / (1e-4 + ||w[o]||_2 / sqrt(fan_in)) * gain / sqrt(fan_in)
 * written in the layout the GEMM kernels stage from:
 *   wt[k/4][o][k%4],  k = tap * cin_pad + ci,  tap = ky*3+kx (taps = 1 or 9),
 * zero-filled for ci >= cin; cin_pad is a multiple of 32 and k_pad == taps*cin_pad.
 * `gain` is read from device memory if gain_ptr != NULL (emb_gain / out_gain are
 * 0-d Parameters, :157,:345), else gain_value is used.
 * `dst_col0`/`dst_cols` place the block into a wider matrix (batched emb_linear). */
typedef struct {
    const float* w;      /* [cout][cin][taps] (PyTorch OIHW, contiguous) */
    int cout, cin, taps;
    int cin_pad, k_pad;
    const float* gain_ptr;
    float gain_value;
    float* wt;           /* [k_pad/4][dst_cols][4] */
    int dst_col0, dst_cols;
    int split;           /* 1: bf16 hi/lo split, wt[(k/8)*2 + hl][dst_cols][8 bf16], hl = 0 hi / 1 lo (same byte count);
                            2: the same split, output-channel major: wt[col][(k/8)*2 + hl][8 bf16] (VH_CONV_GLDS256) */
    int k_off, k_stride; /* split = 2 only, both 0 by default: this weight's K range starts at k_off of rows that are k_stride long - the fused
                            (3x3 + 1x1) weight of vh_conv_args.src1 is two calls, the 3x3 part at 0 and the 1x1 part at 9*cin_pad */
} vh_prep_weight_args;
int vh_prep_weight(vh_ctx* ctx, const vh_prep_weight_args* a);

/* ---- K2/K3/K4 (+K5,K7,K8,K10,K12 fused): implicit-GEMM convolution --------
 * out[m][o] = epi( sum_{tap,ci} pro( src[pixel(m)+tap][ci] ) * wt[tap*cin_pad+ci][o] )
 * F.conv2d(x, w, padding=k//2) / x @ w.t() of MPConv.forward, :123-126, with
 *   source  : one or two NHWC tensors read as a channel concat, each with its own
 *             scalar weight (mp_cat :78-84, never materialised);
 *             up=1 reads the sources at half resolution, nearest-replicated
 *             (resample 'up' :60-61);
 *   pro     : VH_PRO_SILU applies mp_silu (:66-67) to the concatenated input;
 *   epi     : VH_EPI_STORE       y
 *             VH_EPI_SCALE_SILU  mp_silu(y * cvec[row(m)][o])            (:175-176)
 *             VH_EPI_MPSUM       clip(res[m][o]*ta + y*tb, +-clip)       (mp_sum :72-73, clip :204-205)
 *                                res_up=1 reads res at half resolution (the block input of an 'up' block).
 * fp32 operands on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate). */
enum { VH_PREC_F32 = 0, VH_PREC_BF16X3 = 1 };
/* VH_PREC_BF16X3: fp32 emulated by a bf16 hi/lo split (x = hi + lo; a*b ~= ah*bh + ah*bl + al*bh on
 * v_mfma_f32_32x32x16_bf16, fp32 accumulate).  The source must be in the "S8" layout: per pixel, per group
 * of 8 channels, 8 bf16 hi followed by 8 bf16 lo (4 bytes per channel, channel count a multiple of 32, pad
 * channels zero), produced by vh_split / vh_pixnorm / a vh_conv S8 epilogue; weights from vh_prep_weight
 * with split=1.  Scaling, mp_silu and the mp_cat concat are applied by the producer of the S8 tensor. */
enum { VH_CONV_TILE128 = 0, VH_CONV_GLDS256 = 1 };   /* 128x128 register-staged tile | 256-wide direct-to-LDS tile */
enum { VH_PRO_NONE = 0, VH_PRO_SILU = 1 };
enum { VH_EPI_STORE = 0, VH_EPI_SCALE_SILU = 1, VH_EPI_MPSUM = 2, VH_EPI_QKV = 3 };
/* VH_EPI_QKV (1x1 conv, VH_CONV_GLDS256): the attn_qkv / x_attn_kv convolution writes the attention operands itself
 * instead of an fp32 tensor that vh_qkv_split_x3 would read back: per pixel and head, q / k / v are RMS-normalised
 * over their D channels (normalize(dim=2), training/models.py:192-194, :279-293) in the accumulators and stored as
 * vh_qkv_split_x3 stores them (same formats, same buffers, same arguments).  The OUTPUT CHANNELS of the weight must be
 * ordered [head][j][d] (o' = (head*nj + j)*D + d) instead of the reference's (head*D + d)*nj + j, so that a 64- (D = 64) or
 * 32-column (D = 32) accumulator slab is one (head, j): permute the rows of w before vh_prep_weight.
 * Requires D = cout / (heads*nj) == 64 or 32, s % 32 == 0, koff % 16 == 0, out == NULL. */
/* Extra S8 outputs of a convolution (patch-resident kernel only, see `tile`): the result, scaled and optionally passed through mp_silu, written
 * straight into a channel range of a wider S8 tensor - the half of a decoder block's `mp_silu(mp_cat(x, skip))` input (training/models.py:78-84,
 * :174, :403) that this convolution produces, in the form (and with the bits) vh_split would write from the fp32 result:
 *   sink[pixel][c_off + o] = split( silu ? mp_silu(scale * y[pixel][o]) : scale * y[pixel][o] ),   rows of c_total channels. */
typedef struct {
    void* ptr;             /* NULL: unused */
    int c_total, c_off;    /* channels per pixel of the destination tensor (multiple of 32); first channel written (multiple of 32) */
    float scale;           /* the mp_cat weight of this half */
    int silu;              /* 1: mp_silu after scaling (conv_res0's input), 0: raw (conv_skip's input) */
} vh_s8_sink;
typedef struct {
    float* q; void* k; void* v;            /* as vh_qkv_split_args (q unused for nj == 2) */
    int heads, nj, rows_per_b, koff, kl;   /* s = h*w of the convolution; rows = its rows */
    float qscale;
} vh_qkv_epilogue;
typedef struct {
    const float* src0; const float* src1;  /* src1 may be NULL.  fp32: a channel concat with src0 (mp_cat).  bf16x3 + VH_CONV_GLDS256 + taps == 9 (no `up`): an S8
                                              tensor of c1 channels at the output resolution that enters as a 1-TAP TAIL SEGMENT of the K loop,
                                                  out = epi( sum over (tap, ci) of src0[pixel+tap][ci] wt[tap*cin_pad+ci][o]  +  sum over ci of src1[pixel][ci] wt[9*cin_pad+ci][o] ),
                                              i.e. a 3x3 and a 1x1 convolution of two inputs summed in one GEMM: conv_res1 + conv_skip of a decoder
                                              block, `x = mp_sum(conv_skip(x_cat), conv_res1(y), t)` training/models.py:184-186 with the mp_sum
                                              coefficients folded into the two weights and VH_EPI_STORE + clip as the epilogue. */
    int c0, c1;                            /* channels of each source (multiples of 4) */
    float scale0, scale1;
    int rows, h, w;                        /* OUTPUT geometry: rows images of h x w pixels */
    int up;
    int taps;                              /* 1 or 9 */
    int pro;
    const float* wt; int cin_pad, k_pad;   /* from vh_prep_weight: cin_pad % 32 == 0, k_pad == taps*cin_pad (+ c1 with a bf16x3 tail segment) */
    const float* zeros; size_t zeros_bytes; /* a device buffer of zeros, >= cin_pad*4 + 64 bytes: out-of-image taps and pad
                                              channels are read from it instead of being masked */
    int cout;
    float* scratch; size_t scratch_floats; /* optional (VH_CONV_GLDS256): split-K partial sums for grids that would leave most
                                              CUs idle; any stream-ordered temporary, reused by every call */
    float* out;                            /* [rows*h*w][cout] fp32; may be NULL if out_s8 is given */
    void* out_s8; int out_s8_c;            /* optional S8 copy of the result (cout % 32 == 0, out_s8_c == cout) */
    int prec;                              /* VH_PREC_* */
    int kernel;                            /* VH_CONV_TILE128 (weights split 0/1) or VH_CONV_GLDS256 (bf16x3 only, weights split 2) */
    int epi;
    const float* cvec; int cvec_ld;        /* SCALE_SILU: cvec[row*cvec_ld + o] */
    const float* res; int res_up;          /* MPSUM */
    const float* res_scale;                /* MPSUM, optional [rows*h*w] (not with res_up): the residual is res[m][o] * res_scale[m] - a pixel-normalised
                                              tensor given as its raw values and vh_pixnorm's per-pixel factor */
    float ta, tb, clip;                    /* clip <= 0: no clipping; VH_EPI_STORE clips too when clip > 0 */
    const vh_qkv_epilogue* qkv;            /* VH_EPI_QKV only */
    int stagger;                           /* VH_CONV_GLDS256 scheduling hint: 0 = library default, 1 = stagger the DMA issue of SIMD partner
                                              waves, 2 = do not.  Results are identical (bit for bit). */
    int korder;                            /* VH_CONV_GLDS256, taps == 9: order in which the K loop walks (tap, 32-channel chunk) tiles.
                                              VH_KORDER_AUTO: by input size (chunk-major once the input outgrows the Infinity Cache);
                                              VH_KORDER_TAP: all chunks of tap 0, then tap 1, ...; VH_KORDER_CHUNK: the 9 taps of chunk 0, then of
                                              chunk 1, ... (not with `up`: falls back to tap-major).  The sum is the same set of products in a
                                              different order: results agree to fp32 rounding, not bit for bit. */
    int tile;                              /* VH_CONV_GLDS256: workgroup tile, VH_TILE_AUTO (by shape and grid size) or a forced shape
                                              (pixels x output channels): VH_TILE_256x128, VH_TILE_256x256 (needs cout % 256 == 0),
                                              VH_TILE_512x128, VH_TILE_512x64 and VH_TILE_256x64 (need cout <= 64; the latter runs two
                                              workgroups per CU), VH_TILE_256x192 (3x3 only).  A forced shape disables split-K unless
                                              the grid is small; every shape computes the same sums in the same order.
                                              VH_TILE_PATCH16: the patch-resident kernel (conv_patch.hip) - 3x3, cout % 32 == 0 (or cout <= 16 with a plain fp32 store): one
                                              workgroup per 16x16-pixel output tile of one image, its (16+2)^2-pixel input patch staged once per
                                              32-channel chunk and read in place by the nine taps; chunk-major K order (sums agree with the other
                                              tiles to fp32 rounding). */
    vh_s8_sink sink[2];                    /* optional extra S8 outputs (see vh_s8_sink); only a launch that takes the patch-resident kernel writes them
                                              (vh_conv_takes_patch() == 1 or tile == VH_TILE_PATCH16), any other is refused.  With a sink, `out` and
                                              `out_s8` may both be NULL. */
    int tail_f32;                          /* 1: the 1-tap tail segment is read from fp32 NHWC tensors instead of an S8 one - `src1` (c1 channels, times scale1)
                                              and optionally `src2` (c2 channels, times scale2) as a channel concat, i.e. mp_cat(x, skip) itself
                                              (training/models.py:78-84): the scaled values are split into bf16 hi / lo while the tail is staged, the
                                              bits vh_split would have written as the raw S8 form.  c1, c2 multiples of 32, k_pad = 9*cin_pad + c1 + c2.
                                              Patch-resident kernel only (ask vh_conv_takes_patch with these fields set; refused otherwise). */
    const float* src2; int c2; float scale2;
    int src_f32;                           /* 1 (VH_PREC_BF16X3 + VH_CONV_GLDS256, 3x3, patch-resident kernel only - ask vh_conv_takes_patch): src0 (c0 channels, times
                                              scale0) and optionally src1 (c1 channels, times scale1) are fp32 NHWC tensors forming the input as a channel concat,
                                              `pro` applies - the meaning these fields have for VH_PREC_F32 - and the kernel makes the bf16 hi / lo split itself while
                                              it stages each 32-channel chunk of its input patch: mp_silu(mp_cat(x, skip)) never exists in memory.  c0, c1
                                              multiples of 32, cin_pad = c0 + c1; no tail segment. */
} vh_conv_args;
enum { VH_KORDER_AUTO = 0, VH_KORDER_TAP = 1, VH_KORDER_CHUNK = 2 };
enum { VH_TILE_AUTO = 0, VH_TILE_256x128 = 1, VH_TILE_256x256 = 2, VH_TILE_512x128 = 3, VH_TILE_512x64 = 4, VH_TILE_256x64 = 5, VH_TILE_256x192 = 7,
       VH_TILE_PATCH16 = 8 };
int vh_conv(vh_ctx* ctx, const vh_conv_args* a);
/* 1 if vh_conv would run these arguments on the patch-resident kernel (eligible and chosen by the size rule / tile / knob), 0 if not,
 * negative on invalid arguments: lets a host decide whether it may attach sinks. */
int vh_conv_takes_patch(const vh_conv_args* a);

/* ---- K6 (+K9): pixel norm with optional 2x2 mean pooling ------------------
 * normalize(x, dim=1) :37-42 applied after resample 'down' (:58-59, a 2x2 mean):
 *   out[p][c] = v[c] / (1e-4 + ||v||_2 / sqrt(C)),  v = x[p] or mean of the 2x2 block.
 * rows/h/w are the OUTPUT geometry; with pool=1 the input is [rows][2h][2w][c]. */
typedef struct {
    const float* in; float* out;
    int rows, h, w, c;
    int pool;
    int norm;     /* 0: pooling only */
    void* out_s8; /* optional: mp_silu(out) in the S8 layout (c % 32 == 0), the conv_res0 input of :174 */
    float* scale_out; /* optional [rows*h*w]: the per-pixel factor 1 / (1e-4 + ||v||/sqrt(C)).  With it (and out_s8) `out` may be NULL: the
                         normalised tensor is then never materialised - its one other reader, the residual of conv_res1 (:184), takes
                         x * scale[pixel] instead (vh_conv_args.res_scale) */
} vh_pixnorm_args;
int vh_pixnorm(vh_ctx* ctx, const vh_pixnorm_args* a);

/* ---- fp32 NHWC -> S8 (bf16 hi/lo split) with the conv_res0 prologue -----
 * out = split( pro( concat(scale0*src0, scale1*src1) ) ), zero-padded to c_pad channels (multiple of 32):
 * mp_cat :78-84 and mp_silu :66-67,:174 applied once per element instead of once per tap. */
typedef struct {
    const float* src0; const float* src1;  /* src1 may be NULL */
    int c0, c1; float scale0, scale1;
    int pro;                               /* VH_PRO_* */
    long long npix; int c_pad;
    void* out;
    void* out_raw;                         /* optional second S8 output without the prologue (conv_skip's input) */
    int out_c_total, out_c_off;            /* 0, 0: the outputs are dense [npix][c_pad].  Otherwise they are rows of out_c_total channels and the
                                              c_pad channels produced here start at channel out_c_off (multiples of 32): the other channels of the
                                              row are written by someone else (a convolution's vh_s8_sink) */
} vh_split_args;
int vh_split(vh_ctx* ctx, const vh_split_args* a);

/* ---- K11 part 1: q/k/v split + per-head vector norm -----------------------
 * The view/normalize/unbind of :192-194, :279-293: the 1x1-conv output has channel
 * index (head*D + d)*nj + j (j = q,k,v for nj=3; k,v for nj=2); each D-vector is
 * divided by (1e-4 + ||.||_2/sqrt(D)).  q is additionally multiplied by qscale
 * (= log2(e)/sqrt(D): SDPA's 1/sqrt(D) :198 folded with the exp2 softmax).
 * Row `r` of `in` belongs to batch element r / rows_per_b and lands at key offset
 * koff + (r % rows_per_b) * s  (the sequence concat of :296-297, never materialised).
 * Q [b][head][s][D];  K,V [b][head][kl][D]. */
typedef struct {
    const float* in;      /* [rows][s][heads*D*nj] */
    int rows, s, heads, d, nj;
    int rows_per_b, koff, kl;
    float qscale;
    float* q; float* k; float* v;
} vh_qkv_split_args;
int vh_qkv_split(vh_ctx* ctx, const vh_qkv_split_args* a);

/* ---- K11 part 2: attention ------------------------------------------------
 * softmax(q k^T) v of F.scaled_dot_product_attention :198,:305 (no mask, no dropout),
 * flash-style on fp32 MFMA; logits are in log2 units (see qscale above).
 * n_zero_keys: that many extra keys with logit 0 and value 0 are added to the
 * softmax denominator in closed form — the zero features of the unconditional
 * guidance network (:727-736; normalize(0) = 0 gives k = v = 0) — without reading them.
 * out is NHWC [b][s][heads*D] with channel = head*D + d  (:199,:308). */
typedef struct {
    const float* q; const float* k; const float* v;
    int b, heads, s, kl, d;   /* d = 32 or 64 */
    float n_zero_keys;
    float* out;
    int out_s8;               /* vh_attention_x3 only: write `out` in the S8 (bf16 hi/lo) layout for a bf16x3 attn_proj */
    float logit_bound;        /* vh_attention_x3 only: caller's guarantee |q.k| <= logit_bound for every pair (q as passed,
                               * i.e. in log2 units); 0 = none.  vh_qkv_split* output satisfies sqrt(D)*log2(e): the
                               * head vectors are RMS-normalised.  With 0 < bound <= 64 no running maximum is kept
                               * (exp2 of the raw logits cannot overflow fp32), the result is the same softmax. */
} vh_attention_args;
int vh_attention(vh_ctx* ctx, const vh_attention_args* a);

/* bf16x3 variants of the two attention entry points (same argument structs): the split kernel writes
 * K in the S8 layout [b][head][klp][D/8][hi x8|lo x8] and V transposed, [b][head][D][hl][klp] bf16 with key
 * positions permuted inside each group of 16 (bits 2 and 3 swapped), klp = kl rounded up to 64; both
 * buffers must hold b*heads*klp*D*4 bytes.  vh_attention_x3 evaluates every product as
 * hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation and fp32 softmax statistics. */
int vh_qkv_split_x3(vh_ctx* ctx, const vh_qkv_split_args* a);
int vh_attention_x3(vh_ctx* ctx, const vh_attention_args* a);

/* ---- K13 + K4 + K7 + K5: noise/pose embedding -----------------------------
 * emb = mp_silu(mp_sum(emb_noise(fourier(c_noise)), emb_label(geometry), t)) :388-391,:485-488
 * with c_noise = ln(sigma)/4 * time_scale (:638,:667).  sigma row for output row r is
 * sigma[r*sigma_stride]; geometry row r is geometry + r*label_dim (contiguous, so the
 * dual-source view(B,40) of :678 is label_dim=40 over the same buffer).
 * geometry == NULL or label_dim == 0 skips the pose term.  w_* come from vh_prep_weight. */
typedef struct {
    const float* sigma; int sigma_stride; float time_scale;
    const float* geometry; int label_dim; float geometry_scale;  /* 0 for uncond nets (:631) */
    const float* freqs; const float* phases; int cnoise;
    const float* w_noise; int w_noise_kpad;
    const float* w_label; int w_label_kpad;
    float label_balance;
    int rows, cemb;
    int raw;              /* 1: emb = emb_noise(fourier(c_noise)) only (logvar head, :687) */
    float* emb;           /* [rows][cemb] */
} vh_embed_args;
int vh_embed(vh_ctx* ctx, const vh_embed_args* a);

/* ---- K4 batched: every block's emb_linear in one launch -------------------
 * cvec[r][o] = sum_k emb[r][k] * wt[k][o] + 1    (c of :175; gains folded by vh_prep_weight) */
typedef struct {
    const float* emb; int rows, cemb;
    const float* wt; int k_pad; int cols;
    float bias;
    float* out;           /* [rows][cols] */
} vh_linear_args;
int vh_linear(vh_ctx* ctx, const vh_linear_args* a);

/* ---- K14 + input assembly -------------------------------------------------
 * Builds the NHWC network input: up to 4 channel segments, then a constant-ones
 * channel (the bias channel of :394,:495,:544), zero-padded to c_pad.
 *   kind 0: NCHW image   [*,c,h,w], source row = r*row_mul, optionally scaled by
 *           c_in(sigma[r*row_mul]) = 1/sqrt(sigma_data^2+sigma^2)  (:637,:639)
 *   kind 1: NHWC features [*,h,w,c], source row = r*row_mul */
typedef struct {
    const float* ptr; int kind; int c; int c_src; int row_mul; int scale_cin;
} vh_segment;   /* c channels are taken from a source tensor that has c_src channels (c <= c_src) */
typedef struct {
    vh_segment seg[4]; int nseg;
    const float* sigma; float sigma_data;
    int rows, h, w, c_pad;
    float* out;
} vh_assemble_args;
int vh_assemble(vh_ctx* ctx, const vh_assemble_args* a);

/* D_x = c_skip * x + c_out * F_x  (:635-636,:683), NCHW out.
 * x row for output row r is r*row_mul (dual-source: 2). F is NHWC [rows][h][w][fc], first 3 used. */
typedef struct {
    const float* x; int row_mul; const float* f; int fc;
    const float* sigma; float sigma_data;
    int rows, c, h, w;
    float* out;
} vh_precond_out_args;
int vh_precond_out(vh_ctx* ctx, const vh_precond_out_args* a);

/* ---- K13 + K15: depth-warp Fourier features (config 5) --------------------
 * get_warped_features training/utils.py:204-216 (warp_image :189-201, decompose_geometry
 * :84-94): pixel centres are un-projected with the source depth, moved by the inverse
 * of tgt2src, projected with K_tgt, divided (NaN -> 0); both grids are Fourier-embedded
 * with the first 64 features of logvar_fourier (:96-101) per coordinate.
 * geometry is the normalised 20-vector; mean/std are the per-image-size statistics
 * (vivid_amd/geometry.py).  Outputs are NHWC [rows][s][s][128]. */
typedef struct {
    const float* depth;     /* NCHW src with depth in channel `depth_ch` of `src_c` channels */
    int src_c, depth_ch;
    const float* geometry;  /* [rows][20] */
    float mean[20]; float std[20];
    const float* freqs; const float* phases;
    int rows, s;
    float* grid_feat; float* warp_feat;
    const float* nonzero_flag;   /* optional, from vh_nonzero_flag over src[:, :3]: when it reads 0 (the source is all zero) both
                                    outputs are zero grids, the shortcut of training/models.py:647-648, decided on the device */
    float* uv_out;               /* optional [rows][s][s][2]: the warped coordinates (row, column) themselves, before the Fourier embedding
                                    (warp_image training/utils.py:189-201) - what tests hold against an fp64 evaluation */
} vh_warp_args;
int vh_warp_features(vh_ctx* ctx, const vh_warp_args* a);

/* flag[0] = 1.0f if any of the first c_used channels of any row of the NCHW tensor `in` [rows][c_total][hw] is non-zero, else
 * 0.0f: `torch.all(src[:, :3] == 0)` of training/models.py:647 without a host synchronisation. */
typedef struct {
    const float* in; int rows, c_used, c_total, hw;
    float* flag;
} vh_nonzero_args;
int vh_nonzero_flag(vh_ctx* ctx, const vh_nonzero_args* a);

/* ---- resample with a general filter (training/models.py:48-61), NHWC fp32 -----------------------------
 * The default filter [1,1] (2x2 mean / nearest replication) is fused into vh_pixnorm (pool) and vh_conv (up); any other
 * even-length `resample_filter` of Block (:139) goes through this kernel.  taps = f / sum(f):
 *   down: depthwise stride-2 correlation with outer(taps, taps), padding (L-1)/2      [rows][h][w][c] -> [rows][h/2][w/2][c]
 *   up  : depthwise stride-2 transposed convolution with 4*outer(taps, taps)          [rows][h][w][c] -> [rows][2h][2w][c] */
typedef struct {
    const float* in; float* out;
    int rows, h, w, c;          /* INPUT geometry */
    int up;
    int ntaps; float taps[8];
} vh_resample_args;
int vh_resample(vh_ctx* ctx, const vh_resample_args* a);

/* ---- K17: pixel codec (training/encoders.py:58-62) ---------------------------
 * decode=0: out_f32[i] = in_u8[i] / 127.5 - 1          (StandardRGBEncoder.encode_latents)
 * decode=1: out_u8[i]  = uint8(clip(in_f32[i] * 127.5 + 128, 0, 255))   (StandardRGBEncoder.decode; truncation
 *           toward zero like torch's float -> uint8 cast) */
typedef struct {
    const void* in; void* out; size_t n; int decode;
} vh_codec_args;
int vh_codec(vh_ctx* ctx, const vh_codec_args* a);

/* ---- add_depth (training/utils.py:129-139) ----------------------------------
 * Appends a depth map as the last source channel: out[r] = cat(src[r] (c channels), d'), NCHW, with
 * inv_norm=1: d' = ((1/d) / max_r(1/d) - 0.4947) / 0.2294   (per-sample max over the whole map), else d' = d.
 * The depth map itself comes from an external monocular depth model (out of scope). */
typedef struct {
    const float* src; int c;          /* [rows][c][h][w] */
    const float* depth;               /* [rows][1][h][w] */
    int rows, h, w, inv_norm;
    float* out;                       /* [rows][c+1][h][w] */
} vh_add_depth_args;
int vh_add_depth(vh_ctx* ctx, const vh_add_depth_args* a);

/* ---- bilinear resize of NCHW fp32 images (the SR hand-off of generate_images.py:299-302,322) -----------
 * torchvision.transforms.functional.resize on tensors = F.interpolate(mode="bilinear", align_corners=False,
 * antialias=...): source coordinate (i+0.5)*scale-0.5; with antialias and scale > 1 the triangle filter is widened
 * to `scale` (aten upsample_bilinear2d_aa); weights are normalised per output pixel. */
enum { VH_RESIZE_BILINEAR = 0, VH_RESIZE_BICUBIC = 1 };
typedef struct {
    const float* in; float* out;
    int planes;            /* rows * channels */
    int hin, win, hout, wout;
    int antialias;         /* bilinear, align_corners == 0 only */
    int mode;              /* VH_RESIZE_*; bicubic = cubic convolution with A = -0.75, border-clamped taps (aten upsample_bicubic2d) */
    int align_corners;     /* 1: source coordinate i*(in-1)/(out-1) */
    const float* ch_scale; const float* ch_bias; int channels;   /* optional per-channel affine on the result: y*ch_scale[c] + ch_bias[c],
                                                                    c = plane % channels (the /255 and ImageNet normalisation of depth_prepare) */
} vh_resize_args;
int vh_resize_bilinear(vh_ctx* ctx, const vh_resize_args* a);   /* mode must be VH_RESIZE_BILINEAR */
/* General form: also the depth front end of configuration 5 - depth_prepare (bicubic to 518, align_corners, training/utils.py:107-115)
 * and the depth map's way back to image size (bilinear, align_corners, :125).  The depth model between them is external. */
int vh_resize(vh_ctx* ctx, const vh_resize_args* a);

/* ---- K16: sampler update (generate_images.py:93-94,108-109) ---------------
 * d = (x - D)/t_hat;  Euler: x_next = x + (t_next - t_hat) * d           (d_out written)
 * Heun : x_next = x + (t_next - t_hat) * (0.5*d_prev + 0.5*(x_probe - D)/t_next)
 * with classifier-free guidance folded in: D = ref.lerp(D_cond, guidance) (:62) when d_ref != NULL.
 * x rows are read at row*row_mul (dual-source row pairs) and x_next is written to every
 * row of the pair (:96-98,110-111). */
typedef struct {
    const float* x_hat; const float* x_probe;   /* x_probe NULL: Euler step */
    const float* d_cond; const float* d_ref; float guidance;
    float* d_cur;          /* Euler: written; Heun: read */
    float t_hat, t_next;
    int rows, row_mul; size_t row_elems;
    float* x_next;
} vh_sampler_step_args;
int vh_sampler_step(vh_ctx* ctx, const vh_sampler_step_args* a);

/* ---- NHWC <-> NCHW of fp32 tensors --------------------------------------------------------------------
 * The feature lists of `return_features` / `inject_features` are NCHW in the reference (training/models.py:664-670); the library keeps
 * activations NHWC.  rows images of hw pixels x c channels; to_nchw = 1: [rows][hw][c] -> [rows][c][hw], 0: the other way. */
typedef struct { const float* in; float* out; int rows, c, hw; int to_nchw; } vh_layout_args;
int vh_layout(vh_ctx* ctx, const vh_layout_args* a);

/* out[i] = a[i] + s * b[i], rounded after the multiply and after the add like torch's `a + s * b` (a NULL: s * b[i]; b NULL: the constant s):
 * the sampler's x0 = noise * t0 and churn x + sqrt(t_hat^2 - t_cur^2) * S_noise * eps (generate_images.py:72, :81), the super-resolution
 * conditioning noise cond + noisy_sr * eps (training/models.py:658), `torch.full`. */
typedef struct { const float* a; const float* b; float s; float* out; size_t n; } vh_axpy_args;
int vh_axpy(vh_ctx* ctx, const vh_axpy_args* a);

/* ---- FID / PSNR statistics (calculate_metrics.py:147, 158-172) -------------------------------------------
 * vh_moments: outer[fa][fb] += A^T B and (optionally) sum_a[fa] += column sums of A, for detector features A [n][fa],
 * B [n][fb] (fp32, row-major; B may be A).  Products and sums are fp64 on v_mfma_f64_16x16x4_f64, i.e. exactly the
 * reference's `features.to(float64)`; `features.T @ features` is A == B, the off-diagonal block of the joint
 * [image | source] moments is A = image features, B = source features.  The accumulators persist across batches and are
 * all_reduced once at the end (calculate_metrics.py:176-182). */
typedef struct {
    const float* a; const float* b;
    int n, fa, fb;
    double* outer; double* sum_a;          /* sum_a may be NULL */
} vh_moments_args;
int vh_moments(vh_ctx* ctx, const vh_moments_args* a);

/* acc[0] += sum over images of 10*log10(255^2 / mean((x - y)^2)), images as [images][elems] uint8 or fp32 on the
 * [0,255] scale (calculate_metrics.py:147). */
enum { VH_U8 = 0, VH_F32 = 1 };
typedef struct {
    const void* x; const void* y; int images; size_t elems; int dtype;
    double* acc;
    double* per_image;         /* scratch, `images` doubles: per-image values, folded into acc[0] in index order (bit-reproducible) */
} vh_psnr_args;
int vh_psnr_sum(vh_ctx* ctx, const vh_psnr_args* a);

/* ---- whole-network evaluation ---------------------------------------------------------------------------------
 * One NVPrecond evaluation, D_x = net(src, x, sigma, geometry, cond), as ONE call: NVPrecond._forward_dualsource
 * training/models.py:628-689 (dual_source = 1) / NVPrecond.forward :691-749 (dual_source = 0), i.e. EDM preconditioning :633-639, the
 * optional depth-warp / super-resolution input assembly :643-661, UNetEncoder.forward :536-570 on the source rows, XAttnUNet.forward
 * :483-518 on the target rows, D_x :683 - for hosts that do not run Python (last row).  The library generates the
 * architecture from the constructor arguments (UNet.__init__ :322-384, XAttnUNet :413-480, UNetEncoder trimming :524-534,
 * SRXAttnUNet :576-582), names the parameters with the reference's state_dict keys, prepares the weights once (:115-120) and
 * records the evaluation per batch size as a vh_plan over a workspace the caller owns; a call copies the inputs in, replays, and
 * copies D_x out - no allocation, no host synchronisation.  Arithmetic: bf16x3 (see VH_PREC_BF16X3) on the direct-to-LDS kernels -
 * channel counts must be multiples of 32, attention heads 64 (32 for super_res UNets) channels - or exact fp32 (vh_net_config.fp32).
 * An `uncond` net has no encoder: the zero features of :727-736 enter the attention in closed form and `src` / `geometry` may be NULL.
 * vivid_amd.NVPrecond (engine.py) emits the same op sequence from Python; tests/test_hip_net_c.py compares the two bit for bit.
 *
 *   vh_net_create(ctx, &cfg, &net);
 *   for i < vh_net_num_params(net): vh_net_param_info(net, i, &name, &ndim, shape); vh_net_bind_param(net, name, device_fp32_ptr);
 *   vh_net_prepare(net, buf, vh_net_prepared_bytes(net));                 // again after the weights change
 *   vh_net_record(net, B, workspace, vh_net_workspace_bytes(net, B));     // once per batch size
 *   vh_net_run(net, B, src, x, sigma, geometry, cond, cond_noise, out);   // any number of times, on the context's stream
 */
typedef struct vh_net vh_net;
typedef struct {
    int img_resolution, img_channels;          /* img_channels must be 3 (:480) */
    int source_label_dim, target_label_dim;    /* 20 / 40 in the reference's presets (train_nvs.py) */
    int model_channels;                        /* 128 (vivid-base), 64 (vivid-sr) */
    int channel_mult[8]; int num_levels;       /* 1,2,3,4 and 4 */
    int num_blocks;                            /* 3 */
    int attn_resolutions[8]; int num_attn_resolutions;   /* 16, 8 and 2 - absolute resolutions (:331) */
    int extra_attn;                            /* block index that also gets attention at every level but the first, -1 = none (:366) */
    int channel_mult_noise, channel_mult_emb;  /* 0 = the reference's None (:340-341) */
    double label_balance, concat_balance, res_balance, attn_balance;  /* 0.5, 0.5, 0.3, 0.3 (double: the mp_sum / mp_cat coefficients are derived
                                                                          from them in double, like the reference's Python floats, :72-84) */
    double clip_act;                           /* 256; <= 0 = None */
    double sigma_data;                         /* 0.5 */
    int logvar_channels;                       /* 128 */
    int super_res, no_time_enc, depth_input, warp_depth_coor, uncond;
    int dual_source;                           /* 1: the HEAD forward (two source rows per target), 0: the single-source forward */
    float geom_mean[20], geom_std[20];         /* warp_depth_coor only: the geometry statistics for this image size (training/utils.py:38-44, 77-78) */
    double noisy_sr;                           /* super_res: the net sees cond + noisy_sr * N(0,1) on EVERY forward, also at inference (training/models.py:658,
                                                  0.25 in --preset=vivid-sr).  The draws are the caller's: `cond_noise` of the run calls, required while this is != 0 */
    int resample_ntaps; float resample_filter[8];   /* Block.resample_filter (:139): 0 taps = the default [1,1] (fused into vh_pixnorm / vh_conv); any other
                                                  even-length filter (2..8 taps, unnormalised like the reference's argument) runs vh_resample per
                                                  up / down block */
    int fp32;                                  /* 0: bf16x3 arithmetic (the default, the timed path).  1: exact fp32 - VH_PREC_F32 convolutions on the register-staged
                                                  tile, vh_qkv_split / vh_attention, no S8 tensors (vivid_amd.NVPrecond(precision="fp32")); any channel count */
} vh_net_config;
int vh_net_create(vh_ctx* ctx, const vh_net_config* cfg, vh_net** out);
int vh_net_destroy(vh_net* net);
int vh_net_num_params(const vh_net* net);
/* name: a state_dict key of the reference's NVPrecond; shape: up to 4 dims (OIHW for convolutions), unused dims 1; 0-d gains have ndim 0 */
int vh_net_param_info(const vh_net* net, int i, const char** name, int* ndim, int* shape);
int vh_net_bind_param(vh_net* net, const char* name, const float* device_ptr);   /* contiguous fp32, stays owned by the caller */
size_t vh_net_prepared_bytes(const vh_net* net);
int vh_net_prepare(vh_net* net, void* buffer, size_t bytes);                     /* normalised / re-laid-out weights live in `buffer` */
size_t vh_net_workspace_bytes(vh_net* net, int batch);                           /* 0 on error (see vh_last_error) */
int vh_net_record(vh_net* net, int batch, void* workspace, size_t bytes);
/* src [rows][3|4][R][R], x [rows][3][R][R], sigma [rows], geometry [rows][source_label_dim], cond [batch][3][R][R] (super_res),
 * cond_noise [batch][3][R][R] standard-normal draws (super_res with noisy_sr != 0, else NULL), out [batch][3][R][R];
 * rows = batch * (dual_source ? 2 : 1); device fp32, contiguous.  Odd rows of x / sigma are ignored in dual-source mode, like the
 * reference (:676-678). */
int vh_net_run(vh_net* net, int batch, const float* src, const float* x, const float* sigma, const float* geometry, const float* cond,
               const float* cond_noise, float* out);
/* The two halves of an evaluation as separate programs (training/models.py:664-667 and :676-683), for samplers: the encoder sees
 * (src, sigma, geometry) only - never x - and a sampler knows its noise levels in advance, so it can evaluate the encoder ONCE per
 * level (the Heun probe of step i and the Euler call of step i+1 share one) and one level ahead on another stream.
 *   VH_NET_FEATURES  encoder only; the features stay in that program's workspace (two slots, so that one can be filled while the
 *                    other is read);
 *   VH_NET_BOUND     UNet only, reading the features of the VH_NET_FEATURES program of the same (slot, batch) in place - record that
 *                    one first; re-recording the VH_NET_FEATURES program drops the VH_NET_BOUND one (it holds addresses into the other's
 *                    workspace).  `src` is read only by warp_depth_coor nets (may be NULL otherwise);
 *   VH_NET_INJECT    UNet only, on a feature list the CALLER supplies (inject_features, :664-665), see vh_net_run_inject.
 * vh_net_encode + vh_net_run_bound == vh_net_run, bit for bit.  The two VH_NET_BOUND programs of a batch size may be recorded over the SAME
 * workspace (they are replayed one after the other and differ only in the feature pointers they read); the two VH_NET_FEATURES programs
 * need one each. */
enum { VH_NET_FULL = 0, VH_NET_FEATURES = 1, VH_NET_BOUND = 2, VH_NET_INJECT = 3 };
size_t vh_net_workspace_bytes_mode(vh_net* net, int mode, int batch);
int vh_net_record_mode(vh_net* net, int mode, int slot, int batch, void* workspace, size_t bytes);
int vh_net_encode(vh_net* net, int slot, int batch, const float* src, const float* sigma, const float* geometry);
int vh_net_run_bound(vh_net* net, int slot, int batch, const float* src, const float* x, const float* sigma, const float* geometry, const float* cond,
                     const float* cond_noise, float* out);

/* The rest of NVPrecond.forward's protocol (training/models.py:664-670, 685-688), for hosts that want the tensors themselves:
 *   vh_net_num_features / vh_net_feature_shape   the encoder's feature list: entry i is [rows][channels][res][res] (rows = batch * sources)
 *   vh_net_features      `return_features=True`: runs the VH_NET_FEATURES program of slot 0 (record it first) and writes the list as NCHW
 *                        tensors into features_out[i] (device pointers, caller-owned)
 *   vh_net_run_inject    `inject_features=list`: the UNet on a caller-supplied NCHW feature list (VH_NET_INJECT program of slot 0), e.g. the
 *                        zero list, edited features, or the output of vh_net_features
 *   vh_net_logvar        `return_logvar=True`: logvar[batch] = logvar_linear(logvar_fourier(ln(sigma)/4)) (:685-688) - a function of sigma alone;
 *                        sigma [rows] as for vh_net_run.  Not recorded: one small launch on the context's stream. */
int vh_net_num_features(const vh_net* net);
int vh_net_feature_shape(const vh_net* net, int i, int* channels, int* res);
int vh_net_features(vh_net* net, int batch, const float* src, const float* sigma, const float* geometry, float* const* features_out);
int vh_net_run_inject(vh_net* net, int batch, const float* src, const float* x, const float* sigma, const float* geometry, const float* cond,
                      const float* cond_noise, const float* const* features_in, float* out);
int vh_net_logvar(vh_net* net, int batch, const float* sigma, float* logvar);

/* ---- edm_sampler (generate_images.py:43-118) from C ------------------------------------------------------------------------------
 * The guided Heun sampler over vh_net evaluations: rho schedule (:68-70), x0 = noise * t0 (:72), optional churn (:78-84), Euler step and
 * Heun correction (:87-114), classifier-free guidance D = ref.lerp(D, guidance) (:58-62), dual-source row handling (:90-98, :106-111),
 * `no_time_enc` feature reuse (:52-53) - vivid_amd.edm_sampler restated against this ABI, bit for bit (tests/test_hip_net_c.py), with its
 * scheduling: when the VH_NET_FEATURES and VH_NET_BOUND programs of both slots are recorded for `batch`, the encoder runs once per noise
 * level, one level ahead on a second stream; otherwise every call is a whole VH_NET_FULL evaluation.  gnet: the guidance net (an `uncond`
 * vh_net with its VH_NET_FULL program recorded), NULL or == net for none (guidance must then be 1).
 * Randomness is the caller's: `randn(user, dst, n, stream)` must enqueue on `stream` a fill of dst[0..n) (device) with N(0,1) draws; it is
 * called for churn noise (n = rows*3*R*R, once per churned step) and for the super-resolution conditioning noise (n = batch*3*R*R, once per
 * denoiser call of a net with noisy_sr != 0), in the order vivid_amd.edm_sampler consumes torch's generator.
 * src [rows][3|4][R][R], noise [rows][3][R][R], labels [rows][source_label_dim] (NULL for none), cond [batch][3][R][R] or NULL,
 * out [batch][3][R][R] (the even rows of the final state in dual-source mode, :116-118).  workspace: vh_edm_sampler_workspace_bytes(). */
typedef void (*vh_randn_fn)(void* user, float* device_dst, size_t n, void* stream);
typedef struct {
    int num_steps;                     /* 32 */
    double sigma_min, sigma_max, rho;  /* 0.002, 80, 7 */
    double guidance;                   /* 1 = none */
    double S_churn, S_min, S_max, S_noise;   /* 0, 0, inf, 1 */
    const float* t_steps;              /* optional host array [num_steps + 1] (last = 0): these noise levels instead of the rho schedule - e.g. the
                                          levels torch computed, for bit-compatible runs (powf of this libm vs torch's vectorised pow: 1 ulp) */
    vh_randn_fn randn; void* randn_user;
    int guidance_overlap;              /* -1: by size (the guidance net on a second stream when the evaluation cannot fill the chip), 0 never, 1 always */
} vh_sampler_config;
size_t vh_edm_sampler_workspace_bytes(const vh_net* net, int batch);
int vh_edm_sampler(vh_net* net, vh_net* gnet, const vh_sampler_config* cfg, int batch, const float* src, const float* noise, const float* labels,
                   const float* cond, void* workspace, size_t workspace_bytes, float* out);

#ifdef __cplusplus
}
#endif
#endif /* VIVID_HIP_H */
