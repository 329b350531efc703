// Implicit-GEMM convolution / pointwise GEMM on fp32 MFMA for gfx950 (CDNA4).
//
// Replaces the F.conv2d / x @ w.t() call sites of MPConv.forward (reference
// training/models.py:123-126) together with the element-wise ops the reference runs
// around them: mp_silu on the input (:66-67, :174), the two-tensor mp_cat (:78-84,
// :403/:509/:563), nearest 2x upsampling (:60-61), the embedding scale + mp_silu on the
// output (:175-176), and mp_sum + clip with the residual (:72-73, :184, :204-205).
//
// GEMM view: M = rows*h*w output pixels, N = cout, K = taps*cin_pad.
//   A[m][k] is gathered from NHWC activations (k = tap*cin_pad + ci), B[k][n] are the
//   prepared weights wt[k/4][n][k%4].
// Tile: 128(M) x 128(N) x 32(K) per 256-thread workgroup; 4 waves as 2x2, each wave
//   64x64 = 2x2 v_mfma_f32_32x32x2_f32 tiles (exact fp32 products, fp32 accumulate).
// LDS: operands are stored as float4 = 4 consecutive k for one m (or n):
//   sA[k4][m ^ k4], sB[k4][n].  Lane l of a wave reads the float4 of k-group
//   k4 = 2*kg + (l>>5); its 4 components feed 4 successive MFMAs, so within one MFMA the
//   two lane halves supply k = 8kg+j and 8kg+4+j — a permutation of k that A and B share.
//   The XOR keeps the 8 staging lanes that hold the same pixel (different k4) on different
//   16-byte slots (ds_write_b128 is serviced 8 lanes at a time); reads stay conflict-free.
// Two arithmetic modes share all staging code (template parameter PREC):
//   VH_PREC_F32    — fp32 operands, v_mfma_f32_32x32x2_f32 (exact fp32 products);
//   VH_PREC_BF16X3 — fp32 emulated by a bf16 hi/lo split: x = hi + lo (hi = bf16(x), lo = bf16(x - hi)),
//                    a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16 with fp32
//                    accumulation (3 MFMAs per 16-deep k-slab instead of 8 fp32 ones: 5.3x the matrix rate;
//                    per-product relative error <= ~2^-16, gfx950 has no xf32).  Activations arrive
//                    pre-split in the "S8" layout (per pixel, per 8 channels: 8 bf16 hi then 8 bf16 lo —
//                    the same 4 bytes/channel as fp32, so the 16-byte staging units and their addresses
//                    are identical), written by the producing kernel's epilogue / vh_split / vh_pixnorm;
//                    weights are pre-split by vh_prep_weight.  LDS unit u = 2*(8-channel chunk) + {hi,lo}.
// Pipeline: global -> registers for tile t+1 is issued before the MFMAs of tile t, written
//   to the other LDS buffer after them; one barrier per K-tile.
#include "conv_common.h"

namespace {
using namespace vhconv;

constexpr int BM = 128, BN = 128, BK = 32, K4 = BK / 4;

template <int TAPS, int PREC>
__global__ __launch_bounds__(256, 2) void conv_igemm(const ConvK a) {
    __shared__ float4 sA[2][K4 * BM];
    __shared__ float4 sB[2][K4 * BN];

    const int t = threadIdx.x;
    const unsigned tile = xcd_tile_id();
    const int nt = tile % a.NT, mt = tile / a.NT;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- staging maps -----------------------------------------------------
    // K runs tap-major: k = tap*cin_pad + c, cin_pad a multiple of BK, so a K-tile is 32 channels of ONE tap.
    // Per tap each thread derives a base pointer for each of its 4 pixels (or the zero page when the tap falls
    // outside the image); inside a tap a K-tile only adds its channel offset.  Out-of-image taps and pad
    // channels read the zero page, so no select is needed after the load.
    const int k4a = t & 7, ma = t >> 3;          // A: 8 16-byte units x 32 pixels, 4 passes
    const int Hs = a.up ? (a.h >> 1) : a.h, Ws = a.up ? (a.w >> 1) : a.w;
    int py[4], px[4], pbase[4];
    bool pv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = m0 + ma + 32 * i;
        pv[i] = gm < a.M;
        const int g = pv[i] ? gm : 0;
        const int img = g / a.HW;
        const int rem = g - img * a.HW;
        py[i] = rem / a.w;
        px[i] = rem - py[i] * a.w;
        pbase[i] = img * Hs * Ws;
    }
    const int nb = t & 127, k4b = t >> 7;        // B: 128 columns x 2 units, 4 passes
    const bool nvalid = (n0 + nb) < a.cout;
    const float4* wp = a.wt + (size_t)k4b * a.cout + (nvalid ? (n0 + nb) : 0);
    const size_t wstep = (size_t)2 * a.cout;     // float4 units between the passes of one K-tile
    const size_t wtile = (size_t)K4 * a.cout;    // ... and between K-tiles

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[4], rb[4];
    float rsc = 1.f;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* p0[4];
    const float* p1[4];

    auto setup_tap = [&](int tap) {
        int dy = 0, dx = 0;
        if (TAPS == 9) {
            const int ty = tap / 3;
            dy = ty - 1;
            dx = tap - ty * 3 - 1;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = py[i] + dy, xx = px[i] + dx;
            const bool ok = pv[i] && (unsigned)yy < (unsigned)a.h && (unsigned)xx < (unsigned)a.w;
            const size_t pix = (size_t)(pbase[i] + (yy >> a.up) * Ws + (xx >> a.up));
            p0[i] = ok ? a.src0 + pix * a.c0 : a.zeros;
            p1[i] = (ok && a.src1) ? a.src1 + pix * a.c1 : a.zeros;
        }
    };

    auto load_tile = [&](int cc) {
        const int ch = cc + k4a * 4;                         // first of this thread's 4 channels in the tap
        const int sel = ch < a.c0 ? 0 : (ch - a.c0 < a.c1 ? 1 : 2);
        const int off = sel == 0 ? ch : sel == 1 ? ch - a.c0 : k4a * 4;
        rsc = sel == 1 ? a.scale1 : a.scale0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* base = sel == 0 ? p0[i] : sel == 1 ? p1[i] : a.zeros;
            ra[i] = *reinterpret_cast<const float4*>(base + off);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = wp[i * wstep];
        wp += wtile;
    };

    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v = ra[i];
            if constexpr (PREC == VH_PREC_F32) {
                v.x *= rsc; v.y *= rsc; v.z *= rsc; v.w *= rsc;          // zero page * scale = 0; silu(0) = 0
                if (a.pro == VH_PRO_SILU) {
                    v.x = mp_silu_dev(v.x); v.y = mp_silu_dev(v.y);
                    v.z = mp_silu_dev(v.z); v.w = mp_silu_dev(v.w);
                }
            }
            sA[buf][k4a * BM + ((ma + 32 * i) ^ k4a)] = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v = rb[i];
            if (!nvalid) v = zero4;
            sB[buf][(k4b + 2 * i) * BN + nb] = v;
        }
    };

    const int wv = t >> 6, l = t & 63, lr = l & 31, hh = l >> 5;
    const int wm = wv >> 1, wn = wv & 1;

    auto compute = [&](int buf) {
        if constexpr (PREC == VH_PREC_F32) {
#pragma unroll
            for (int kg = 0; kg < BK / 8; ++kg) {
                const int k4 = kg * 2 + hh;
                float4 af[2], bf[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) af[mi] = sA[buf][k4 * BM + ((wm * 64 + mi * 32 + lr) ^ k4)];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) bf[ni] = sB[buf][k4 * BN + wn * 64 + ni * 32 + lr];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
                        const float av = j == 0 ? af[mi].x : j == 1 ? af[mi].y : j == 2 ? af[mi].z : af[mi].w;
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            const float bv = j == 0 ? bf[ni].x : j == 1 ? bf[ni].y : j == 2 ? bf[ni].z : bf[ni].w;
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mi][ni], 0, 0, 0);
                        }
                    }
                }
            }
        } else {
            // 32x32x16 bf16: lane (row = l&31, h = l>>5) supplies k = 8h..8h+7 of the slab = chunk 2*slab + h.
#pragma unroll
            for (int sl = 0; sl < BK / 16; ++sl) {
                const int uh = (sl * 2 + hh) * 2, ul = uh + 1;
                bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int m = wm * 64 + mi * 32 + lr;
                    ah[mi] = *reinterpret_cast<const bf16x8*>(&sA[buf][uh * BM + (m ^ uh)]);
                    al[mi] = *reinterpret_cast<const bf16x8*>(&sA[buf][ul * BM + (m ^ ul)]);
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int n = wn * 64 + ni * 32 + lr;
                    bh[ni] = *reinterpret_cast<const bf16x8*>(&sB[buf][uh * BN + n]);
                    bl[ni] = *reinterpret_cast<const bf16x8*>(&sB[buf][ul * BN + n]);
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                    }
            }
        }
    };

    // ---- main loop ----------------------------------------------------------
    const int KT = a.k_pad / BK;
    int tap = 0, cc = 0;
    setup_tap(0);
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) {
            cc += BK;
            if (cc >= a.cin_pad) {
                cc = 0;
                ++tap;
                setup_tap(tap);
            }
            load_tile(cc);
        }
        compute(buf);
        if (kt + 1 < KT) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue -----------------------------------------------------------
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            conv_epilogue_tile(a, acc[mi][ni], m0 + wm * 64 + mi * 32, n0 + wn * 64 + ni * 32 + lr, hh);
}

}  // namespace

int vh_conv_x3_glds_dispatch(vh_ctx* ctx, const vh_conv_args& a, vhconv::ConvK k, double flops, double bytes);
int vh_conv_patch_choice(const vh_conv_args& a, long long M, long long* pwgs);      // conv_x3.hip: 1 = the patch-resident kernel runs these arguments

extern "C" int vh_conv_takes_patch(const vh_conv_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_conv_takes_patch: null args");
    if (p->prec != VH_PREC_BF16X3 || p->kernel != VH_CONV_GLDS256 || p->rows <= 0 || p->h <= 0 || p->w <= 0 || p->cout <= 0) return 0;
    long long pwgs = 0;
    return vh_conv_patch_choice(*p, (long long)p->rows * p->h * p->w, &pwgs) == 1 ? 1 : 0;
}

extern "C" int vh_conv(vh_ctx* ctx, const vh_conv_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_conv: null args");
    const vh_conv_args a = *p;
    VH_REQUIRE(a.taps == 1 || a.taps == 9, "vh_conv: taps must be 1 or 9 (got %d)", a.taps);
    const bool has_sink = a.sink[0].ptr || a.sink[1].ptr;
    VH_REQUIRE(a.src0 && a.wt && (a.out || a.out_s8 || a.epi == VH_EPI_QKV || has_sink), "vh_conv: null tensor");
    for (int i = 0; i < 2; ++i)
        VH_REQUIRE(!a.sink[i].ptr || (a.sink[i].c_total % 32 == 0 && a.sink[i].c_off % 32 == 0 && a.sink[i].c_off >= 0 && a.sink[i].c_off + a.cout <= a.sink[i].c_total &&
                                      a.cout % 32 == 0 && vh_aligned16(a.sink[i].ptr)),
                   "vh_conv: sink %d: c_total / c_off must be multiples of 32 with c_off + cout <= c_total (cout %% 32 == 0), 16-byte aligned", i);
    VH_REQUIRE(a.prec == VH_PREC_F32 || a.prec == VH_PREC_BF16X3, "vh_conv: bad prec %d", a.prec);
    // bf16x3 + VH_CONV_GLDS256 + 3x3: a second S8 source is a 1-TAP TAIL SEGMENT of the K loop (see vh_conv_args.src1), not a channel concat
    const bool tail = a.prec == VH_PREC_BF16X3 && a.src1 != nullptr && !a.src_f32;
    if (a.src_f32) {
        VH_REQUIRE(a.prec == VH_PREC_BF16X3 && a.kernel == VH_CONV_GLDS256 && a.taps == 9 && a.epi != VH_EPI_QKV && !a.tail_f32,
                   "vh_conv: src_f32 belongs to a 3x3 VH_PREC_BF16X3 / VH_CONV_GLDS256 convolution without a tail segment");
        VH_REQUIRE(a.c0 % 32 == 0 && a.c1 % 32 == 0 && a.cin_pad == a.c0 + a.c1, "vh_conv: src_f32 needs c0, c1 multiples of 32 and cin_pad == c0 + c1 (got %d, %d, %d)", a.c0, a.c1, a.cin_pad);
    } else if (a.prec == VH_PREC_BF16X3) {
        VH_REQUIRE(a.pro == VH_PRO_NONE && a.scale0 == 1.0f, "vh_conv: bf16x3 takes pre-split (S8) sources; scale/silu/concat belong to their producer");
        VH_REQUIRE(a.c0 % 32 == 0 && a.cin_pad == a.c0, "vh_conv: bf16x3 needs c0 == cin_pad, a multiple of 32 (got %d, %d)", a.c0, a.cin_pad);
        VH_REQUIRE(!tail || (a.kernel == VH_CONV_GLDS256 && a.taps == 9 && !a.up && a.c1 > 0 && a.c1 % 32 == 0 && (a.scale1 == 1.0f || a.tail_f32) && a.epi != VH_EPI_QKV),
                   "vh_conv: a bf16x3 second source is the 1-tap tail of a 3x3 VH_CONV_GLDS256 convolution without `up` (c1 %% 32 == 0, scale1 == 1)");
    }
    VH_REQUIRE(!a.tail_f32 || (tail && (a.src2 ? (a.c2 > 0 && a.c2 % 32 == 0 && vh_aligned16(a.src2)) : a.c2 == 0)),
               "vh_conv: tail_f32 needs a bf16x3 tail segment: src1 (c1 %% 32 == 0) and optionally src2 (c2 %% 32 == 0), fp32 NHWC, 16-byte aligned");
    VH_REQUIRE(a.tail_f32 || (!a.src2 && a.c2 == 0), "vh_conv: src2 / c2 belong to tail_f32");
    const int tail_c = tail ? a.c1 + (a.tail_f32 ? a.c2 : 0) : 0;
    VH_REQUIRE(!a.out_s8 || (a.cout % 32 == 0 && a.out_s8_c == a.cout), "vh_conv: S8 output needs cout %% 32 == 0 and out_s8_c == cout");
    VH_REQUIRE(a.rows > 0 && a.h > 0 && a.w > 0 && a.cout > 0, "vh_conv: bad geometry");
    VH_REQUIRE(a.c0 > 0 && a.c0 % 4 == 0, "vh_conv: c0 must be a positive multiple of 4 (got %d)", a.c0);
    VH_REQUIRE(a.src1 ? (a.c1 > 0 && a.c1 % 4 == 0) : a.c1 == 0, "vh_conv: bad c1 %d", a.c1);
    VH_REQUIRE(a.cin_pad % BK == 0 && a.cin_pad >= a.c0 + (tail ? 0 : a.c1), "vh_conv: cin_pad %d must be a multiple of %d and >= c0+c1 = %d", a.cin_pad, BK, a.c0 + a.c1);
    VH_REQUIRE(a.k_pad == a.taps * a.cin_pad + tail_c, "vh_conv: k_pad %d != taps*cin_pad%s %d", a.k_pad, tail ? " + c1 (+ c2)" : "", a.taps * a.cin_pad + tail_c);
    VH_REQUIRE(a.zeros && vh_aligned16(a.zeros) && a.zeros_bytes >= (size_t)std::max(a.cin_pad, tail ? std::max(a.c1, a.c2) : 0) * 4 + 64, "vh_conv: zero page missing or smaller than max(cin_pad, c1)*4+64 bytes");
    VH_REQUIRE(vh_aligned16(a.src0) && vh_aligned16(a.src1) && vh_aligned16(a.wt), "vh_conv: source/weight pointers must be 16-byte aligned");
    // the epilogues read the residual / cvec rows and write the fp32 / S8 outputs as 16-byte vectors whenever cout % 4 == 0
    VH_REQUIRE(a.cout % 4 != 0 || (vh_aligned16(a.out) && vh_aligned16(a.out_s8) && vh_aligned16(a.res) && (a.cvec_ld % 4 != 0 || vh_aligned16(a.cvec))),
               "vh_conv: with cout %% 4 == 0, out, out_s8, res (and cvec when cvec_ld %% 4 == 0) must be 16-byte aligned");
    VH_REQUIRE(!a.up || (a.h % 2 == 0 && a.w % 2 == 0), "vh_conv: up needs even output size");
    VH_REQUIRE(a.pro == VH_PRO_NONE || a.pro == VH_PRO_SILU, "vh_conv: bad prologue");
    VH_REQUIRE(a.epi >= VH_EPI_STORE && a.epi <= VH_EPI_QKV, "vh_conv: bad epilogue");
    VH_REQUIRE(a.stagger >= 0 && a.stagger <= 2, "vh_conv: stagger must be 0, 1 or 2");
    VH_REQUIRE(a.korder >= VH_KORDER_AUTO && a.korder <= VH_KORDER_CHUNK, "vh_conv: korder must be VH_KORDER_AUTO, _TAP or _CHUNK");
    VH_REQUIRE((a.tile >= VH_TILE_AUTO && a.tile <= VH_TILE_256x64) || a.tile == VH_TILE_256x192 || a.tile == VH_TILE_PATCH16, "vh_conv: tile must be one of VH_TILE_*");
    VH_REQUIRE(a.tile != VH_TILE_256x192 || a.taps == 9, "vh_conv: VH_TILE_256x192 exists for 3x3 convolutions only");
    VH_REQUIRE(a.tile == VH_TILE_AUTO || a.kernel == VH_CONV_GLDS256, "vh_conv: a forced tile shape exists only for VH_CONV_GLDS256");
    if (a.epi == VH_EPI_QKV) {
        VH_REQUIRE(a.qkv && a.taps == 1 && a.kernel == VH_CONV_GLDS256 && !a.out && !a.out_s8, "vh_conv: QKV epilogue needs qkv args, a 1x1 GLDS convolution and no other output");
        const vh_qkv_epilogue& e = *a.qkv;
        VH_REQUIRE((e.nj == 2 || (e.nj == 3 && e.q)) && e.k && e.v, "vh_conv: QKV epilogue: nj must be 2 or 3 (3 needs q), k and v given");
        VH_REQUIRE(e.heads > 0 && (a.cout == e.heads * 64 * e.nj || a.cout == e.heads * 32 * e.nj), "vh_conv: QKV epilogue: cout %d != heads*D*nj with D = 64 or 32", a.cout);
        VH_REQUIRE((a.h * a.w) % 32 == 0 && e.koff % 16 == 0 && e.koff >= 0, "vh_conv: QKV epilogue needs h*w %% 32 == 0 and koff %% 16 == 0");
        VH_REQUIRE(e.rows_per_b > 0 && a.rows % e.rows_per_b == 0 && e.koff + e.rows_per_b * a.h * a.w <= e.kl, "vh_conv: QKV epilogue: keys do not fit");
        VH_REQUIRE(vh_aligned16(e.q) && vh_aligned16(e.k) && vh_aligned16(e.v), "vh_conv: QKV epilogue: pointers must be 16-byte aligned");
    }
    VH_REQUIRE(a.epi != VH_EPI_SCALE_SILU || (a.cvec && a.cvec_ld >= a.cout), "vh_conv: SCALE_SILU needs cvec with ld >= cout");
    VH_REQUIRE(a.epi != VH_EPI_MPSUM || a.res, "vh_conv: MPSUM needs res");
    VH_REQUIRE(!a.res_scale || (a.epi == VH_EPI_MPSUM && !a.res_up), "vh_conv: res_scale belongs to an MPSUM epilogue without res_up");
    VH_REQUIRE(!(a.epi == VH_EPI_MPSUM && a.res_up) || (a.h % 2 == 0 && a.w % 2 == 0), "vh_conv: res_up needs even output size");
    const long long M = (long long)a.rows * a.h * a.w;
    VH_REQUIRE(M < (1LL << 31) - BM, "vh_conv: too many pixels");
    const long long MT = (M + BM - 1) / BM, NT = (a.cout + BN - 1) / BN;
    VH_REQUIRE(MT * NT < (1LL << 31), "vh_conv: grid too large");

    ConvK k;
    k.src0 = a.src0; k.src1 = a.src1; k.zeros = a.zeros; k.c0 = a.c0; k.c1 = a.c1; k.scale0 = a.scale0; k.scale1 = a.scale1;
    k.h = a.h; k.w = a.w; k.up = a.up ? 1 : 0; k.pro = a.pro;
    k.wt = reinterpret_cast<const float4*>(a.wt); k.cin_pad = a.cin_pad; k.k_pad = a.k_pad; k.cout = a.cout;
    k.out = a.out; k.out_s8 = static_cast<unsigned short*>(a.out_s8); k.out_s8_c = a.out_s8_c; k.epi = a.epi; k.cvec = a.cvec; k.cvec_ld = a.cvec_ld; k.res = a.res; k.res_up = a.res_up ? 1 : 0; k.res_scale = a.res_scale;
    k.ta = a.ta; k.tb = a.tb; k.clip = a.clip;
    k.M = (int)M; k.HW = a.h * a.w; k.NT = (int)NT;
    k.div_hw = fastdiv_make((unsigned)(a.h * a.w)); k.div_w = fastdiv_make((unsigned)a.w); k.div_c0u = fastdiv_make((unsigned)(a.c0 / 4));
    k.ksplit = 1; k.scratch = nullptr; k.korder = 0; k.stagger = 0; k.dbg = nullptr;
    k.ptx = k.pty = 0; k.div_ptx = k.div_ptiles = fastdiv_make(1);
    k.src2 = a.src2; k.c2 = a.c2; k.scale2 = a.scale2; k.tail_f32 = a.tail_f32 ? 1 : 0; k.src_f32 = a.src_f32 ? 1 : 0;
    for (int i = 0; i < 2; ++i) {
        k.sk_ptr[i] = static_cast<unsigned short*>(a.sink[i].ptr); k.sk_ct[i] = a.sink[i].c_total; k.sk_off[i] = a.sink[i].c_off;
        k.sk_scale[i] = a.sink[i].scale; k.sk_silu[i] = a.sink[i].silu;
    }
    VH_REQUIRE(!has_sink || a.kernel == VH_CONV_GLDS256, "vh_conv: S8 sinks are written by the patch-resident kernel (VH_CONV_GLDS256) only");
    k.q = nullptr; k.qk = k.qv = nullptr; k.q_heads = k.q_nj = k.q_rows_per_b = k.q_koff = k.q_klp = 0; k.q_d = 64; k.q_scale = 1.f;
    if (a.epi == VH_EPI_QKV) {
        const vh_qkv_epilogue& e = *a.qkv;
        k.q = e.q; k.qk = static_cast<unsigned short*>(e.k); k.qv = static_cast<unsigned short*>(e.v);
        k.q_heads = e.heads; k.q_nj = e.nj; k.q_rows_per_b = e.rows_per_b; k.q_koff = e.koff; k.q_klp = (e.kl + 63) / 64 * 64; k.q_d = a.cout / (e.heads * e.nj); k.q_scale = e.qscale;
    }
    VH_REQUIRE(!a.scratch || vh_aligned16(a.scratch), "vh_conv: scratch must be 16-byte aligned");
    const int taps = a.taps, prec = a.prec;
    const unsigned grid = (unsigned)(MT * NT);
    // algorithmic work: 2*M*cout*cin*taps FLOPs; bytes = input + weights + output (+ residual), each once
    const double cin = tail ? (double)a.c0 : (double)a.c0 + a.c1;
    const double flops = 2.0 * (double)M * a.cout * (cin * a.taps + tail_c);
    const double in_px = a.up ? (double)M / 4 : (double)M;
    double bytes = 4.0 * (in_px * cin + cin * a.taps * a.cout + (double)M * a.cout + (tail ? (double)M * tail_c + (double)tail_c * a.cout : 0.0));
    if (a.epi == VH_EPI_MPSUM) bytes += 4.0 * (a.res_up ? (double)M / 4 : (double)M) * a.cout;
    VH_REQUIRE(a.kernel == VH_CONV_TILE128 || (a.kernel == VH_CONV_GLDS256 && a.prec == VH_PREC_BF16X3),
               "vh_conv: kernel %d unknown or not available for this precision", a.kernel);
    if (a.kernel == VH_CONV_GLDS256) return vh_conv_x3_glds_dispatch(ctx, a, k, flops, bytes);
    return vh_dispatch(ctx, taps == 9 ? VH_TAG_CONV3 : VH_TAG_CONV1, flops, bytes, [k, taps, prec, grid](hipStream_t s) -> int {
        if (prec == VH_PREC_BF16X3) {
            if (taps == 9) hipLaunchKernelGGL((conv_igemm<9, VH_PREC_BF16X3>), dim3(grid), dim3(256), 0, s, k);
            else hipLaunchKernelGGL((conv_igemm<1, VH_PREC_BF16X3>), dim3(grid), dim3(256), 0, s, k);
        } else {
            if (taps == 9) hipLaunchKernelGGL((conv_igemm<9, VH_PREC_F32>), dim3(grid), dim3(256), 0, s, k);
            else hipLaunchKernelGGL((conv_igemm<1, VH_PREC_F32>), dim3(grid), dim3(256), 0, s, k);
        }
        return vh_check_launch("conv_igemm");
    });
}
