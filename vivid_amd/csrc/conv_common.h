// Shared pieces of the convolution kernels (conv_igemm.hip: 128x128 register-staged tile, fp32 and bf16x3;
// conv_x3.hip: 256-wide direct-to-LDS tile, bf16x3).
#pragma once
#include "ctx.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace vhconv {

// Division of a 31-bit value by a launch-invariant divisor without the ~40-instruction integer-division sequence:
// q = umulhi(n, mul) >> shr with mul = floor(2^(31+l)/d) + 1, shr = l - 1, l = ceil(log2 d); exact for 0 <= n < 2^31
// (error of n*mul/2^(31+l) against n/d is < 2^-l <= 1/d).  mul == 0 encodes d == 1.
struct FastDiv { unsigned mul, shr; };
inline FastDiv fastdiv_make(unsigned d) {
    if (d <= 1) return FastDiv{0u, 0u};
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    return FastDiv{(unsigned)(((1ull << (31 + l)) / d) + 1), l - 1};
}
__device__ __forceinline__ int fastdiv(int n, const FastDiv f) { return f.mul ? (int)(__umulhi((unsigned)n, f.mul) >> f.shr) : n; }

struct ConvK {
    const float* src0; const float* src1; const float* zeros;
    int c0, c1; float scale0, scale1;
    int h, w, up, pro;
    const float4* wt; int cin_pad, k_pad, cout;
    float* out; unsigned short* out_s8; int out_s8_c; int epi;
    const float* cvec; int cvec_ld;
    const float* res; int res_up;
    const float* res_scale;         // optional per-pixel factor of the residual (MPSUM): res[m][o] * res_scale[m]
    float ta, tb, clip;
    int M, HW, NT;
    FastDiv div_hw, div_w;          // n / HW and n / w
    FastDiv div_c0u;                // n / (c0 / 4): a 16-byte-unit offset into src0 back to (pixel, unit) - the tail segment of conv_x3_glds
    // VH_EPI_QKV: attention operand buffers and the key-sequence placement of vh_qkv_split_x3
    float* q; unsigned short* qk; unsigned short* qv; int q_heads, q_nj, q_rows_per_b, q_koff, q_klp, q_d; float q_scale;
    int stagger;                    // conv_x3_glds: waves 4-7 issue their DMA in the middle of their MFMAs instead of before them
    int korder;                     // conv_x3_glds, 9 taps: 0 = tap-major K order, 1 = channel-chunk-major (see the kernel)
    unsigned long long* dbg;        // VH_CLOCK builds: [workgroup][2] = shader cycles, 100 MHz ticks of the K loop
    // conv_x3_patch: extra S8 outputs (vh_s8_sink) - the result scaled, optionally through mp_silu, into channels [off, off + cout) of rows of ct channels
    unsigned short* sk_ptr[2]; int sk_ct[2], sk_off[2]; float sk_scale[2]; int sk_silu[2];
    int src_f32;                    // conv_x3_patch: src0 / src1 are fp32 NHWC sources of the MAIN loop (channel concat, scale0 / scale1, pro), vh_conv_args.src_f32
    const float* src2; int c2; float scale2; int tail_f32;   // conv_x3_patch: fp32 tail sources (vh_conv_args.tail_f32): src1 / src2 fp32 NHWC, c1 / c2 channels, scale1 / scale2
    int ptx, pty; FastDiv div_ptx, div_ptiles;   // conv_x3_patch: 16x16-pixel tiles per image row / column, n / ptx, n / (ptx*pty)
    int ksplit; float* scratch;     // split-K: this launch covers K-tiles [ks*KT/ksplit, (ks+1)*KT/ksplit) and
                                    // writes raw partial sums to scratch[ks][M][cout]; a reducer applies the epilogue
};

__device__ __forceinline__ float mp_silu_dev(float v) {
    // silu(v)/0.596 = v / (1 + exp(-v)) / 0.596
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * v);
    return v * __builtin_amdgcn_rcpf(1.0f + e) * (1.0f / 0.596f);
}

__device__ __forceinline__ unsigned bf16_rn_bits(float v) {      // v_cvt_pk_bf16_f32: round to nearest even
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v);
}

// x ~= hi + lo, both bf16 (round to nearest even); written into the S8 layout.
__device__ __forceinline__ void store_s8(unsigned short* base, size_t pix, int cpad, int ch, float v) {
    const unsigned hi = bf16_rn_bits(v);
    const unsigned lo = bf16_rn_bits(v - __uint_as_float(hi << 16));
    unsigned short* p = base + (pix * cpad + (size_t)(ch & ~7)) * 2 + (ch & 7);
    p[0] = (unsigned short)hi;
    p[8] = (unsigned short)lo;
}


// Epilogue of ONE 32x32 accumulator tile (passed by value so the kernel's accumulator array never has its
// address taken - a by-reference array of f32x16 ends up in scratch).  C/D map of the 32x32 MFMA:
// column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  row0: first pixel row of the tile; gn: this lane's column.
__device__ __forceinline__ void conv_epilogue_tile(const ConvK& a, const f32x16 accv, int row0, int gn, int hh) {
    if (gn >= a.cout) return;
    const int Hr = a.h >> 1, Wr = a.w >> 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gm = row0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (gm >= a.M) continue;
        float y = accv[r];
        if (a.epi == VH_EPI_SCALE_SILU) {
            const int img = fastdiv(gm, a.div_hw);
            y = mp_silu_dev(y * a.cvec[(size_t)img * a.cvec_ld + gn]);
        } else if (a.epi == VH_EPI_MPSUM) {
            size_t rrow = (size_t)gm;
            if (a.res_up) {
                const int img = fastdiv(gm, a.div_hw);
                const int rem = gm - img * a.HW;
                const int yy = fastdiv(rem, a.div_w), xx = rem - yy * a.w;
                rrow = (size_t)((img * Hr + (yy >> 1)) * Wr + (xx >> 1));
            }
            y = a.res[rrow * a.cout + gn] * (a.res_scale ? a.ta * a.res_scale[rrow] : a.ta) + y * a.tb;
            if (a.clip > 0.f) y = fminf(fmaxf(y, -a.clip), a.clip);
        } else if (a.epi == VH_EPI_STORE && a.clip > 0.f) {
            y = fminf(fmaxf(y, -a.clip), a.clip);
        }
        if (a.out) a.out[(size_t)gm * a.cout + gn] = y;
        if (a.out_s8) store_s8(a.out_s8, (size_t)gm, a.out_s8_c, gn, y);
    }
}

// Epilogue of one 32x32 accumulator tile, transposed through a per-wave LDS patch so that every lane handles
// 4 consecutive output channels of one pixel: residual / cvec loads and the fp32 / S8 stores are 16-byte (8-byte for
// the bf16 halves) accesses, 8 lanes per 128-byte line, instead of 4-byte accesses - a quarter of the memory
// instructions (the accumulator-layout epilogue is store-issue bound).  `patch`: 32 x 36 floats owned by this wave.
struct EpiAux;
__device__ __forceinline__ void conv_epilogue_patch(const ConvK& a, int row0, int col0, const float* patch, int lane, const EpiAux* aux);

__device__ __forceinline__ void conv_epilogue_tile_lds(const ConvK& a, const f32x16 accv, int row0, int col0,
                                                        float* patch, int lane) {
    constexpr int LD = 36;
    const int lr = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * hh) * LD + lr] = accv[r];
    // (same wave wrote and reads: the compiler orders the ds_read behind the ds_writes with lgkmcnt)
    conv_epilogue_patch(a, row0, col0, patch, lane, nullptr);
}

// Same for a 32x32 block held as 2x2 accumulator tiles of v_mfma_f32_16x16x32 (C/D map: column = lane&15,
// row = 4*(lane>>4) + reg).
__device__ __forceinline__ void conv_epilogue_tiles16_lds(const ConvK& a, const f32x4 t00, const f32x4 t01, const f32x4 t10,
                                                           const f32x4 t11, int row0, int col0, float* patch, int lane,
                                                           const EpiAux* aux = nullptr) {
    constexpr int LD = 36;
    const int c = lane & 15, rb = (lane >> 4) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        patch[(rb + r) * LD + c] = t00[r];
        patch[(rb + r) * LD + 16 + c] = t01[r];
        patch[(16 + rb + r) * LD + c] = t10[r];
        patch[(16 + rb + r) * LD + 16 + c] = t11[r];
    }
    conv_epilogue_patch(a, row0, col0, patch, lane, aux);
}

// Read-out of a 32 x 36-float patch: every lane handles 4 consecutive output channels of one pixel.
// Epilogue of 4 consecutive output channels gn..gn+3 of pixel gm (y = raw sums): shared by the LDS patch read-out
// and by the split-K reducer.
// `aux`: the residual (MPSUM) or cvec (SCALE_SILU) values of these 4 channels when the caller fetched them ahead
// (conv_epilogue_prefetch), else nullptr.
__device__ __forceinline__ void conv_epilogue_vec4(const ConvK& a, int gm, int gn, float (&y)[4], const float4* aux = nullptr) {
    if (gm >= a.M || gn >= a.cout) return;
    const bool full = gn + 3 < a.cout;
    if (aux && a.epi == VH_EPI_SCALE_SILU) {
        const float c[4] = {aux->x, aux->y, aux->z, aux->w};
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = mp_silu_dev(y[j] * c[j]);
    } else if (aux && a.epi == VH_EPI_MPSUM) {
        const float rv[4] = {aux->x, aux->y, aux->z, aux->w};
        const float ta = a.res_scale ? a.ta * a.res_scale[gm] : a.ta;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            y[j] = rv[j] * ta + y[j] * a.tb;
            if (a.clip > 0.f) y[j] = fminf(fmaxf(y[j], -a.clip), a.clip);
        }
    } else if (a.epi == VH_EPI_SCALE_SILU) {
        const int img = fastdiv(gm, a.div_hw);
        const float* cp = a.cvec + (size_t)img * a.cvec_ld + gn;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (full || gn + j < a.cout) y[j] = mp_silu_dev(y[j] * cp[j]);
    } else if (a.epi == VH_EPI_STORE) {
        if (a.clip > 0.f) {
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = fminf(fmaxf(y[j], -a.clip), a.clip);
        }
    } else if (a.epi == VH_EPI_MPSUM) {
        size_t rrow = (size_t)gm;
        if (a.res_up) {
            const int Hr = a.h >> 1, Wr = a.w >> 1;
            const int img = fastdiv(gm, a.div_hw);
            const int rem = gm - img * a.HW;
            const int yy = fastdiv(rem, a.div_w), xx = rem - yy * a.w;
            rrow = (size_t)((img * Hr + (yy >> 1)) * Wr + (xx >> 1));
        }
        const float* rp = a.res + rrow * a.cout + gn;
        float rv[4];
        if (full && (a.cout & 3) == 0) {
            const float4 t = *reinterpret_cast<const float4*>(rp);
            rv[0] = t.x; rv[1] = t.y; rv[2] = t.z; rv[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) rv[j] = (gn + j < a.cout) ? rp[j] : 0.f;
        }
        const float ta = a.res_scale ? a.ta * a.res_scale[rrow] : a.ta;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            y[j] = rv[j] * ta + y[j] * a.tb;
            if (a.clip > 0.f) y[j] = fminf(fmaxf(y[j], -a.clip), a.clip);
        }
    }
    if (a.out) {
        float* op = a.out + (size_t)gm * a.cout + gn;
        if (full && (a.cout & 3) == 0) {
            *reinterpret_cast<float4*>(op) = make_float4(y[0], y[1], y[2], y[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (gn + j < a.cout) op[j] = y[j];
        }
    }
    if (a.out_s8) {       // cout % 32 == 0 here: the 4 channels are half of one 8-channel chunk
        unsigned h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = bf16_rn_bits(y[j]);
            l[j] = bf16_rn_bits(y[j] - __uint_as_float(h[j] << 16));
        }
        unsigned short* q = a.out_s8 + ((size_t)gm * a.out_s8_c + (size_t)(gn & ~7)) * 2 + (gn & 7);
        *reinterpret_cast<uint2*>(q) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
        *reinterpret_cast<uint2*>(q + 8) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
    }
}

// The residual / cvec values the read-out of block (row0, col0) will need, fetched AHEAD of it: the epilogue walks its 32x32
// blocks one after the other, each ending in stores the compiler may not move the next block's loads across, so without
// this every block waits out one full global-load latency.  Valid only when `ok` (aligned, whole vectors in range).
struct EpiAux { float4 v[4]; float rs[4]; bool ok; };      // rs: ta * res_scale of the four rows (ta when there is no res_scale)
__device__ __forceinline__ EpiAux conv_epilogue_prefetch(const ConvK& a, int row0, int col0, int lane) {
    EpiAux x;
    const int cg = lane & 7, rsub = lane >> 3;
    const int gn = col0 + 4 * cg;
    x.ok = (a.epi == VH_EPI_MPSUM || a.epi == VH_EPI_SCALE_SILU) && (a.cout & 3) == 0 && (a.epi != VH_EPI_SCALE_SILU || (a.cvec_ld & 3) == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        x.rs[i] = a.ta;
        const int gm = row0 + rsub + 8 * i;
        if (!x.ok || gm >= a.M || gn + 3 >= a.cout) continue;
        if (a.epi == VH_EPI_MPSUM) {
            size_t rrow = (size_t)gm;
            if (a.res_up) {
                const int Hr = a.h >> 1, Wr = a.w >> 1;
                const int img = fastdiv(gm, a.div_hw);
                const int rem = gm - img * a.HW;
                const int yy = fastdiv(rem, a.div_w), xx = rem - yy * a.w;
                rrow = (size_t)((img * Hr + (yy >> 1)) * Wr + (xx >> 1));
            }
            x.v[i] = *reinterpret_cast<const float4*>(a.res + rrow * a.cout + gn);
        } else {
            x.v[i] = *reinterpret_cast<const float4*>(a.cvec + (size_t)fastdiv(gm, a.div_hw) * a.cvec_ld + gn);
        }
    }
    return x;
}

__device__ __forceinline__ void conv_epilogue_patch(const ConvK& a, int row0, int col0, const float* patch, int lane,
                                                     const EpiAux* aux = nullptr) {
    constexpr int LD = 36;
    const int cg = lane & 7, rsub = lane >> 3;
    const int gn = col0 + 4 * cg;
    const bool use_aux = aux && aux->ok && gn + 3 < a.cout;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rl = rsub + 8 * i;
        const float4 v = *reinterpret_cast<const float4*>(&patch[rl * LD + 4 * cg]);
        float y[4] = {v.x, v.y, v.z, v.w};
        conv_epilogue_vec4(a, row0 + rl, gn, y, use_aux ? &aux->v[i] : nullptr);
    }
}

// S8 store of 4 consecutive channels that are one half of an 8-channel chunk ([hi x8 | lo x8], 32 bytes): the lane pair (even, odd)
// that holds the two halves swaps one 8-byte piece so that the even lane writes the whole hi half and the odd lane the whole lo
// half - ONE 16-byte store per lane instead of two 8-byte ones (8-byte stores run at 0.5-0.7x the 16-byte rate, and the epilogue
// is paced by its store instructions).  `chunk`: address of the chunk's first bf16; all 64 lanes must take part.
__device__ __forceinline__ void s8_store_half_chunk(unsigned short* chunk, bool odd, const float (&y)[4]) {
    unsigned h[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = bf16_rn_bits(y[j]);
        lo[j] = bf16_rn_bits(y[j] - __uint_as_float(h[j] << 16));
    }
    const unsigned H0 = h[0] | (h[1] << 16), H1 = h[2] | (h[3] << 16), L0 = lo[0] | (lo[1] << 16), L1 = lo[2] | (lo[3] << 16);
    const unsigned r0 = __shfl_xor(odd ? H0 : L0, 1), r1 = __shfl_xor(odd ? H1 : L1, 1);      // even lane receives the partner's hi, odd its lo
    *reinterpret_cast<uint4*>(chunk + (odd ? 8 : 0)) = odd ? make_uint4(r0, r1, L0, L1) : make_uint4(H0, H1, r0, r1);
}

// Fast form of the prefetch + read-out pair for a 32x32 block that lies wholly inside the output (row0+32 <= M, col0+32 <= cout,
// cout % 4 == 0), with the epilogue kind a compile-time constant: no per-element bounds tests, alignment tests or epilogue
// dispatch, one 64-bit address per lane and block.  The generic conv_epilogue_patch / _vec4 walks ~50 scalar branches per
// 4-channel group; s_memtime stamps put it at ~4000 cycles per block, 15 us of a 95 us tile at 256x256.
template <int EPI>
__device__ __forceinline__ EpiAux conv_epilogue_prefetch_fast(const ConvK& a, int row0, int col0, int lane) {
    EpiAux x;
    x.ok = true;
    const int cg = lane & 7, rsub = lane >> 3;
    const int gn = col0 + 4 * cg;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = row0 + rsub + 8 * i;
        if (EPI == VH_EPI_MPSUM) {
            size_t rrow = (size_t)gm;
            if (a.res_up) {
                const int Hr = a.h >> 1, Wr = a.w >> 1;
                const int img = fastdiv(gm, a.div_hw);
                const int rem = gm - img * a.HW;
                const int yy = fastdiv(rem, a.div_w), xx = rem - yy * a.w;
                rrow = (size_t)((img * Hr + (yy >> 1)) * Wr + (xx >> 1));
            }
            x.v[i] = *reinterpret_cast<const float4*>(a.res + rrow * a.cout + gn);
            x.rs[i] = a.res_scale ? a.ta * a.res_scale[rrow] : a.ta;
        } else if (EPI == VH_EPI_SCALE_SILU) {
            x.v[i] = *reinterpret_cast<const float4*>(a.cvec + (size_t)fastdiv(gm, a.div_hw) * a.cvec_ld + gn);
            x.rs[i] = 0.f;
        } else {
            x.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            x.rs[i] = 0.f;
        }
    }
    return x;
}

template <int EPI>
__device__ __forceinline__ void conv_epilogue_block_fast(const ConvK& a, const f32x4 t00, const f32x4 t01, const f32x4 t10, const f32x4 t11,
                                                          int row0, int col0, float* patch, int lane, const EpiAux& aux) {
    constexpr int LD = 36;
    {
        const int c = lane & 15, rb = (lane >> 4) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            patch[(rb + r) * LD + c] = t00[r];
            patch[(rb + r) * LD + 16 + c] = t01[r];
            patch[(16 + rb + r) * LD + c] = t10[r];
            patch[(16 + rb + r) * LD + 16 + c] = t11[r];
        }
    }
    const int cg = lane & 7, rsub = lane >> 3;
    const size_t e0 = (size_t)(row0 + rsub) * a.cout + col0 + 4 * cg;     // this lane's first output element; rows follow at 8*cout
    const int sub = 4 * (cg & 1);                                          // position of its 4 channels inside their 8-channel S8 chunk
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(&patch[(rsub + 8 * i) * LD + 4 * cg]);
        float y[4] = {v.x, v.y, v.z, v.w};
        if (EPI == VH_EPI_SCALE_SILU) {
            const float c[4] = {aux.v[i].x, aux.v[i].y, aux.v[i].z, aux.v[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = mp_silu_dev(y[j] * c[j]);
        } else if (EPI == VH_EPI_MPSUM) {
            const float rv[4] = {aux.v[i].x, aux.v[i].y, aux.v[i].z, aux.v[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                y[j] = rv[j] * aux.rs[i] + y[j] * a.tb;
                if (a.clip > 0.f) y[j] = fminf(fmaxf(y[j], -a.clip), a.clip);
            }
        } else if (a.clip > 0.f) {                                   // VH_EPI_STORE with a clip: the fused conv_res1 + conv_skip of a decoder block
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = fminf(fmaxf(y[j], -a.clip), a.clip);
        }
        const size_t eo = e0 + (size_t)(8 * i) * a.cout;
        if (a.out) *reinterpret_cast<float4*>(a.out + eo) = make_float4(y[0], y[1], y[2], y[3]);
        if (a.out_s8) s8_store_half_chunk(a.out_s8 + (eo - sub) * 2, sub != 0, y);
    }
}

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so
// workgroup ids b, b+8, b+16, ... share an L2.  Give each of those 8 groups one contiguous run of tiles
// (bijective for any grid size): adjacent pixel tiles - which re-read each other's halo rows - and the
// N-tiles of one pixel tile then hit the same L2.
__device__ __forceinline__ unsigned xcd_tile_id() {
    const unsigned G = gridDim.x, bid = blockIdx.x;
    const unsigned q = G >> 3, r = G & 7u, xcd = bid & 7u, i = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

}  // namespace vhconv
