// Flash-style attention on bf16 MFMA with fp32 emulated by a hi/lo split ("bf16x3"), gfx950.
//
// Same algorithm and call sites as attention.hip (F.scaled_dot_product_attention,
// reference training/models.py:198,:305); every fp32 product a*b is evaluated as
// a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
// (x = hi + lo, hi = bf16(x), lo = bf16(x - hi)); softmax statistics stay in fp32.
//
// Operand formats (written by qkv_split_x3_k below, so the hot loop only copies 16-byte units):
//   Q   fp32 [bh][S][D], pre-scaled by log2(e)/sqrt(D); split to bf16 hi/lo in registers once.
//   K   "S8": [bh][KLp][D/8][hi x8 | lo x8] bf16 (a key row is D*4 bytes).
//   V^T [bh][D][hl][KLp] bf16, key positions permuted inside every group of 16
//       (pos = key with bits 2 and 3 swapped) so that the 8 keys one lane half needs for a
//       16-key MFMA step — the S^T accumulator row map — are one contiguous 16-byte unit.
// Kernels: attn_fwd_x3_m16 (64-channel heads, bounded logits, KL > 64: the benchmark's attention; hand-placed three-stage step on
// the 16x16x32 MFMA), attn_fwd_bf16x3_pipe (32-channel heads and running-maximum cases; its <64, true> instantiation is the 32x32x16
// form of the placed step, knob attn_m16 = 0), attn_fwd_bf16x3 (short sequences).
// One workgroup = NW waves x 32 queries of one (batch, head); K/V stream through LDS in 64-key
// tiles, double-buffered, one barrier per tile.  S^T = K Q^T puts the query on the lane, P stays
// in registers (accumulators -> bf16 pairs -> B operand of O^T += V^T P^T), as in attention.hip.
// The running max is only raised (and O rescaled) when some query's max grew by more than 2^8
// (fp32 accumulators: no precision is lost by the deferred scale).
#include "ctx.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int KT = 64;
constexpr float RESCALE_THR = 8.0f;

#ifndef VH_ATTN_PLACED
#define VH_ATTN_PLACED 1           // hand-placed three-stage step for D = 64 without running maximum (0: compiler-scheduled two-stage step, the A/B reference)
#endif

struct AttnXK {
    const float* q; const uint4* k; const uint4* vt; float* out;
    int heads, s, kl, klp, c;
    float n_zero;
    int out_s8;
    int nx, ny;                 // query tiles per (batch, head); number of (batch, head) pairs; the grid is 1-D with nx*ny workgroups
    int xcd;                    // 1: XCD-local (batch, head) placement
    unsigned long long* dbg;    // VH_CLOCK builds: [workgroup][2] = shader cycles, 100 MHz ticks of the key loop
};

// Workgroup -> (query tile, batch*head).  Workgroup ids are dealt round-robin over the 8 XCDs (each with its own 4 MB L2), so with
// the plain order the query tiles of one (batch, head) are spread over all 8 and each XCD streams that head's K/V from beyond its
// L2 for its share of the tiles (measured: 4.9x the algorithmic bytes cross the fabric).  Here every (batch, head) belongs to
// one XCD (its workgroups are ids xcd, xcd+8, ...): its query tiles run side by side on that XCD's 32 CUs and walk the same K/V
// tiles at about the same time.  Placement is a speed matter only; any mapping is correct.
__device__ __forceinline__ void attn_coords(const AttnXK& a, int& bx, int& by) {
    const unsigned bid = blockIdx.x;
    if (a.xcd && (a.ny & 7) == 0) {
        const unsigned xcd = bid & 7u, i = bid >> 3;
        by = (int)(xcd + 8u * (i / (unsigned)a.nx));
        bx = (int)(i % (unsigned)a.nx);
    } else {
        bx = (int)(bid % (unsigned)a.nx);
        by = (int)(bid / (unsigned)a.nx);
    }
}

__device__ __forceinline__ unsigned bf16_rn_bits(float v) {
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v);
}

// 8 floats -> bf16x8 hi and lo fragments (plain casts: hipcc emits v_cvt_pk_bf16_f32, round to nearest even)
__device__ __forceinline__ void split8(const float* v, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = (__bf16)v[j];
        lo[j] = (__bf16)(v[j] - (float)hi[j]);
    }
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma_abl16(bf16x8 a, bf16x8 b, f32x16 c) {
    f32x4v q0 = {c[0], c[1], c[2], c[3]}, q1 = {c[4], c[5], c[6], c[7]};
    q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, q0, 0, 0, 0);
    q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, q1, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { c[i] = q0[i]; c[4 + i] = q1[i]; }
    return c;
}

template <int D, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_bf16x3(const AttnXK a) {
    constexpr int NT = NW * 64;
    constexpr int KU = D / 4;                 // 16-byte units per key row (hi+lo)       : 16 for D=64
    constexpr int K_UNITS = KU * KT;          // units per K tile                        : 1024
    constexpr int V_UNITS = D * 2 * (KT / 8); // units per V^T tile (D rows x {hi,lo} x 8): 1024
    __shared__ uint4 sK[2][K_UNITS];          // [unit u = 2*chunk+hl][key ^ (u&7)]
    __shared__ uint4 sV[2][V_UNITS];          // [d][slot ^ (d&15)], slot = hl*8 + key/8

    const int t = threadIdx.x;
    const int wv = t >> 6, l = t & 63, lr = l & 31, hh = l >> 5;
    int bx, bh;
    attn_coords(a, bx, bh);
    const int b = bh / a.heads, hd = bh - b * a.heads;
    const int q0 = bx * (NW * 32) + wv * 32;
    const int qrow = q0 + lr;

    // ---- Q fragments -------------------------------------------------------------------------
    bf16x8 qh[D / 16], ql[D / 16];
    {
        const float* Qb = a.q + ((size_t)bh * a.s + (qrow < a.s ? qrow : 0)) * D;
#pragma unroll
        for (int sl = 0; sl < D / 16; ++sl) {
            float v[8];
            const float4 p0 = *reinterpret_cast<const float4*>(Qb + sl * 16 + hh * 8);
            const float4 p1 = *reinterpret_cast<const float4*>(Qb + sl * 16 + hh * 8 + 4);
            v[0] = p0.x; v[1] = p0.y; v[2] = p0.z; v[3] = p0.w; v[4] = p1.x; v[5] = p1.y; v[6] = p1.z; v[7] = p1.w;
            if (qrow >= a.s) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.f;
            }
            split8(v, qh[sl], ql[sl]);
        }
    }

    f32x16 oacc[D / 32];
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    // Running max starts at 0 — exact for the zero-logit phantom keys; without them the first 32-key
    // sub-tile replaces it by its own max before anything is exponentiated (see `first` below).
    float mrow = 0.f;
    float lsum = a.n_zero;

    // ---- staging maps (all per-thread addresses are loop-invariant; a tile only advances the pointers) ----
    constexpr int KPT = K_UNITS / NT, VPT = V_UNITS / NT;   // units per thread per tile
    const uint4* Kb = a.k + (size_t)bh * a.klp * KU;
    const uint4* Vb = a.vt + (size_t)bh * D * 2 * (a.klp / 8);
    const uint4* kp[KPT];
    const uint4* vp[VPT];
    int ksl[KPT], vsl[VPT], kkey[KPT], vku[VPT];             // LDS slots; key / key-unit of each staged unit
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const int idx = t + i * NT;
        const int key = idx / KU, u = idx - key * KU;
        kp[i] = Kb + (size_t)key * KU + u;
        ksl[i] = u * KT + (key ^ (u & 7));
        kkey[i] = key;
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int idx = t + i * NT;
        const int row = idx >> 3, ku = idx & 7;               // row = d*2 + hl
        const int d = row >> 1, hl = row & 1;
        vp[i] = Vb + (size_t)row * (a.klp / 8) + ku;
        vsl[i] = d * 16 + ((hl * 8 + ku) ^ (d & 15));
        vku[i] = ku;
    }
    uint4 rk[KPT], rv[VPT];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    auto load_tile = [&]() {                                   // klp is a multiple of KT: always in range
#pragma unroll
        for (int i = 0; i < KPT; ++i) { rk[i] = *kp[i]; kp[i] += KT * KU; }
#pragma unroll
        for (int i = 0; i < VPT; ++i) { rv[i] = *vp[i]; vp[i] += KT / 8; }
    };
    auto store_tile = [&](int buf, int k0, bool tail) {
        if (!tail) {                                             // full tiles carry no mask arithmetic
#pragma unroll
            for (int i = 0; i < KPT; ++i) sK[buf][ksl[i]] = rk[i];
#pragma unroll
            for (int i = 0; i < VPT; ++i) sV[buf][vsl[i]] = rv[i];
            return;
        }
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            uint4 v = rk[i];
            if (k0 + kkey[i] >= a.kl) v = zero4;
            sK[buf][ksl[i]] = v;
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            uint4 v = rv[i];
            if (k0 + (vku[i] >> 1) * 16 + 16 > a.kl) {           // the unit's 16-key group reaches past the end
                unsigned short* e = reinterpret_cast<unsigned short*>(&v);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int pos = vku[i] * 8 + j;               // position -> key: bits 2 and 3 swapped
                    const int key = (pos & ~12) | ((pos & 4) << 1) | ((pos & 8) >> 1);
                    if (k0 + key >= a.kl) e[j] = 0;
                }
            }
            sV[buf][vsl[i]] = v;
        }
    };

    // One 64-key tile.  TAIL (last tile, kl % 64 != 0) masks keys >= kl; full tiles carry no mask code.
    auto tile_body = [&](int buf, int k0, auto tailc) {
        constexpr bool TAIL = decltype(tailc)::value;
        const bool first_tile = (k0 == 0) && !(a.n_zero > 0.f);
#pragma unroll
        for (int ks = 0; ks < KT / 32; ++ks) {
            // ---- S^T - m = K Q^T - m : the running max enters as the initial accumulator -------------
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = -mrow;
            const int key = ks * 32 + lr;
#pragma unroll
            for (int sl = 0; sl < D / 16; ++sl) {
                const int uh = (sl * 2 + hh) * 2, ul = uh + 1;
                const bf16x8 kh = *reinterpret_cast<const bf16x8*>(&sK[buf][uh * KT + (key ^ (uh & 7))]);
                const bf16x8 kl_ = *reinterpret_cast<const bf16x8*>(&sK[buf][ul * KT + (key ^ (ul & 7))]);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl_, qh[sl], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[sl], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[sl], sacc, 0, 0, 0);
            }
            if constexpr (TAIL) {
                const int kbase = k0 + ks * 32 + 4 * hh;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + (r & 3) + 8 * (r >> 2) >= a.kl) sacc[r] = -INFINITY;
            }
            // ---- online softmax; sacc holds s - mrow ---------------------------------------------------
            float mx = sacc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            if (first_tile && ks == 0) {                    // no max yet: adopt this sub-tile's (O and l are still 0)
                mrow = mx;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] -= mx;
            } else if (__any(mx > RESCALE_THR)) {           // wave-uniform: raise the running max for every query
                const float dm = fmaxf(mx, 0.f);            // new max - old max
                const float alpha = __builtin_amdgcn_exp2f(-dm);
                lsum *= alpha;
                mrow += dm;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] -= dm;
#pragma unroll
                for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
            }
            float pv[16];
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pv[r] = __builtin_amdgcn_exp2f(sacc[r]);
                rs += pv[r];
            }
            rs += __shfl_xor(rs, 32);
            lsum += rs;
            // ---- O^T += V^T P^T : k-step s2 uses accumulator registers 8*s2 .. 8*s2+7 ------------------
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 ph, pl;
                split8(&pv[8 * s2], ph, pl);
                const int kuh = ks * 4 + s2 * 2 + hh;         // unit (8 permuted keys) inside the 64-key row
#pragma unroll
                for (int dt = 0; dt < D / 32; ++dt) {
                    const int d = dt * 32 + lr;
                    const bf16x8 vh = *reinterpret_cast<const bf16x8*>(&sV[buf][d * 16 + (kuh ^ (d & 15))]);
                    const bf16x8 vl = *reinterpret_cast<const bf16x8*>(&sV[buf][d * 16 + ((8 + kuh) ^ (d & 15))]);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, oacc[dt], 0, 0, 0);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, oacc[dt], 0, 0, 0);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, oacc[dt], 0, 0, 0);
                }
            }
        }
    };

    const int ntiles = (a.kl + KT - 1) / KT;
    const bool ragged = (a.kl % KT) != 0;
    load_tile();
    store_tile(0, 0, ragged && ntiles == 1);
    __syncthreads();
    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        const int k0 = tile * KT;
        const bool last = tile + 1 == ntiles;
        if (!last) load_tile();
        if (last && ragged) tile_body(buf, k0, std::true_type{});
        else tile_body(buf, k0, std::false_type{});
        if (!last) store_tile(buf ^ 1, k0 + KT, ragged && tile + 2 == ntiles);
        __syncthreads();
    }

    if (qrow < a.s && a.out_s8) {
        // S8 row of this query: channel hd*D + d; d = dt*32 + 8g + 4hh + (0..3) -> 8-byte pieces of a chunk's hi and lo halves
        const float inv = 1.0f / lsum;
        unsigned short* op = reinterpret_cast<unsigned short*>(a.out) + (((size_t)b * a.s + qrow) * a.c + hd * D) * 2;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned h[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = oacc[dt][4 * g + j] * inv;
                    h[j] = bf16_rn_bits(v);
                    l[j] = bf16_rn_bits(v - __uint_as_float(h[j] << 16));
                }
                unsigned short* q8 = op + (dt * 32 + 8 * g) * 2 + 4 * hh;
                *reinterpret_cast<uint2*>(q8) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
                *reinterpret_cast<uint2*>(q8 + 8) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
            }
    } else if (qrow < a.s) {
        const float inv = 1.0f / lsum;
        float* op = a.out + ((size_t)b * a.s + qrow) * a.c + hd * D;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 o;
                o.x = oacc[dt][4 * g + 0] * inv; o.y = oacc[dt][4 * g + 1] * inv;
                o.z = oacc[dt][4 * g + 2] * inv; o.w = oacc[dt][4 * g + 3] * inv;
                *reinterpret_cast<float4*>(op + dt * 32 + 8 * g + 4 * hh) = o;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Software-pipelined form for long sequences (8 waves, >= 2 key tiles).  One step = one 32-key sub-tile:
//     S_next = K(j+1) Q^T - m        12 MFMAs   } one scheduling region: the matrix pipe works on the NEXT
//     P      = exp2(S_cur), l += ..  ~100 VALU  } sub-tile's logits while the VALU finishes the current softmax
//     O     += V(j)^T P              12 MFMAs
// (in the plain kernel above the two waves of a SIMD run the same QK -> softmax -> PV sequence in lock step,
// re-aligned by the barrier every tile, so MFMA time and VALU time add up instead of overlapping).
// K lives in a 3-slot LDS ring (tile t+1's first sub-tile is needed while tile t is still being consumed),
// V in 2 slots; K[t+2] and V[t+1] are fetched to registers during tile t and stored before its closing barrier.
// Key masking of a ragged last tile is done through the MFMA's initial accumulator (-inf + x = -inf).
// NOMAX: the caller bounds the logits (vh_attention_args.logit_bound), so exp2 is taken of the raw logits: no row
// maximum, no rescale branch, a zero initial accumulator.  The phantom zero-logit keys then weigh exactly 1 each.
template <int D, bool NOMAX = false>
__global__ __launch_bounds__(512, 2) void attn_fwd_bf16x3_pipe(const AttnXK a) {
    constexpr int NT = 512;
    constexpr int KU = D / 4;
    constexpr int K_UNITS = KU * KT, V_UNITS = D * 2 * (KT / 8);
    __shared__ uint4 sK[3][K_UNITS];
    constexpr bool PLACED = VH_ATTN_PLACED && NOMAX && D == 64;     // the hand-placed three-stage step (below)
    __shared__ uint4 sV[PLACED ? 3 : 2][V_UNITS];           // (placed: a tile's values are still read one step into the next tile)

    const int t = threadIdx.x;
    const int wv = t >> 6, l = t & 63, lr = l & 31, hh = l >> 5;
    int bx, bh;
    attn_coords(a, bx, bh);
    const int b = bh / a.heads, hd = bh - b * a.heads;
    const int qrow = bx * 256 + wv * 32 + lr;

    bf16x8 qh[D / 16], ql[D / 16];
    {
        const float* Qb = a.q + ((size_t)bh * a.s + (qrow < a.s ? qrow : 0)) * D;
#pragma unroll
        for (int sl = 0; sl < D / 16; ++sl) {
            float v[8];
            const float4 p0 = *reinterpret_cast<const float4*>(Qb + sl * 16 + hh * 8);
            const float4 p1 = *reinterpret_cast<const float4*>(Qb + sl * 16 + hh * 8 + 4);
            v[0] = p0.x; v[1] = p0.y; v[2] = p0.z; v[3] = p0.w; v[4] = p1.x; v[5] = p1.y; v[6] = p1.z; v[7] = p1.w;
            if (qrow >= a.s) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.f;
            }
            split8(v, qh[sl], ql[sl]);
        }
    }
    f32x16 oacc[D / 32];
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float lsum = a.n_zero;

    constexpr int KPT = K_UNITS / NT, VPT = V_UNITS / NT;
    const uint4* kp[KPT];
    const uint4* vp[VPT];
    int ksl[KPT], vsl[VPT], kkey[KPT], vku[VPT];
    {
        const uint4* Kb = a.k + (size_t)bh * a.klp * KU;
        const uint4* Vb = a.vt + (size_t)bh * D * 2 * (a.klp / 8);
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int idx = t + i * NT;
            const int key = idx / KU, u = idx - key * KU;
            kp[i] = Kb + (size_t)key * KU + u;
            ksl[i] = u * KT + (key ^ (u & 7));
            kkey[i] = key;
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int idx = t + i * NT;
            const int row = idx >> 3, ku = idx & 7;
            const int d = row >> 1, hl = row & 1;
            vp[i] = Vb + (size_t)row * (a.klp / 8) + ku;
            vsl[i] = d * 16 + ((hl * 8 + ku) ^ (d & 15));
            vku[i] = ku;
        }
    }
    uint4 rk[KPT], rv[VPT];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    auto loadK = [&]() {
#pragma unroll
        for (int i = 0; i < KPT; ++i) { rk[i] = *kp[i]; kp[i] += KT * KU; }
    };
    auto loadV = [&]() {
#pragma unroll
        for (int i = 0; i < VPT; ++i) { rv[i] = *vp[i]; vp[i] += KT / 8; }
    };
    // `tail` (the ragged last tile only) is tested ONCE per call: the full-tile path carries no mask arithmetic (with the test
    // inside the element loops hipcc evaluated 14 compares + selects per tile whether or not a tail existed)
    auto storeK = [&](int slot, int k0, bool tail) __attribute__((always_inline)) {
        if (!tail) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) sK[slot][ksl[i]] = rk[i];
            return;
        }
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            uint4 v = rk[i];
            if (k0 + kkey[i] >= a.kl) v = zero4;
            sK[slot][ksl[i]] = v;
        }
    };
    auto storeV = [&](int slot, int k0, bool tail) __attribute__((always_inline)) {
        if (!tail) {
#pragma unroll
            for (int i = 0; i < VPT; ++i) sV[slot][vsl[i]] = rv[i];
            return;
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            uint4 v = rv[i];
            if (k0 + (vku[i] >> 1) * 16 + 16 > a.kl) {
                unsigned short* e = reinterpret_cast<unsigned short*>(&v);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int pos = vku[i] * 8 + j;
                    const int key = (pos & ~12) | ((pos & 4) << 1) | ((pos & 8) >> 1);
                    if (k0 + key >= a.kl) e[j] = 0;
                }
            }
            sV[slot][vsl[i]] = v;
        }
    };

    const int ntiles = (a.kl + KT - 1) / KT;
    const bool ragged = (a.kl % KT) != 0;
    auto is_tail = [&](int tile) { return ragged && tile == ntiles - 1; };

    // The running max enters the logits as the MFMA chain's initial accumulator: a register tile holding -m in
    // every element, kept across sub-tiles (16 VGPRs instead of 16 v_mov per sub-tile).
    f32x16 negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;
    // keys past the end of a ragged last tile get -inf logits (exp2 -> 0)
    auto mask_tail = [&](f32x16& sacc, int k0, int ks) {
        const int kbase = k0 + ks * 32 + 4 * hh;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (kbase + (r & 3) + 8 * (r >> 2) >= a.kl) sacc[r] = -INFINITY;
    };
    auto qk = [&](f32x16& sacc, int slot, int ks) __attribute__((always_inline)) {
        const int key = ks * 32 + lr;
#pragma unroll
        for (int sl = 0; sl < D / 16; ++sl) {
            const int uh = (sl * 2 + hh) * 2, ul = uh + 1;
            const bf16x8 kh = *reinterpret_cast<const bf16x8*>(&sK[slot][uh * KT + (key ^ (uh & 7))]);
            const bf16x8 kl_ = *reinterpret_cast<const bf16x8*>(&sK[slot][ul * KT + (key ^ (ul & 7))]);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl_, qh[sl], sl == 0 ? negm : sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[sl], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[sl], sacc, 0, 0, 0);
        }
    };

    f32x16 scur, snext;

    // one pipeline step: finish sub-tile (vslot, vks) whose logits are in scur; start the next one if any
    // (the sub-tile after the last one is computed too, from whatever the idle K slot holds, and never used: a
    // `has_next` branch around qk() would cut region B in two and its MFMAs would issue without the VALU work)
    auto step = [&](int vslot, int vks, int nslot, int nks, int nk0, bool ntail) __attribute__((always_inline)) {
        // region A: row max of the current logits (scur = s - m), rare raise of the running max
        float mx = 0.f;
        if constexpr (!NOMAX) {
            mx = scur[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, scur[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
        }
        if (!NOMAX && __any(mx > RESCALE_THR)) {
            const float dm = fmaxf(mx, 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-dm);
            lsum *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { scur[r] -= dm; negm[r] -= dm; }
#pragma unroll
            for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
        }
        // region B: next sub-tile's QK^T on the matrix pipe, this sub-tile's exp / sum / bf16 split on the VALU
        qk(snext, nslot, nks);
        float rs = 0.f;
        bf16x8 ph[2], pl[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pv[j] = __builtin_amdgcn_exp2f(scur[8 * s2 + j]);
                rs += pv[j];
            }
            split8(pv, ph[s2], pl[s2]);
        }
        rs += __shfl_xor(rs, 32);
        lsum += rs;
        // O^T += V^T P^T
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int kuh = vks * 4 + s2 * 2 + hh;
#pragma unroll
            for (int dt = 0; dt < D / 32; ++dt) {
                const int d = dt * 32 + lr;
                const bf16x8 vh = *reinterpret_cast<const bf16x8*>(&sV[vslot][d * 16 + (kuh ^ (d & 15))]);
                const bf16x8 vl = *reinterpret_cast<const bf16x8*>(&sV[vslot][d * 16 + ((8 + kuh) ^ (d & 15))]);
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph[s2], oacc[dt], 0, 0, 0);
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl[s2], oacc[dt], 0, 0, 0);
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph[s2], oacc[dt], 0, 0, 0);
            }
        }
        if (ntail) mask_tail(snext, nk0, nks);               // (after the matrix work: keeps region B one basic block)
        scur = snext;
    };

    // ---- hand-placed three-stage step (D = 64, no running maximum) ------------------------------------------------------
    // Two waves share a SIMD's matrix pipe and its vector issue port.  Per step a wave has 24 MFMAs (32 pipe cycles each, 8 of them
    // on the port) and ~90 VALU instructions (4 cycles of the port each, exp2 8): the pair needs 2 x (192 + ~400) port cycles against
    // 2 x 768 pipe cycles, so the pipe can only stay busy if EVERY MFMA gap carries its ~1/24 share of the VALU work - a gap with 7
    // VALU beside each of the two waves' MFMAs is port-bound (measured: the two-stage placed form, all softmax work beside the
    // QK^T MFMAs, gained 1 %).  A uniform spread needs the softmax to be independent of BOTH MFMA groups of its step, hence three
    // stages: step i issues QK^T of sub-tile i+1, exp2 / row sum / hi-lo split of sub-tile i, and P.V of sub-tile i-1 (P carried one
    // step in registers, V ring of 3 slots because a tile's values are read one step into the next tile).  The step is written as
    // 24 scheduling regions (sched_barrier(0): nothing moves across) of one MFMA + one third of a logit pair's softmax; LDS
    // fragments are read 3+ regions before their MFMA, the first K fragments of the next step in this step's tail (kpre_*).
    // Row sums stay per lane half (two running sums); the cross-half add happens once, after the key loop.
    bf16x8 kpre_h, kpre_l;
    // per-lane LDS byte addresses of the fragments in slot 0 (K: sub-tile 0; V: d < 32).  Slot, K sub-tile and d >= 32 are compile-time
    // in the placed loop (it is unrolled over the three ring positions), so they become the ds_read's immediate offset and a fragment
    // address costs no VALU at all (index form: v_or + v_lshl_add per address, ~20 of ~105 VALU per step)
    unsigned koff[D / 16][2], voff[2][2][2];                 // [slab][hi, lo]; [keys 0-31 / 32-63 of the tile][16-key half][hi, lo]
    {
        const unsigned kbase = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&sK[0][0];
        const unsigned vbase = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&sV[0][0];
#pragma unroll
        for (int sl = 0; sl < D / 16; ++sl)
#pragma unroll
            for (int lohi = 0; lohi < 2; ++lohi) {
                const int u = (sl * 2 + hh) * 2 + lohi;
                koff[sl][lohi] = kbase + (unsigned)(u * KT + (lr ^ (u & 7))) * 16u;
                asm("" : "+v"(koff[sl][lohi]));
                koff[sl][lohi] &= 0x3ffffu;              // (known-positive base: lets the slot / sub-tile constant fold into the ds_read offset field)
            }
#pragma unroll
        for (int vks = 0; vks < 2; ++vks)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int lohi = 0; lohi < 2; ++lohi) {
                    const int kuh = vks * 4 + s2 * 2 + hh;
                    voff[vks][s2][lohi] = vbase + (unsigned)(lr * 16 + ((lohi * 8 + kuh) ^ (lr & 15))) * 16u;
                    asm("" : "+v"(voff[vks][s2][lohi]));
                    voff[vks][s2][lohi] &= 0x3ffffu;
                }
    }
    typedef const __attribute__((address_space(3))) bf16x8* lds_frag_ptr;
    auto kfrag = [&](int slot, int ks, int sl, int lohi) __attribute__((always_inline)) -> bf16x8 {
        return *(lds_frag_ptr)(size_t)(koff[sl][lohi] + (unsigned)(slot * (K_UNITS * 16) + ks * 512));
    };
    auto vfrag = [&](int slot, int vks, int s2, int dt, int lohi) __attribute__((always_inline)) -> bf16x8 {
        return *(lds_frag_ptr)(size_t)(voff[vks][s2][lohi] + (unsigned)(slot * (V_UNITS * 16) + dt * 8192));
    };
#define VH_SB() __builtin_amdgcn_sched_barrier(0)
#if defined(VH_ATTN_ABLATE) && (VH_ATTN_ABLATE & 4)     // timing ablation build (WRONG results): every 32x32x16 MFMA as two 16x16x32 MFMAs (same FLOPs)
#define VH_MFMA(a_, b_, c_) mfma_abl16(a_, b_, c_)
#else
#define VH_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, c_, 0, 0, 0)
#endif
#define VH_MFMA_O(a_, b_, c_) c_ = VH_MFMA(a_, b_, c_)
    float lsum0 = 0.f, lsum1 = 0.f;
    // pin / pout: {hi keys 0-15, lo keys 0-15, hi keys 16-31, lo keys 16-31} of the previous / this sub-tile's P
    // skc / svc: integral constants, the K / V ring slot this step's regions also fill from the staging registers (-1: none; the
    // generic tile body stores after its steps, with the ragged-tail masks)
    auto step3 = [&](int vslot, int vks, int nslot, int nks, int nk0, bool ntail, int pslot, int pks,
                     const bf16x8 (&pin)[4], bf16x8 (&pout)[4], auto skc, auto svc) __attribute__((always_inline)) {
      if constexpr (D == 64) {
        constexpr int SK = decltype(skc)::value, SV = decltype(svc)::value;
        static_assert(KPT == 2 && VPT == 2, "step3 places two K and two V staging stores");
        float e[16];
        unsigned hw[8], lw[8];
        float t0[8], t1[8];
        auto cvt2 = [](float x, float y) __attribute__((always_inline)) -> unsigned {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            bf16x2 v; v[0] = (__bf16)x; v[1] = (__bf16)y;                // one v_cvt_pk_bf16_f32
            unsigned r = __builtin_bit_cast(unsigned, v);
            asm("" : "+v"(r));                                           // opaque: hipcc otherwise converts x a second time on its own for x - hi(x)
            return r;
        };
        // one logit pair = 10 VALU in three pieces
        auto EA = [&](int p) __attribute__((always_inline)) {          // 2 exp2
            e[2 * p] = __builtin_amdgcn_exp2f(scur[2 * p]); e[2 * p + 1] = __builtin_amdgcn_exp2f(scur[2 * p + 1]);
        };
        auto EB = [&](int p) __attribute__((always_inline)) {          // 5 VALU
            hw[p] = cvt2(e[2 * p], e[2 * p + 1]);
            lsum0 += e[2 * p]; lsum1 += e[2 * p + 1];
            t0[p] = __uint_as_float(hw[p] << 16); t1[p] = __uint_as_float(hw[p] & 0xffff0000u);
        };
        auto EC = [&](int p) __attribute__((always_inline)) {          // 3 VALU
            lw[p] = cvt2(e[2 * p] - t0[p], e[2 * p + 1] - t1[p]);
        };
        auto frag4 = [](unsigned a0, unsigned a1, unsigned a2, unsigned a3) __attribute__((always_inline)) -> bf16x8 {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 v = {a0, a1, a2, a3};
            return __builtin_bit_cast(bf16x8, v);
        };
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        // 1
        const bf16x8 k1l = kfrag(nslot, nks, 1, 1), k1h = kfrag(nslot, nks, 1, 0);
        const bf16x8 v00l = vfrag(vslot, vks, 0, 0, 1), v00h = vfrag(vslot, vks, 0, 0, 0);
        snext = VH_MFMA(kpre_l, qh[0], z); EA(0); VH_SB();
        snext = VH_MFMA(kpre_h, ql[0], snext); EB(0); VH_SB();
        snext = VH_MFMA(kpre_h, qh[0], snext); EC(0); VH_SB();
        // 4
        const bf16x8 k2l = kfrag(nslot, nks, 2, 1), k2h = kfrag(nslot, nks, 2, 0);
        VH_MFMA_O(v00l, pin[0], oacc[0]); EA(1); VH_SB();
        if constexpr (SK >= 0) sK[SK][ksl[0]] = rk[0];
        snext = VH_MFMA(k1l, qh[1], snext); EB(1); VH_SB();
        const bf16x8 v01l = vfrag(vslot, vks, 0, 1, 1), v01h = vfrag(vslot, vks, 0, 1, 0);
        VH_MFMA_O(v00h, pin[1], oacc[0]); EC(1); VH_SB();
        // 7
        snext = VH_MFMA(k1h, ql[1], snext); EA(2); VH_SB();
        if constexpr (SK >= 0) sK[SK][ksl[1]] = rk[1];
        VH_MFMA_O(v00h, pin[0], oacc[0]); EB(2); VH_SB();
        const bf16x8 k3l = kfrag(nslot, nks, 3, 1), k3h = kfrag(nslot, nks, 3, 0);
        snext = VH_MFMA(k1h, qh[1], snext); EC(2); VH_SB();
        // 10
        VH_MFMA_O(v01l, pin[0], oacc[1]); EA(3); VH_SB();
        const bf16x8 v10l = vfrag(vslot, vks, 1, 0, 1), v10h = vfrag(vslot, vks, 1, 0, 0);
        snext = VH_MFMA(k2l, qh[2], snext); EB(3); VH_SB();
        if constexpr (SV >= 0) sV[SV][vsl[0]] = rv[0];
        VH_MFMA_O(v01h, pin[1], oacc[1]); EC(3); VH_SB();
        // 13
        snext = VH_MFMA(k2h, ql[2], snext); EA(4); VH_SB();
        VH_MFMA_O(v01h, pin[0], oacc[1]); EB(4); VH_SB();
        const bf16x8 v11l = vfrag(vslot, vks, 1, 1, 1), v11h = vfrag(vslot, vks, 1, 1, 0);
        snext = VH_MFMA(k2h, qh[2], snext); EC(4); VH_SB();
        // 16
        if constexpr (SV >= 0) sV[SV][vsl[1]] = rv[1];
        VH_MFMA_O(v10l, pin[2], oacc[0]); EA(5); VH_SB();
        snext = VH_MFMA(k3l, qh[3], snext); EB(5); VH_SB();
        VH_MFMA_O(v10h, pin[3], oacc[0]); EC(5); VH_SB();
        // 19
        snext = VH_MFMA(k3h, ql[3], snext); EA(6); VH_SB();
        kpre_l = kfrag(pslot, pks, 0, 1); kpre_h = kfrag(pslot, pks, 0, 0);
        VH_MFMA_O(v10h, pin[2], oacc[0]); EB(6); VH_SB();
        snext = VH_MFMA(k3h, qh[3], snext); EC(6); VH_SB();
        // 22
        VH_MFMA_O(v11l, pin[2], oacc[1]); EA(7); VH_SB();
        VH_MFMA_O(v11h, pin[3], oacc[1]); EB(7); VH_SB();
        VH_MFMA_O(v11h, pin[2], oacc[1]); EC(7); VH_SB();
        // (the row sums are needed only after the loop and hipcc sinks their adds into the loop's last block, keeping all 32 exp2
        //  results alive until then; one volatile use per step keeps them in this block, where the sched_barriers hold them in place)
        asm volatile("" : "+v"(lsum0), "+v"(lsum1));
        pout[0] = frag4(hw[0], hw[1], hw[2], hw[3]); pout[1] = frag4(lw[0], lw[1], lw[2], lw[3]);
        pout[2] = frag4(hw[4], hw[5], hw[6], hw[7]); pout[3] = frag4(lw[4], lw[5], lw[6], lw[7]);
        if (ntail) mask_tail(snext, nk0, nks);
        scur = snext;
      }
    };
    // P.V of the last sub-tile (its P is still in registers when the key loop ends)
    auto drain3 = [&](int vslot, int vks, const bf16x8 (&pin)[4]) __attribute__((always_inline)) {
      if constexpr (D == 64) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16x8 vl = vfrag(vslot, vks, s2, dt, 1), vh = vfrag(vslot, vks, s2, dt, 0);
                VH_MFMA_O(vl, pin[2 * s2], oacc[dt]);
                VH_MFMA_O(vh, pin[2 * s2 + 1], oacc[dt]);
                VH_MFMA_O(vh, pin[2 * s2], oacc[dt]);
            }
      }
    };

    // prologue: K[0], V[0], K[1] in LDS; logits of sub-tile 0
    loadK(); loadV();
    storeK(0, 0, is_tail(0)); storeV(0, 0, is_tail(0));
    if constexpr (PLACED) {                                  // the first step multiplies P = 0 with "tile -1": that slot must hold finite numbers
#pragma unroll
        for (int i = 0; i < VPT; ++i) sV[2][vsl[i]] = zero4;
    }
    if (ntiles > 1) { loadK(); storeK(1, KT, is_tail(1)); }
    __syncthreads();
    qk(scur, 0, 0);
    if (is_tail(0)) mask_tail(scur, 0, 0);
    if (!NOMAX && !(a.n_zero > 0.f)) {                      // no max yet: adopt sub-tile 0's (O and l are still 0)
        float mx = scur[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, scur[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
#pragma unroll
        for (int r = 0; r < 16; ++r) { scur[r] -= mx; negm[r] = -mx; }
    }

#ifdef VH_CLOCK
    unsigned long long ck_m0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    int ks0 = 0, ks1 = 1, ks2 = 2;
    if constexpr (PLACED) {
        bf16x8 PA[4], PB[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) PB[i][j] = (__bf16)0.f;
        kpre_l = kfrag(0, 1, 0, 1); kpre_h = kfrag(0, 1, 0, 0);
        // ring position of tile t = t mod 3 for K and V alike: K of tiles t, t+1, t+2 and V of tiles t, t+1, t-1 sit in slots (p, p+1, p+2) mod 3
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using IN = std::integral_constant<int, -1>;
        auto tile3 = [&](auto pc, int tile) __attribute__((always_inline)) {
            constexpr int P0 = decltype(pc)::value, P1 = (P0 + 1) % 3, P2 = (P0 + 2) % 3;
            const int k0 = tile * KT;
            const bool more1 = tile + 1 < ntiles, more2 = tile + 2 < ntiles;
            if (more2) loadK();
            if (more1) loadV();
            step3(P2, 1, P0, 1, k0, is_tail(tile), P1, 0, PB, PA, IN{}, IN{});        // P.V of (tile-1, keys 32..63)
            step3(P0, 0, P1, 0, k0 + KT, is_tail(tile + 1), P1, 1, PA, PB, IN{}, IN{}); // P.V of (tile, keys 0..31); the next tile's first step reads K slot P1
            if (more2) storeK(P2, k0 + 2 * KT, is_tail(tile + 2));
            if (more1) storeV(P1, k0 + KT, is_tail(tile + 1));
            __syncthreads();
        };
        // a tile whose two successors exist and hold no ragged tail: no branches, the staging stores ride in the second step's regions
        auto tileM = [&](auto pc, int tile) __attribute__((always_inline)) {
            constexpr int P0 = decltype(pc)::value, P1 = (P0 + 1) % 3, P2 = (P0 + 2) % 3;
            const int k0 = tile * KT;
            loadK(); loadV();
            // (the tail tests are never true here; left in as run-time branches they end the basic block after each step - in one block
            //  of two or six steps instruction selection already emits the first step's VALU in clumps, before any sched_barrier applies)
            step3(P2, 1, P0, 1, k0, is_tail(tile), P1, 0, PB, PA, IN{}, IN{});
            step3(P0, 0, P1, 0, k0 + KT, is_tail(tile + 1), P1, 1, PA, PB, std::integral_constant<int, P2>{}, std::integral_constant<int, P1>{});
#if !(defined(VH_ATTN_ABLATE) && (VH_ATTN_ABLATE & 2))   // timing ablation build (WRONG results): no tile barrier
            __syncthreads();
#endif
        };
        const int nmain = ntiles - (ragged ? 3 : 2);
        int tile = 0;
        for (; tile + 3 <= nmain; tile += 3) {              // three tiles per trip, straight-line (a per-tile dispatch on the ring phase made hipcc shuffle the live state between the bodies)
            tileM(I0{}, tile); tileM(I1{}, tile + 1); tileM(I2{}, tile + 2);
        }
        if (tile < ntiles) tile3(I0{}, tile);               // the last <= 5 tiles, generic
        if (tile + 1 < ntiles) tile3(I1{}, tile + 1);
        if (tile + 2 < ntiles) tile3(I2{}, tile + 2);
        if (tile + 3 < ntiles) tile3(I0{}, tile + 3);
        if (tile + 4 < ntiles) tile3(I1{}, tile + 4);
        const int vs2 = (ntiles + 2) % 3;                   // slot of the last tile ((ntiles - 1) mod 3)
        drain3(vs2, 1, PB);
        lsum += lsum0 + lsum1;
        lsum += __shfl_xor(lsum0 + lsum1, 32);
    } else
    for (int tile = 0; tile < ntiles; ++tile) {
        const int k0 = tile * KT;
        const bool more1 = tile + 1 < ntiles, more2 = tile + 2 < ntiles;
#if defined(VH_ATTN_ABLATE) && (VH_ATTN_ABLATE & 1)   // timing ablation build (WRONG results): no K/V staging
        (void)more1; (void)more2;
        step(tile & 1, 0, ks0, 1, k0, false);
        step(tile & 1, 1, ks1, 0, k0 + KT, false);
#else
        if (more2) loadK();
        if (more1) loadV();
        step(tile & 1, 0, ks0, 1, k0, is_tail(tile));
        step(tile & 1, 1, ks1, 0, k0 + KT, is_tail(tile + 1));
        if (more2) storeK(ks2, k0 + 2 * KT, is_tail(tile + 2));
        if (more1) storeV((tile + 1) & 1, k0 + KT, is_tail(tile + 1));
#endif
#if !(defined(VH_ATTN_ABLATE) && (VH_ATTN_ABLATE & 2))   // timing ablation build (WRONG results): no tile barrier
        __syncthreads();
#endif
        const int tmp = ks0; ks0 = ks1; ks1 = ks2; ks2 = tmp;
    }
#ifdef VH_CLOCK
    {
        const unsigned long long ck_m1 = __builtin_amdgcn_s_memtime(), ck_r1 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (a.dbg && t == 0) { a.dbg[(size_t)blockIdx.x * 2] = ck_m1 - ck_m0; a.dbg[(size_t)blockIdx.x * 2 + 1] = ck_r1 - ck_r0; }
    }
#endif

    if (qrow < a.s && a.out_s8) {
        const float inv = 1.0f / lsum;
        unsigned short* op = reinterpret_cast<unsigned short*>(a.out) + (((size_t)b * a.s + qrow) * a.c + hd * D) * 2;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned h[4], lo4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = oacc[dt][4 * g + j] * inv;
                    h[j] = bf16_rn_bits(v);
                    lo4[j] = bf16_rn_bits(v - __uint_as_float(h[j] << 16));
                }
                unsigned short* q8 = op + (dt * 32 + 8 * g) * 2 + 4 * hh;
                *reinterpret_cast<uint2*>(q8) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
                *reinterpret_cast<uint2*>(q8 + 8) = make_uint2(lo4[0] | (lo4[1] << 16), lo4[2] | (lo4[3] << 16));
            }
    } else if (qrow < a.s) {
        const float inv = 1.0f / lsum;
        float* op = a.out + ((size_t)b * a.s + qrow) * a.c + hd * D;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 o;
                o.x = oacc[dt][4 * g + 0] * inv; o.y = oacc[dt][4 * g + 1] * inv;
                o.z = oacc[dt][4 * g + 2] * inv; o.w = oacc[dt][4 * g + 3] * inv;
                *reinterpret_cast<float4*>(op + dt * 32 + 8 * g + 4 * hh) = o;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// attn_fwd_x3_m16: 64-channel heads, bounded logits (no running maximum), KL > 64 - the attention of the benchmark's networks.
// The three-stage placed step of attn_fwd_bf16x3_pipe (step i: QK^T of sub-tile i+1, exp2 / row sum / hi-lo split of sub-tile i, P.V of
// sub-tile i-1; compile-time ring slots, staging stores inside the step) on v_mfma_f32_16x16x32_bf16: same FLOPs and matrix-pipe cycles
// per step as the 32x32x16 form (48 x 16 instead of 24 x 32 cycles), but the chip holds a higher clock under it - measured here with
// the in-kernel stamps, same device, 2.10 GHz against 1.62 GHz at +2 % cycles (MI355X_MICROARCH.md, DVFS give-back item 7; the
// convolutions use the same shape for the same reason).
//   S^T tile (mk, nq) = 16 keys x 16 queries; a wave holds 2 x 2 of them per 32-key sub-tile (32 queries per wave as before).
//   A = K rows: lane (i = l&15, g = l>>4) supplies key (i&7) + 8 mk + 16 (i>>3) of the sub-tile, channels 32 kstep + 8 g .. +7: with
//   that row order the accumulators of lane group g (rows 4g..4g+3 of both tiles) are exactly the eight keys of one 16-byte unit of
//   the V^T image (positions 8h..8h+7 of 16-key block blk, g = 2 blk + h), so P never leaves registers here either: the two tiles'
//   accumulators, split to bf16 hi/lo pairs, are the B operand (32 keys x 16 queries) of O^T += V^T P^T, and the global V^T format is
//   the one the 32x32x16 kernels read.  K slots in LDS carry one more swizzle term (bit 3 ^= key bit 4) so that those 16 rows fall in
//   16 different bank groups.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int D>
__global__ __launch_bounds__(512, 2) void attn_fwd_x3_m16(const AttnXK a) {
    static_assert(D == 64 || D == 32, "head dims 64 and 32");
    constexpr int NT = 512, KU = D / 4;
    constexpr int KSTEPS = D / 32, MD = D / 16;              // 32-channel steps of QK^T; 16-channel blocks of O^T
    constexpr int K_UNITS = KU * KT, V_UNITS = D * 2 * (KT / 8);
    constexpr int KPT = K_UNITS / NT, VPT = V_UNITS / NT;
    static_assert(KPT == VPT && (KPT == 2 || KPT == 1), "K and V staging units per thread and tile");
    __shared__ uint4 sK[3][K_UNITS];
    __shared__ uint4 sV[3][V_UNITS];

    const int t = threadIdx.x;
    const int wv = t >> 6, l = t & 63, li = l & 15, g = l >> 4;
    int bx, bh;
    attn_coords(a, bx, bh);
    const int b = bh / a.heads, hd = bh - b * a.heads;
    const int qrow0 = bx * 256 + wv * 32 + li;               // this lane's query in block nq: qrow0 + 16 nq

    // Q fragments (B operand): query li of block nq, channels 32 kstep + 8 g .. +7
    bf16x8 qh[2][KSTEPS], ql[2][KSTEPS];
#pragma unroll
    for (int nq = 0; nq < 2; ++nq) {
        const int qrow = qrow0 + 16 * nq;
        const float* Qb = a.q + ((size_t)bh * a.s + (qrow < a.s ? qrow : 0)) * D + g * 8;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            float v[8];
            const float4 p0 = *reinterpret_cast<const float4*>(Qb + ks * 32);
            const float4 p1 = *reinterpret_cast<const float4*>(Qb + ks * 32 + 4);
            v[0] = p0.x; v[1] = p0.y; v[2] = p0.z; v[3] = p0.w; v[4] = p1.x; v[5] = p1.y; v[6] = p1.z; v[7] = p1.w;
            if (qrow >= a.s) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.f;
            }
            split8(v, qh[nq][ks], ql[nq][ks]);
        }
    }
    f32x4 o[MD][2];                                           // O^T: channels 16 md + 4 g + r of query (nq, li)
#pragma unroll
    for (int md = 0; md < MD; ++md)
#pragma unroll
        for (int nq = 0; nq < 2; ++nq)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[md][nq][r] = 0.f;

    // staging: global -> registers -> LDS, as in attn_fwd_bf16x3_pipe (K rows get the extra swizzle term)
    const uint4* kp[KPT];
    const uint4* vp[VPT];
    int ksl[KPT], vsl[VPT], kkey[KPT], vku[VPT];
    {
        const uint4* Kb = a.k + (size_t)bh * a.klp * KU;
        const uint4* Vb = a.vt + (size_t)bh * D * 2 * (a.klp / 8);
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int idx = t + i * NT;
            const int key = idx / KU, u = idx - key * KU;
            kp[i] = Kb + (size_t)key * KU + u;
            ksl[i] = u * KT + (key ^ (u & 7) ^ (((key >> 4) & 1) << 3));
            kkey[i] = key;
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int idx = t + i * NT;
            const int row = idx >> 3, ku = idx & 7;
            const int d = row >> 1, hl = row & 1;
            vp[i] = Vb + (size_t)row * (a.klp / 8) + ku;
            vsl[i] = d * 16 + ((hl * 8 + ku) ^ (d & 15));
            vku[i] = ku;
        }
    }
    uint4 rk[KPT], rv[VPT];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    auto loadK = [&]() {
#pragma unroll
        for (int i = 0; i < KPT; ++i) { rk[i] = *kp[i]; kp[i] += KT * KU; }
    };
    auto loadV = [&]() {
#pragma unroll
        for (int i = 0; i < VPT; ++i) { rv[i] = *vp[i]; vp[i] += KT / 8; }
    };
    auto storeK = [&](int slot, int k0, bool tail) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            uint4 v = rk[i];
            if (tail && k0 + kkey[i] >= a.kl) v = zero4;
            sK[slot][ksl[i]] = v;
        }
    };
    auto storeV = [&](int slot, int k0, bool tail) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            uint4 v = rv[i];
            if (tail && k0 + (vku[i] >> 1) * 16 + 16 > a.kl) {
                unsigned short* e = reinterpret_cast<unsigned short*>(&v);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int pos = vku[i] * 8 + j;
                    const int key = (pos & ~12) | ((pos & 4) << 1) | ((pos & 8) >> 1);
                    if (k0 + key >= a.kl) e[j] = 0;
                }
            }
            sV[slot][vsl[i]] = v;
        }
    };

    const int ntiles = (a.kl + KT - 1) / KT;
    const bool ragged = (a.kl % KT) != 0;
    auto is_tail = [&](int tile) { return ragged && tile == ntiles - 1; };

    // fragment addresses: per-lane LDS byte addresses in slot 0 (K: first 32 keys of the tile); slot, sub-tile and 16-channel block
    // are compile-time offsets of the ds_read
    unsigned koff[2][KSTEPS][2], voff[2][2];                      // [mk][kstep][hi, lo]; [keys 0-31 / 32-63][hi, lo]
    {
        const unsigned kbase = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&sK[0][0];
        const unsigned vbase = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&sV[0][0];
#pragma unroll
        for (int mk = 0; mk < 2; ++mk)
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
                for (int lohi = 0; lohi < 2; ++lohi) {
                    const int key = (li & 7) + 8 * mk + 16 * (li >> 3);
                    const int u = (ks * 4 + g) * 2 + lohi;
                    koff[mk][ks][lohi] = kbase + (unsigned)(u * KT + (key ^ (u & 7) ^ ((li >> 3) << 3))) * 16u;
                    asm("" : "+v"(koff[mk][ks][lohi]));
                    koff[mk][ks][lohi] &= 0x3ffffu;
                }
#pragma unroll
        for (int vks = 0; vks < 2; ++vks)
#pragma unroll
            for (int lohi = 0; lohi < 2; ++lohi) {
                voff[vks][lohi] = vbase + (unsigned)(li * 16 + ((lohi * 8 + vks * 4 + g) ^ li)) * 16u;
                asm("" : "+v"(voff[vks][lohi]));
                voff[vks][lohi] &= 0x3ffffu;
            }
    }
    typedef const __attribute__((address_space(3))) bf16x8* lds_frag_ptr;
    auto kfrag = [&](int slot, int ks32, int mk, int kstep, int lohi) __attribute__((always_inline)) -> bf16x8 {
        return *(lds_frag_ptr)(size_t)(koff[mk][kstep][lohi] + (unsigned)(slot * (K_UNITS * 16) + ks32 * 512));
    };
    auto vfrag = [&](int slot, int vks, int md, int lohi) __attribute__((always_inline)) -> bf16x8 {
        return *(lds_frag_ptr)(size_t)(voff[vks][lohi] + (unsigned)(slot * (V_UNITS * 16) + md * 4096));
    };
#define VH_MFMA16(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, b_, c_, 0, 0, 0)
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    // keys past the end of a ragged last tile get -inf logits: element r of tile mk in lane group g is key 4(g&1) + r + 16(g>>1) + 8 mk
    auto mask_tail = [&](f32x4 (&s)[2][2], int k0, int ks32) {
        const int kb = k0 + ks32 * 32 + 4 * (g & 1) + 16 * (g >> 1);
#pragma unroll
        for (int mk = 0; mk < 2; ++mk)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (kb + 8 * mk + r >= a.kl) { s[mk][0][r] = -INFINITY; s[mk][1][r] = -INFINITY; }
    };

    f32x4 scur[2][2], snext[2][2];
    float lsum[2] = {0.f, 0.f};                              // per lane: its 8 keys per step of query (nq, li); the 4 lane groups are added after the loop
    bf16x8 kpre[2], vpre[2];                                 // first K / V fragment pair of the next step, read one step ahead ([hi, lo])

    // one step = 24 regions of (one QK^T MFMA, one P.V MFMA, one third of a logit pair's softmax); pin / pout: [nq][hi, lo]
    auto step = [&](auto vslc, auto vksc, auto nslc, auto nksc, int nk0, bool ntail, auto pslc, auto pksc, auto pvslc, auto pvksc,
                    const bf16x8 (&pin)[2][2], bf16x8 (&pout)[2][2], auto skc, auto svc) __attribute__((always_inline)) {
        constexpr int VSL = decltype(vslc)::value, VKS = decltype(vksc)::value, NSL = decltype(nslc)::value, NKS = decltype(nksc)::value;
        constexpr int PSL = decltype(pslc)::value, PKS = decltype(pksc)::value, PVSL = decltype(pvslc)::value, PVKS = decltype(pvksc)::value;
        constexpr int SK = decltype(skc)::value, SV = decltype(svc)::value;
        float e[16];
        unsigned hw[8], lw[8];
        float t0[8], t1[8];
        // MFMA lists: QK^T blocks (mk, kstep) of 6 = {lo.hi, lo.hi | hi.lo, hi.lo | hi.hi, hi.hi} for query blocks 0, 1; P.V blocks of 6 per
        // 16 channels.  D = 64: 24 + 24, one of each per region; D = 32: 12 + 12, alternating regions.
        constexpr int NQB = 2 * KSTEPS, NPB = MD;             // blocks of 6 MFMAs
        bf16x8 kf[NQB][2], vf[NPB][2];                        // [block][hi, lo]
        kf[0][0] = kpre[0]; kf[0][1] = kpre[1]; vf[0][0] = vpre[0]; vf[0][1] = vpre[1];
        auto cvt2 = [](float x, float y) __attribute__((always_inline)) -> unsigned {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            bf16x2 v; v[0] = (__bf16)x; v[1] = (__bf16)y;
            unsigned r = __builtin_bit_cast(unsigned, v);
            asm("" : "+v"(r));                               // opaque: hipcc otherwise converts x a second time on its own for x - hi(x)
            return r;
        };
        auto frag4 = [](unsigned a0, unsigned a1, unsigned a2, unsigned a3) __attribute__((always_inline)) -> bf16x8 {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 v = {a0, a1, a2, a3};
            return __builtin_bit_cast(bf16x8, v);
        };
        static_for<0, 24>([&](auto rc) __attribute__((always_inline)) {
            constexpr int R = decltype(rc)::value;
            constexpr bool HAS_Q = D == 64 || (R & 1) == 0, HAS_P = D == 64 || (R & 1) == 1;
            constexpr int QI = D == 64 ? R : R / 2, PI = D == 64 ? R : R / 2;
            // fragment reads, issued five MFMAs of their list ahead of the block's first one
            if constexpr (HAS_Q && QI % 6 == 1 && QI / 6 + 1 < NQB) {
                constexpr int nb = QI / 6 + 1;
                kf[nb][0] = kfrag(NSL, NKS, nb / KSTEPS, nb % KSTEPS, 0); kf[nb][1] = kfrag(NSL, NKS, nb / KSTEPS, nb % KSTEPS, 1);
            }
            if constexpr (HAS_P && PI % 6 == 1 && PI / 6 + 1 < NPB) {
                constexpr int nb = PI / 6 + 1;
                vf[nb][0] = vfrag(VSL, VKS, nb, 0); vf[nb][1] = vfrag(VSL, VKS, nb, 1);
            }
            if constexpr (R == 19) {
                kpre[0] = kfrag(PSL, PKS, 0, 0, 0); kpre[1] = kfrag(PSL, PKS, 0, 0, 1);
                vpre[0] = vfrag(PVSL, PVKS, 0, 0); vpre[1] = vfrag(PVSL, PVKS, 0, 1);
            }
            if constexpr (SK >= 0 && R == 4) sK[SK][ksl[0]] = rk[0];
            if constexpr (SK >= 0 && R == 8 && KPT == 2) sK[SK][ksl[KPT - 1]] = rk[KPT - 1];
            if constexpr (SV >= 0 && R == 12) sV[SV][vsl[0]] = rv[0];
            if constexpr (SV >= 0 && R == 16 && VPT == 2) sV[SV][vsl[VPT - 1]] = rv[VPT - 1];
            if constexpr (HAS_Q) {
                constexpr int blk = QI / 6, j = QI % 6, nq = j & 1, mk = blk / KSTEPS, kstep = blk % KSTEPS;
                const bf16x8 afrag = kf[blk][j < 2 ? 1 : 0];
                const bf16x8 bfrag = (j == 2 || j == 3) ? ql[nq][kstep] : qh[nq][kstep];
                snext[mk][nq] = VH_MFMA16(afrag, bfrag, (kstep == 0 && j < 2) ? zero : snext[mk][nq]);
            }
            if constexpr (HAS_P) {
                constexpr int blk = PI / 6, j = PI % 6, nq = j & 1;
                const bf16x8 afrag = vf[blk][j < 2 ? 1 : 0];
                const bf16x8 bfrag = (j == 2 || j == 3) ? pin[nq][1] : pin[nq][0];
                o[blk][nq] = VH_MFMA16(afrag, bfrag, o[blk][nq]);
            }
            // softmax of pair p = R / 3: word w of query block q2's P fragment = elements 2(w&1), 2(w&1)+1 of tile mk = w >> 1
            {
                constexpr int p = R / 3, piece = R % 3, q2 = p >> 2, w = p & 3, mk = w >> 1, r0 = 2 * (w & 1);
                if constexpr (piece == 0) {
                    e[2 * p] = __builtin_amdgcn_exp2f(scur[mk][q2][r0]); e[2 * p + 1] = __builtin_amdgcn_exp2f(scur[mk][q2][r0 + 1]);
                } else if constexpr (piece == 1) {
                    hw[p] = cvt2(e[2 * p], e[2 * p + 1]);
                    lsum[q2] += e[2 * p]; lsum[q2] += e[2 * p + 1];
                    t0[p] = __uint_as_float(hw[p] << 16); t1[p] = __uint_as_float(hw[p] & 0xffff0000u);
                } else {
                    lw[p] = cvt2(e[2 * p] - t0[p], e[2 * p + 1] - t1[p]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // (row sums are needed only after the loop: one volatile use per step keeps their adds in this block, where the sched_barriers hold them)
        asm volatile("" : "+v"(lsum[0]), "+v"(lsum[1]));
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
            pout[q2][0] = frag4(hw[4 * q2], hw[4 * q2 + 1], hw[4 * q2 + 2], hw[4 * q2 + 3]);
            pout[q2][1] = frag4(lw[4 * q2], lw[4 * q2 + 1], lw[4 * q2 + 2], lw[4 * q2 + 3]);
        }
        if (ntail) mask_tail(snext, nk0, NKS);
#pragma unroll
        for (int mk = 0; mk < 2; ++mk)
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) scur[mk][q2] = snext[mk][q2];
    };

    // prologue: K[0], V[0], K[1] in LDS; logits of sub-tile 0; V slot 2 ("tile -1", multiplied by P = 0 in the first step) zeroed
    loadK(); loadV();
    storeK(0, 0, is_tail(0)); storeV(0, 0, is_tail(0));
#pragma unroll
    for (int i = 0; i < VPT; ++i) sV[2][vsl[i]] = zero4;
    if (ntiles > 1) { loadK(); storeK(1, KT, is_tail(1)); }
    __syncthreads();
#pragma unroll
    for (int mk = 0; mk < 2; ++mk)
#pragma unroll
        for (int nq = 0; nq < 2; ++nq) {
            f32x4 acc = zero;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const bf16x8 kh = kfrag(0, 0, mk, ks, 0), kl_ = kfrag(0, 0, mk, ks, 1);
                acc = VH_MFMA16(kl_, qh[nq][ks], acc);
                acc = VH_MFMA16(kh, ql[nq][ks], acc);
                acc = VH_MFMA16(kh, qh[nq][ks], acc);
            }
            scur[mk][nq] = acc;
        }
    if (is_tail(0)) mask_tail(scur, 0, 0);

    bf16x8 PA[2][2], PB[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int j = 0; j < 8; ++j) PB[i][h2][j] = (__bf16)0.f;
    kpre[0] = kfrag(0, 1, 0, 0, 0); kpre[1] = kfrag(0, 1, 0, 0, 1);          // the first step computes the logits of (tile 0, keys 32..63)
    vpre[0] = vfrag(2, 1, 0, 0); vpre[1] = vfrag(2, 1, 0, 1);                //   and multiplies P = 0 with "tile -1"
#ifdef VH_CLOCK
    unsigned long long ck_m0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    // ring position of tile t = t mod 3 for K and V alike: K of tiles t, t+1, t+2 and V of tiles t, t+1, t-1 sit in slots (p, p+1, p+2) mod 3
    using IN = std::integral_constant<int, -1>;
    auto tile_body = [&](auto pc, auto mainc, int tile) __attribute__((always_inline)) {
        constexpr int P0 = decltype(pc)::value, P1 = (P0 + 1) % 3, P2 = (P0 + 2) % 3;
        constexpr bool MAIN = decltype(mainc)::value;        // two successors, no ragged tail among them: no conditions, staging stores inside the second step
        using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
        using S0 = std::integral_constant<int, P0>; using S1 = std::integral_constant<int, P1>; using S2 = std::integral_constant<int, P2>;
        const int k0 = tile * KT;
        const bool more1 = MAIN || tile + 1 < ntiles, more2 = MAIN || tile + 2 < ntiles;
        if (more2) loadK();
        if (more1) loadV();
        // step 1: logits of (tile, keys 32..63) from K slot P0; P.V of (tile-1, keys 32..63) from V slot P2; next step: K (tile+1, 0..31) = slot P1, V (tile, 0..31) = slot P0
        step(S2{}, C1{}, S0{}, C1{}, k0, is_tail(tile), S1{}, C0{}, S0{}, C0{}, PB, PA, IN{}, IN{});
        // step 2: logits of (tile+1, keys 0..31) from K slot P1; P.V of (tile, keys 0..31); next step: K (tile+1, 32..63) = slot P1, V (tile, 32..63) = slot P0
        if constexpr (MAIN) step(S0{}, C0{}, S1{}, C0{}, k0 + KT, is_tail(tile + 1), S1{}, C1{}, S0{}, C1{}, PA, PB, S2{}, S1{});
        else {
            step(S0{}, C0{}, S1{}, C0{}, k0 + KT, is_tail(tile + 1), S1{}, C1{}, S0{}, C1{}, PA, PB, IN{}, IN{});
            if (more2) storeK(P2, k0 + 2 * KT, is_tail(tile + 2));
            if (more1) storeV(P1, k0 + KT, is_tail(tile + 1));
        }
        __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    const int nmain = ntiles - (ragged ? 3 : 2);
    int tile = 0;
    for (; tile + 3 <= nmain; tile += 3) {                  // three tiles per trip, straight-line
        tile_body(I0{}, std::true_type{}, tile); tile_body(I1{}, std::true_type{}, tile + 1); tile_body(I2{}, std::true_type{}, tile + 2);
    }
    if (tile < ntiles) tile_body(I0{}, std::false_type{}, tile);             // the last <= 5 tiles, generic
    if (tile + 1 < ntiles) tile_body(I1{}, std::false_type{}, tile + 1);
    if (tile + 2 < ntiles) tile_body(I2{}, std::false_type{}, tile + 2);
    if (tile + 3 < ntiles) tile_body(I0{}, std::false_type{}, tile + 3);
    if (tile + 4 < ntiles) tile_body(I1{}, std::false_type{}, tile + 4);
    // P.V of the last sub-tile (its P is still in registers): V slot (ntiles - 1) mod 3, keys 32..63
    {
        const int vs = (ntiles + 2) % 3;
#pragma unroll
        for (int md = 0; md < MD; ++md) {
            const bf16x8 vh = *(lds_frag_ptr)(size_t)(voff[1][0] + (unsigned)(vs * (V_UNITS * 16) + md * 4096));
            const bf16x8 vl = *(lds_frag_ptr)(size_t)(voff[1][1] + (unsigned)(vs * (V_UNITS * 16) + md * 4096));
#pragma unroll
            for (int nq = 0; nq < 2; ++nq) {
                o[md][nq] = VH_MFMA16(vl, PB[nq][0], o[md][nq]);
                o[md][nq] = VH_MFMA16(vh, PB[nq][1], o[md][nq]);
                o[md][nq] = VH_MFMA16(vh, PB[nq][0], o[md][nq]);
            }
        }
    }
#ifdef VH_CLOCK
    {
        const unsigned long long ck_m1 = __builtin_amdgcn_s_memtime(), ck_r1 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (a.dbg && t == 0) { a.dbg[(size_t)blockIdx.x * 2] = ck_m1 - ck_m0; a.dbg[(size_t)blockIdx.x * 2 + 1] = ck_r1 - ck_r0; }
    }
#endif

    // output: lane (li, g) holds channels 16 md + 4 g .. +3 of queries qrow0 and qrow0 + 16
#pragma unroll
    for (int nq = 0; nq < 2; ++nq) {
        float ls = lsum[nq];
        ls += __shfl_xor(ls, 16);
        ls += __shfl_xor(ls, 32);
        const float inv = 1.0f / (ls + a.n_zero);
        const int qrow = qrow0 + 16 * nq;
        if (qrow >= a.s) continue;
        if (a.out_s8) {
            // S8: per 8 channels [hi x8 | lo x8] bf16; this lane owns half a chunk (4 channels): two 8-byte stores
            unsigned short* op = reinterpret_cast<unsigned short*>(a.out) + (((size_t)b * a.s + qrow) * a.c + hd * D) * 2;
#pragma unroll
            for (int md = 0; md < MD; ++md) {
                unsigned h[4], lo4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = o[md][nq][r] * inv;
                    h[r] = bf16_rn_bits(v);
                    lo4[r] = bf16_rn_bits(v - __uint_as_float(h[r] << 16));
                }
                unsigned short* q8 = op + (md * 16 + 8 * (g >> 1)) * 2 + 4 * (g & 1);
                *reinterpret_cast<uint2*>(q8) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
                *reinterpret_cast<uint2*>(q8 + 8) = make_uint2(lo4[0] | (lo4[1] << 16), lo4[2] | (lo4[3] << 16));
            }
        } else {
            float* op = a.out + ((size_t)b * a.s + qrow) * a.c + hd * D;
#pragma unroll
            for (int md = 0; md < MD; ++md) {
                float4 w;
                w.x = o[md][nq][0] * inv; w.y = o[md][nq][1] * inv; w.z = o[md][nq][2] * inv; w.w = o[md][nq][3] * inv;
                *reinterpret_cast<float4*>(op + md * 16 + 4 * g) = w;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// q/k/v split + head norm writing the operand formats above (view/normalize/unbind of
// training/models.py:192-194, :279-293; sequence concat :296-297 via koff).
// One workgroup = one (row, head) x 64 consecutive pixels.
struct SplitXK {
    const float* in; int rows, s, heads, nj, rows_per_b, koff, kl, klp; float qscale;
    float* q; unsigned short* k; unsigned short* vt;
};

__device__ __forceinline__ int perm16(int key) {      // swap bits 2 and 3
    return (key & ~12) | ((key & 4) << 1) | ((key & 8) >> 1);
}

// All global traffic is 12/8-byte-per-lane contiguous loads (a lane's q,k,v or k,v of one channel) and 16-byte stores:
// K rows (S8) and V^T rows are assembled in LDS and written as whole units, Q is one 256-byte row per wave store.
template <int D, int NJ>
__global__ __launch_bounds__(256) void qkv_split_x3_k(const SplitXK a) {
    constexpr int PG = 256 / D;                  // pixels handled per pass
    __shared__ unsigned short sv[D * 2 * 64];    // V^T: [d][hl][64 local positions]
    __shared__ unsigned short sk[64 * D * 2];    // K:   [pixel][D/8][hi8|lo8]
    const int t = threadIdx.x;
    const int d = t % D, g = t / D;
    const int ntile = (a.s + 63) / 64;
    const int tileid = blockIdx.x % ntile;
    const int rh = blockIdx.x / ntile;
    const int head = rh % a.heads, row = rh / a.heads;
    const int bb = row / a.rows_per_b, seg = row - bb * a.rows_per_b;
    const size_t bhq = (size_t)bb * a.heads + head;
    const int s0 = tileid * 64;
    const int kbase = a.koff + seg * a.s;          // key index of pixel 0 of this row
    // fast path: the tile maps onto whole 16-key groups of this row's key range (V^T transposed through LDS)
    const bool fast = (a.s % 16 == 0) && (((kbase + s0) & 15) == 0);
    const float rsd = rsqrtf((float)D);
    // all of this thread's loads first (one dwordx3 / dwordx2 per lane and pixel, 64 / PG of them in flight: with one
    // load per loop iteration the kernel sat at 2.8 TB/s, latency-bound)
    constexpr int NIT = 64 / PG;
    float vall[NIT][NJ];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int s = min(s0 + g + it * PG, a.s - 1);
        const float* in = a.in + (((size_t)row * a.s + s) * a.heads * D + (size_t)head * D + d) * NJ;
#pragma unroll
        for (int j = 0; j < NJ; ++j) vall[it][j] = in[j];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int pp = g + it * PG;
        const int s = s0 + pp;
        const bool ok = s < a.s;
        const float* v = vall[it];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float ss = ok ? v[j] * v[j] : 0.f;
#pragma unroll
            for (int o = D / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
            const float y = (ok ? v[j] : 0.f) / (1e-4f + sqrtf(ss) * rsd);
            const bool is_q = (NJ == 3 && j == 0);
            const bool is_k = (NJ == 3) ? (j == 1) : (j == 0);
            if (is_q) {
                if (ok) a.q[(bhq * a.s + s) * D + d] = y * a.qscale;
                continue;
            }
            const unsigned hi = bf16_rn_bits(y);
            const unsigned lo = bf16_rn_bits(y - __uint_as_float(hi << 16));
            const int key = kbase + s;
            if (is_k) {
                unsigned short* kp = sk + (pp * D + (d & ~7)) * 2 + (d & 7);
                kp[0] = (unsigned short)hi;
                kp[8] = (unsigned short)lo;
            } else if (fast) {
                const int lp = perm16(pp);                       // tile start is 16-aligned: permute locally
                sv[(d * 2 + 0) * 64 + lp] = (unsigned short)hi;
                sv[(d * 2 + 1) * 64 + lp] = (unsigned short)lo;
            } else if (ok) {
                const size_t base = (bhq * D + d) * 2;
                a.vt[(base + 0) * a.klp + perm16(key)] = (unsigned short)hi;
                a.vt[(base + 1) * a.klp + perm16(key)] = (unsigned short)lo;
            }
        }
    }
    __syncthreads();
    const int nvalid = min(64, a.s - s0);
    // K: nvalid consecutive keys x D*4 bytes are one contiguous run of the K buffer
    {
        constexpr int UPK = D / 4;                               // 16-byte units per key
        uint4* kdst = reinterpret_cast<uint4*>(a.k + (bhq * a.klp + kbase + s0) * D * 2);
        const uint4* ksrc = reinterpret_cast<const uint4*>(sk);
        for (int idx = t; idx < nvalid * UPK; idx += 256) kdst[idx] = ksrc[idx];
    }
    if (fast) {
        // D*2 rows of 64 positions = 8 units of 16 B each (nvalid is a multiple of 16 here)
        for (int idx = t; idx < D * 2 * 8; idx += 256) {
            const int rowi = idx >> 3, ku = idx & 7;
            if (ku * 8 >= nvalid) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(&sv[rowi * 64 + ku * 8]);
            *reinterpret_cast<uint4*>(a.vt + (bhq * D * 2 + rowi) * a.klp + kbase + s0 + ku * 8) = v;
        }
    }
}

}  // namespace

extern "C" int vh_qkv_split_x3(vh_ctx* ctx, const vh_qkv_split_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_qkv_split_x3: null args");
    const vh_qkv_split_args a = *p;
    VH_REQUIRE(a.in && a.k && a.v, "vh_qkv_split_x3: null tensor");
    VH_REQUIRE(a.nj == 2 || (a.nj == 3 && a.q), "vh_qkv_split_x3: nj must be 2 or 3 (3 needs q)");
    VH_REQUIRE(a.d == 32 || a.d == 64, "vh_qkv_split_x3: head dim %d unsupported (32 or 64)", a.d);
    VH_REQUIRE(a.rows > 0 && a.s > 0 && a.heads > 0 && a.rows_per_b > 0 && a.rows % a.rows_per_b == 0, "vh_qkv_split_x3: bad geometry");
    VH_REQUIRE(a.koff >= 0 && a.koff + a.rows_per_b * a.s <= a.kl, "vh_qkv_split_x3: keys do not fit");
    VH_REQUIRE(vh_aligned16(a.k) && vh_aligned16(a.v), "vh_qkv_split_x3: pointers must be 16-byte aligned");
    const int klp = (a.kl + KT - 1) / KT * KT;
    SplitXK k{a.in, a.rows, a.s, a.heads, a.nj, a.rows_per_b, a.koff, a.kl, klp, a.qscale, a.q,
              static_cast<unsigned short*>(static_cast<void*>(a.k)), static_cast<unsigned short*>(static_cast<void*>(a.v))};
    const long long nblk = (long long)a.rows * a.heads * ((a.s + 63) / 64);
    VH_REQUIRE(nblk < (1LL << 31), "vh_qkv_split_x3: grid too large");
    const int d = a.d;
    const double bytes = 8.0 * (double)a.rows * a.s * a.heads * a.d * a.nj;
    return vh_dispatch(ctx, VH_TAG_QKVSPLIT, 0.0, bytes, [k, d, nblk](hipStream_t s) -> int {
        if (d == 64 && k.nj == 3) hipLaunchKernelGGL((qkv_split_x3_k<64, 3>), dim3((unsigned)nblk), dim3(256), 0, s, k);
        else if (d == 64) hipLaunchKernelGGL((qkv_split_x3_k<64, 2>), dim3((unsigned)nblk), dim3(256), 0, s, k);
        else if (k.nj == 3) hipLaunchKernelGGL((qkv_split_x3_k<32, 3>), dim3((unsigned)nblk), dim3(256), 0, s, k);
        else hipLaunchKernelGGL((qkv_split_x3_k<32, 2>), dim3((unsigned)nblk), dim3(256), 0, s, k);
        return vh_check_launch("qkv_split_x3_k");
    });
}

int vh_diag_attn() { return VH_DIAG_FLAG; }

extern "C" int vh_attention_x3(vh_ctx* ctx, const vh_attention_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_attention_x3: null args");
    const vh_attention_args a = *p;
    VH_REQUIRE(a.q && a.k && a.v && a.out, "vh_attention_x3: null tensor");
    VH_REQUIRE(a.d == 32 || a.d == 64, "vh_attention_x3: head dim %d unsupported (32 or 64)", a.d);
    VH_REQUIRE(a.b > 0 && a.heads > 0 && a.s > 0 && a.kl > 0, "vh_attention_x3: bad geometry");
    VH_REQUIRE(vh_aligned16(a.q) && vh_aligned16(a.k) && vh_aligned16(a.v) && vh_aligned16(a.out), "vh_attention_x3: pointers must be 16-byte aligned");
    VH_REQUIRE((long long)a.b * a.heads < 65536, "vh_attention_x3: b*heads too large for grid.y");
    VH_REQUIRE(a.n_zero_keys >= 0.f, "vh_attention_x3: negative n_zero_keys");
    const int klp = (a.kl + KT - 1) / KT * KT;
    AttnXK k{a.q, static_cast<const uint4*>(static_cast<const void*>(a.k)), static_cast<const uint4*>(static_cast<const void*>(a.v)),
             a.out, a.heads, a.s, a.kl, klp, a.heads * a.d, a.n_zero_keys, a.out_s8, 0, 0, vh_knob(VH_KNOB_ATTN_XCD), vh_debug_ptr()};
    VH_REQUIRE(!a.out_s8 || (a.heads * a.d) % 32 == 0, "vh_attention_x3: S8 output needs heads*d %% 32 == 0");
    const int d = a.d;
    // One 8-wave workgroup per CU for long sequences (two independent 4-wave workgroups per CU measured 2.4x
    // slower: every workgroup stages its own copy of the K/V stream).
    // (64-channel heads always take the 8-wave form: its 4-wave instantiation needs more than 256 VGPRs and spills)
    const int nw = (a.s > 128 || a.d == 64) ? 8 : 4;
    const bool use_pipe = vh_knob(VH_KNOB_ATTN_PIPE) != 0;
    const bool pipe = use_pipe && nw == 8 && a.kl > KT;
    VH_REQUIRE(a.logit_bound >= 0.f, "vh_attention_x3: negative logit_bound");
    const bool nomax_on = vh_knob(VH_KNOB_ATTN_NOMAX) != 0;
    const bool nomax = nomax_on && a.logit_bound > 0.f && a.logit_bound <= 64.f;
    k.nx = (a.s + nw * 32 - 1) / (nw * 32);
    k.ny = a.b * a.heads;
    VH_REQUIRE((long long)k.nx * k.ny < (1LL << 31), "vh_attention_x3: grid too large");
    const dim3 grid((unsigned)(k.nx * k.ny));
    const double bhd = (double)a.b * a.heads;
    const double flops = 4.0 * bhd * a.s * a.kl * a.d;
    const double bytes = 4.0 * bhd * a.d * (2.0 * a.s + 2.0 * a.kl);
    const bool m16 = vh_knob(VH_KNOB_ATTN_M16) != 0;        // knob "attn_m16" = 0: the 32x32x16 placed kernel (A/B)
    return vh_dispatch(ctx, VH_TAG_ATTN, flops, bytes, [k, d, nw, pipe, nomax, m16, grid](hipStream_t s) -> int {
        if (pipe && d == 64 && nomax && m16) hipLaunchKernelGGL(attn_fwd_x3_m16<64>, grid, dim3(512), 0, s, k);
        else if (pipe && nomax && m16) hipLaunchKernelGGL(attn_fwd_x3_m16<32>, grid, dim3(512), 0, s, k);
        else if (pipe && d == 64 && nomax) hipLaunchKernelGGL((attn_fwd_bf16x3_pipe<64, true>), grid, dim3(512), 0, s, k);
        else if (pipe && d == 64) hipLaunchKernelGGL((attn_fwd_bf16x3_pipe<64>), grid, dim3(512), 0, s, k);
        else if (pipe && nomax) hipLaunchKernelGGL((attn_fwd_bf16x3_pipe<32, true>), grid, dim3(512), 0, s, k);
        else if (pipe) hipLaunchKernelGGL((attn_fwd_bf16x3_pipe<32>), grid, dim3(512), 0, s, k);
        else if (d == 64 && nw == 8) hipLaunchKernelGGL((attn_fwd_bf16x3<64, 8>), grid, dim3(512), 0, s, k);
        else if (nw == 8) hipLaunchKernelGGL((attn_fwd_bf16x3<32, 8>), grid, dim3(512), 0, s, k);
        else hipLaunchKernelGGL((attn_fwd_bf16x3<32, 4>), grid, dim3(256), 0, s, k);
        return vh_check_launch("attn_fwd_bf16x3");
    });
}
