// bf16x3 implicit-GEMM convolution, 256-pixel tiles, direct-to-LDS staging (gfx950).
//
// Same operation and epilogues as conv_igemm.hip (reference training/models.py:123-126 with the fused
// element-wise ops listed there); this is the throughput kernel of the bf16x3 mode.  What changed and why
// (numbers from profiles/: the 128x128 register-staged kernel sits at ~34 % of the bf16 MFMA peak with its
// LDS store path (ds_write_b128, ~79 B/clk/CU) and address VALU as busy as the matrix pipe):
//   * one workgroup = 8 waves computes 256 pixels x 256 (or 128) output channels, each wave 128x64 (or 64x64):
//     half the LDS bytes and staging instructions per MFMA;
//   * operand tiles go global -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no ds_write); the
//     LDS image is [row][8 units] with unit' = unit ^ ((row>>1)&7), applied on the SOURCE side (the DMA writes
//     64 consecutive 16-byte slots per wave instruction; a lane picks which unit of which row it fetches), so
//     8 lanes still read one 128-byte line of one pixel and ds_read_b128 of 32 consecutive rows is conflict-free;
//   * weights are laid out [cout][K] in the same S8 chunking as activations, so B tiles stage exactly like A.
// K runs tap-major in 32-channel tiles; per tap each thread derives its 4 pixel pointers (zero page outside
// the image).  Two LDS stages; the loads of tile t+1 are in flight during the MFMAs of tile t.
#include "conv_common.h"
#include <algorithm>
#include <type_traits>

// The Makefile builds this file twice: VH_CONV_TU=9 (the 3x3 kernels, the split-K reducer and the dispatcher; compiled with
// -mllvm -amdgpu-sched-strategy=iterative-ilp, +2.0..2.8 % on the 3x3 shapes of C2) and VH_CONV_TU=1 (the 1x1 kernels, default
// scheduler: iterative-ilp costs them 2 %).  0 = everything in one object (tools' variant builds).
#ifndef VH_CONV_TU
#define VH_CONV_TU 0
#endif

#ifndef VH_EPI_PD
#define VH_EPI_PD 1          // epilogue blocks whose residual / cvec values are in flight ahead of the one being written (3 measured the same)
#endif

namespace {
using namespace vhconv;

constexpr int BK = 32;

// 16-byte direct global -> LDS load (LDS-DMA): every lane supplies its own global source, the LDS destination is
// the wave-uniform `lds_wave_base` + lane*16.  Written as inline asm so that hipcc does not count it: with the
// builtin the compiler drains the DMA (s_waitcnt vmcnt(0)) before the first ds_read that follows, which would
// serialise load and compute; here the kernel waits for it by hand, once per K-tile, just before the barrier
// that publishes the stage (cdna_hip_programming.md 5.7: M0 carries the LDS address and is saved/restored inside
// the statement).  Compiled in the device pass only (a templated __global__ function that contains the builtin
// or this asm directly gets no host launch stub from hipcc).
__device__ __forceinline__ void glds16(const float4* gsrc, unsigned lds_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    // lds_addr: wave-uniform LDS byte address, already in a scalar register (the caller derives it from one readfirstlane per
    // kernel); M0 is declared clobbered instead of saved and restored (nothing else in these kernels uses it)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_addr) : "memory", "m0");
#endif
}

__device__ __forceinline__ void wait_dma() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// M16 = true uses v_mfma_f32_16x16x32_bf16 (one MFMA covers the whole 32-channel K-tile) instead of two
// 16-deep v_mfma_f32_32x32x16_bf16 slabs: same cycles per FLOP, but the chip holds a higher clock on the 16x16
// shape (MI355X_MICROARCH.md, DVFS give-back item 7).  The LDS unit order is then hl*4 + chunk (conflict-free
// for the 16-row operand reads) instead of chunk*2 + hl; only the source-side mapping of the DMA changes.
// CHUNK = true: channel-chunk-major K order (9 taps only).  A template parameter rather than a run-time flag so that each
// instantiation keeps only ITS per-slot state in registers (pixel coordinates for tap-major, centre pointer + tap mask for
// chunk-major): with both live the 512x128 tile needed 256 VGPRs + 40 bytes of scratch per lane.
// OCC = waves per SIMD the register budget is set for: 2 (<= 256 VGPRs, one workgroup per CU) for the big tiles; 4 (<= 128 VGPRs) for the
// 256x64 tile, whose 80 KB of LDS fit twice per CU - two independent workgroups, so that one's prologue / epilogue / turnaround runs
// under the other's K loop (the full-resolution 64-channel layers of the SR net have K loops of 18 K-tiles: most of a 512x64 tile's time
// was outside its loop).
// TAIL = true: the K loop continues, after the nine taps, over a 1-tap segment read from a second S8 source (vh_conv_args.src1: conv_res1 +
// conv_skip of a decoder block as one GEMM).  A template parameter so that the plain instantiations keep their register allocation (the
// 512x128 tile sits 2-20 registers under the limit); the tail's pointers are derived from state that is live in the loop anyway.
// (a, b) -> packed bf16 hi pair and lo pair of the hi + lo split: one v_cvt_pk_bf16_f32 per pair (the same round-to-nearest-even values as two
// bf16_rn_bits calls each, without their shifts and ors)
__device__ __forceinline__ void bf16_split_pair(float a, float b, unsigned& H, unsigned& L) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t h; h[0] = (__bf16)a; h[1] = (__bf16)b;
    H = __builtin_bit_cast(unsigned, h);
    bf16x2_t q; q[0] = (__bf16)(a - __uint_as_float(H << 16)); q[1] = (__bf16)(b - __uint_as_float(H & 0xFFFF0000u));
    L = __builtin_bit_cast(unsigned, q);
}

// x as seen through a DPP control word (quad_perm 0x00-0xFF, row_half_mirror 0x141, row_mirror 0x140): lanes exchange inside their row of 16
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}

template <int TAPS, int WAVES_M, int WAVES_N, int MI, int NI, bool M16, bool CHUNK, int OCC = 2, bool TAIL = false>
__global__ __launch_bounds__(512, OCC) void conv_x3_glds(const ConvK a) {
    static_assert(!TAIL || TAPS == 9, "the tail segment exists for 3x3 convolutions only");
    static_assert(!CHUNK || TAPS == 9, "chunk-major order exists for 3x3 convolutions only");
    static_assert(WAVES_M * WAVES_N == 8, "8 waves");
    constexpr int BM = WAVES_M * MI * 32, BN = WAVES_N * NI * 32;
    constexpr int RA = BM / 64, RB = BN / 64;             // glds rounds: 512 slots (64 rows x 8 units) per round
    __shared__ float4 sA[2][BM * 8];
    __shared__ float4 sB[2][BN * 8];

#ifdef VH_CLOCK
    const unsigned long long ck_e0 = __builtin_amdgcn_s_memtime();
#endif
    const int t = threadIdx.x;
    const int w = t >> 6, l = t & 63, lr = l & 31, hh = l >> 5;
    const unsigned tile0 = xcd_tile_id();
    const unsigned ntiles_mn = gridDim.x / a.ksplit;
    const int ks = tile0 / ntiles_mn;                      // K slice of this workgroup (split-K; 0 when ksplit == 1)
    const unsigned tile = tile0 - ks * ntiles_mn;
    const int nt = tile % a.NT, mt = tile / a.NT;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- staging maps ---------------------------------------------------------------------------------
    // slot (round j, wave w, lane l) = j*512 + w*64 + l  ->  row = slot>>3 = j*64 + w*8 + (l>>3), unit' = l&7.
    // (row>>1)&7 = ((w&1)<<2) | (l>>4) for every round, so a thread fetches the same unit u of RA (RB) rows.
    const int rsub = w * 8 + (l >> 3);
    const int uslot = (l & 7) ^ (((w & 1) << 2) | (l >> 4));          // logical unit held by this lane's LDS slot
    const int u = M16 ? ((uslot & 3) * 2 + (uslot >> 2)) : uslot;      // ... and where it sits in the S8 row
    const int Hs = a.up ? (a.h >> 1) : a.h, Ws = a.up ? (a.w >> 1) : a.w;
    int py[RA], px[RA], pbase[RA];
    bool pv[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
        const int gm = m0 + j * 64 + rsub;
        pv[j] = gm < a.M;
        const int g = pv[j] ? gm : 0;
        if (TAPS == 1 && !a.up) {                           // a 1x1 tap reads pixel g itself: no (image, y, x) split (two integer divisions per slot)
            py[j] = 0; px[j] = g; pbase[j] = 0;
            continue;
        }
        const int img = fastdiv(g, a.div_hw);
        const int rem = g - img * a.HW;
        py[j] = fastdiv(rem, a.div_w);
        px[j] = rem - py[j] * a.w;
        pbase[j] = img * Hs * Ws;
    }
    const float4* pa[RA];                                  // per-tap A pointers (already + unit u)
    const float4* pb[RB];                                  // B row pointers (+ unit u); the K offset is added per tile
    const float4* zp = reinterpret_cast<const float4*>(a.zeros);
    const int KU = a.k_pad >> 2;                           // 16-byte units per weight row
    bool bzero[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int gn = n0 + j * 64 + rsub;
        bzero[j] = gn >= a.cout;
        pb[j] = a.wt + (size_t)(bzero[j] ? 0 : gn) * KU + u;
    }

    // chunk-major K order (a.korder): the tap changes every K-tile, so instead of re-deriving coordinates the kernel keeps,
    // per slot, the pointer of the centre pixel and a 9-bit mask of the taps that fall inside the image; a tap's pointer is
    // then centre + (dy*W + dx)*C (one scalar offset) or the zero page.  Not for `up` (the source offset is not uniform).
    const float4* pc[RA];
    unsigned pmask[RA];
    if constexpr (CHUNK) {
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            // tap (dy, dx) is inside the image iff row py+dy and column px+dx are: 3 row bits x 3 column bits instead of 9 x 4 compares
            const unsigned xm = (px[j] > 0 ? 1u : 0u) | 2u | (px[j] < a.w - 1 ? 4u : 0u);
            const unsigned ym = (py[j] > 0 ? 1u : 0u) | 2u | (py[j] < a.h - 1 ? 4u : 0u);
            const unsigned m = ((ym & 1u) ? xm : 0u) | ((ym & 2u) ? xm << 3 : 0u) | ((ym & 4u) ? xm << 6 : 0u);
            pmask[j] = pv[j] ? m : 0u;
            pc[j] = reinterpret_cast<const float4*>(a.src0 + (size_t)(pbase[j] + py[j] * a.w + px[j]) * a.c0) + u;
        }
    }
    auto setup_tap_fast = [&](int tap) {
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const long long off = (long long)(dy * a.w + dx) * (a.c0 >> 2);       // in 16-byte units, wave-uniform
#pragma unroll
        for (int j = 0; j < RA; ++j) pa[j] = ((pmask[j] >> tap) & 1u) ? pc[j] + off : zp;
    };
    // "tap 9" = the 1-tap tail segment of a fused (3x3 + 1x1) convolution: K-tiles 9*cin_pad/32 ... read the SECOND source (c1 channels,
    // same pixels, no halo) against the weight columns appended behind the nine taps.  Without `up` the pixel index of row g is g itself.
    // (its slot coordinates are recomputed from the lane id (mbcnt) and the wave id (kept in an SGPR) instead of being kept in VGPRs across
    //  the main loop, where the 512x128 tile has none to spare)
    const int w_s = __builtin_amdgcn_readfirstlane(w);
    auto setup_tail = [&]() {
        if constexpr (TAIL) {
            const int l2 = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            const int rsub2 = w_s * 8 + (l2 >> 3);
            const int uslot2 = (l2 & 7) ^ (((w_s & 1) << 2) | (l2 >> 4));
            const int u2 = M16 ? ((uslot2 & 3) * 2 + (uslot2 >> 2)) : uslot2;
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                const int gm = m0 + j * 64 + rsub2;
                pa[j] = gm < a.M ? reinterpret_cast<const float4*>(a.src1 + (size_t)gm * a.c1) + u2 : zp;
            }
        }
    };
    auto setup_tap = [&](int tap) {
        if (TAIL && tap == 9) { setup_tail(); return; }
        int dy = 0, dx = 0;
        if (TAPS == 9) {
            const int ty = tap / 3;
            dy = ty - 1;
            dx = tap - ty * 3 - 1;
        }
        if (TAPS == 1 && !a.up) {
#pragma unroll
            for (int j = 0; j < RA; ++j) pa[j] = pv[j] ? reinterpret_cast<const float4*>(a.src0 + (size_t)px[j] * a.c0) + u : zp;
            return;
        }
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int yy = py[j] + dy, xx = px[j] + dx;
            const bool ok = pv[j] && (unsigned)yy < (unsigned)a.h && (unsigned)xx < (unsigned)a.w;
            const size_t pix = (size_t)(pbase[j] + (yy >> a.up) * Ws + (xx >> a.up));
            pa[j] = ok ? reinterpret_cast<const float4*>(a.src0 + pix * a.c0) + u : zp;
        }
    };

    // this wave's LDS destinations as scalar byte addresses (one readfirstlane each; the stage and round offsets are scalar adds)
    unsigned ldsA_w = 0, ldsB_w = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    ldsA_w = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)&sA[0][w * 64]);
    ldsB_w = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)&sB[0][w * 64]);
#endif
    // issue the DMA of one K-tile (cc = channel offset inside the tap) into LDS stage `st`
    auto issue = [&](int st, int tap, int cc) {
        const int cu = cc >> 2;                            // channel offset in 16-byte units
        const int bu = (tap * a.cin_pad + cc) >> 2;        // K offset of the weight tile, same units
        const unsigned la = ldsA_w + (unsigned)st * (unsigned)(BM * 8 * 16), lb = ldsB_w + (unsigned)st * (unsigned)(BN * 8 * 16);
#pragma unroll
        for (int j = 0; j < RA; ++j)
            glds16(pa[j] + cu, la + j * 8192u);            // LDS slot (round j, wave w): float4 index j*512 + w*64
#pragma unroll
        for (int j = 0; j < RB; ++j)
            glds16(bzero[j] ? zp : pb[j] + bu, lb + j * 8192u);
    };

    f32x16 acc[M16 ? 1 : MI][M16 ? 1 : NI];
    f32x4 acc16[M16 ? MI * 2 : 1][M16 ? NI * 2 : 1];
    if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < MI * 2; ++i)
#pragma unroll
            for (int j = 0; j < NI * 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }

    const int wm = w / WAVES_N, wn = w % WAVES_N;
    const int swz = (lr >> 1) & 7;                         // (row>>1)&7 of every row this lane reads
    const int arow = (wm * MI * 32 + lr) * 8, brow = (wn * NI * 32 + lr) * 8;
    // 16x16x32: lane (row = l&15, kg = l>>4) supplies k = 8kg..8kg+7 = chunk kg of the K-tile
    const int l15 = l & 15, kg = l >> 4, swz16 = l15 >> 1;
    const int arow16 = (wm * MI * 32 + l15) * 8, brow16 = (wn * NI * 32 + l15) * 8;
    const int u16h = kg ^ swz16, u16l = (4 + kg) ^ swz16;

    // `mid`: work placed between the two A batches (the staggered waves issue their DMA there)
    auto compute = [&](int st, auto&& mid) {
        if constexpr (M16) {
            bf16x8 bh[NI * 2], bl[NI * 2];
#pragma unroll
            for (int j = 0; j < NI * 2; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8*>(&sB[st][brow16 + j * 128 + u16h]);
                bl[j] = *reinterpret_cast<const bf16x8*>(&sB[st][brow16 + j * 128 + u16l]);
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {             // A fragments in two batches: 64 instead of 96 registers
                bf16x8 ah[MI], al[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    ah[i] = *reinterpret_cast<const bf16x8*>(&sA[st][arow16 + (half * MI + i) * 128 + u16h]);
                    al[i] = *reinterpret_cast<const bf16x8*>(&sA[st][arow16 + (half * MI + i) * 128 + u16l]);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI * 2; ++j) {
                        f32x4& c = acc16[half * MI + i][j];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], c, 0, 0, 0);
                    }
                if (half == 0) mid();
            }
        } else {
            mid();
#pragma unroll
            for (int sl = 0; sl < BK / 16; ++sl) {
                // 32x32x16 bf16: lane (row = l&31, h = l>>5) supplies k = 8h..8h+7 of the slab = chunk 2*sl + h
                const int uh = ((sl * 2 + hh) * 2) ^ swz, ul = ((sl * 2 + hh) * 2 + 1) ^ swz;
                bf16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    ah[mi] = *reinterpret_cast<const bf16x8*>(&sA[st][arow + mi * 256 + uh]);
                    al[mi] = *reinterpret_cast<const bf16x8*>(&sA[st][arow + mi * 256 + ul]);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    bh[ni] = *reinterpret_cast<const bf16x8*>(&sB[st][brow + ni * 256 + uh]);
                    bl[ni] = *reinterpret_cast<const bf16x8*>(&sB[st][brow + ni * 256 + ul]);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                    }
            }
        }
    };

    // ---- main loop: DMA of tile t+1 in flight during the MFMAs of tile t; one barrier per tile ------------
    const int KTall = a.k_pad / BK;
    const int kt0 = (int)((long long)KTall * ks / a.ksplit), KT = (int)((long long)KTall * (ks + 1) / a.ksplit) - kt0;
    // K-tile order.  Tap-major (all channel chunks of tap 0, then tap 1, ...) derives the pixel pointers once per tap, but a
    // pixel's 128-byte line comes back for the next tap only after a whole channel sweep of every tile on the XCD:
    // at the full-resolution levels that is more than the 4 MB L2 and each tap re-fetches its input from beyond it (~5x the
    // algorithmic bytes, profiles/).  Chunk-major (the 9 taps of channels 0-31, then of 32-63, ...) re-uses a line within
    // 9 consecutive K-tiles.
    constexpr bool chunk_major = CHUNK;
    int tap, cc;
    const int kt_tail = TAPS * (a.cin_pad / BK);            // first K-tile of the tail segment (== KTall when there is none)
    if (TAIL && kt0 >= kt_tail) { tap = 9; cc = (kt0 - kt_tail) * BK; setup_tail(); }
    else if constexpr (chunk_major) { cc = (kt0 / 9) * BK; tap = kt0 - (kt0 / 9) * 9; setup_tap_fast(tap); }
    else { tap = (kt0 * BK) / a.cin_pad; cc = kt0 * BK - tap * a.cin_pad; setup_tap(tap); }
#ifdef VH_CLOCK
    const unsigned long long ck_p1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    issue(0, tap, cc);
    wait_dma();
#ifdef VH_CLOCK
    const unsigned long long ck_p2 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (a.dbg && t == 0) { a.dbg[(size_t)blockIdx.x * 12 + 10] = ck_p1 - ck_e0; a.dbg[(size_t)blockIdx.x * 12 + 11] = ck_p2 - ck_p1; }
#endif
    __syncthreads();
    // Stagger (vh_conv_args.stagger = 1; MI355X_MICROARCH.md "try a stagger"): waves 0-3 fetch tile kt+1 before their MFMAs and their
    // SIMD partners 4-7 between the two halves of theirs, so that one wave's DMA issue runs under the other's matrix work.  Round 1
    // shipped it for the 512x128 tile (+3..6 % there with the DMA issue as it then was: ~120 cycles per piece).  With the lean issue
    // (scalar LDS addresses, no M0 save/restore: ~70 cycles per piece) and the short epilogue, same-device A/B of round 2 has the plain
    // order ahead on every tile shape (512x128: 1.06-1.07x vs 0.97-1.02x staggered; 256-row tiles: staggered -6..-20 %), so the
    // default is off everywhere; the hint stays in the ABI (results are bit-identical either way).
    const bool late = a.stagger && w >= 4;
#ifdef VH_CLOCK   // diagnostic build: shader clock (s_memtime) against the 100 MHz reference (s_memrealtime) around the K loop
    unsigned long long ck_m0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    // TAIL: the slice's K-tiles are [main tiles of the nine taps | tail tiles]; the main loop below is the plain kernel's, bounded by KTm, and
    // the tail tiles run in a second, simpler loop (same pixels, next 32 channels of the second source per tile).  Keeping the tail out of the
    // main loop keeps its register allocation: with the tail's pointer derivation inside it the 512x128 tile spilled in the loop.
    const int KTm = TAIL ? (kt0 >= kt_tail ? 0 : (kt0 + KT <= kt_tail ? KT : kt_tail - kt0)) : KT;
    for (int kt = 0; kt < KTm; ++kt) {
        const int st = kt & 1;
        auto fetch_next = [&]() {
            if (kt + 1 < KTm) {
                if constexpr (chunk_major) {
                    if (++tap == 9) { tap = 0; cc += BK; }
                    setup_tap_fast(tap);
                } else {
                    cc += BK;
                    if (cc >= a.cin_pad) {
                        cc = 0;
                        ++tap;
                        setup_tap(tap);
                    }
                }
                issue(st ^ 1, tap, cc);
            }
        };
        if (!late) fetch_next();
        compute(st, [&]() { if (late) fetch_next(); });
#if !(defined(VH_CONV_ABLATE) && (VH_CONV_ABLATE & 1))      // timing ablation builds (WRONG results): 1 = no wait for the DMA, 2 = no tile barrier
        wait_dma();                                        // this wave's DMA of tile kt+1 has landed ...
#endif
#if !(defined(VH_CONV_ABLATE) && (VH_CONV_ABLATE & 2))
        __syncthreads();                                   // ... and so has every other wave's
#endif
    }
    if constexpr (TAIL) {
        if (KT > KTm) {
            if (KTm > 0) {                                     // (KTm == 0: this slice starts inside the tail and its first tile is already staged)
                tap = 9; cc = 0;
                setup_tail();
                issue(KTm & 1, 9, 0);
                wait_dma();
                __syncthreads();
            }
            for (int kt = KTm; kt < KT; ++kt) {
                const int st = kt & 1;
                if (kt + 1 < KT) { cc += BK; issue(st ^ 1, 9, cc); }
                compute(st, []() {});
                wait_dma();
                __syncthreads();
            }
        }
    }
#ifdef VH_CLOCK
    {
        const unsigned long long ck_m1 = __builtin_amdgcn_s_memtime(), ck_r1 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (a.dbg && t == 0) {
            a.dbg[(size_t)blockIdx.x * 12] = ck_m1 - ck_m0; a.dbg[(size_t)blockIdx.x * 12 + 1] = ck_r1 - ck_r0;
            a.dbg[(size_t)blockIdx.x * 12 + 2] = ck_m0 - ck_e0;             // prologue: entry -> first K-tile
        }
    }
    const unsigned long long ck_l1 = __builtin_amdgcn_s_memtime();
#define VH_CLOCK_EXIT() do { const unsigned long long ck_x = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); \
        if (a.dbg && t == 0) a.dbg[(size_t)blockIdx.x * 12 + 3] = ck_x - ck_l1; } while (0)     /* epilogue: instructions issued (stores not drained) */
#else
#define VH_CLOCK_EXIT()
#endif
    // the loop's last barrier has retired every read of the stages: reuse sA as 8 per-wave transpose patches
    float* patch = reinterpret_cast<float*>(&sA[0][0]) + w * (32 * 36);
    if constexpr (TAPS == 1 && M16 && NI == 2) {
        if (a.epi == VH_EPI_QKV) {
            // Fused q/k/v split (see VH_EPI_QKV in vivid_hip.h).  With 64-channel heads this wave's 64 accumulator columns are one
            // (head, j); with 32-channel heads (the super-resolution UNet) each 32-column half is one.
            // (wave coordinates through one readfirstlane: the block / head / key arithmetic below then runs on the scalar unit)
            const int wq = __builtin_amdgcn_readfirstlane(w), wms = wq / WAVES_N, wns = wq % WAVES_N;
            const int colw = n0 + wns * 64;
            if (colw >= a.cout) return;
            const int D = a.q_d;
            const float rsd = D == 64 ? 0.125f : 0.17677669529663687f;            // 1/sqrt(D)
            int head_[NI], j_[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int sl = D == 64 ? (colw >> 6) : ((colw >> 5) + ni);
                head_[ni] = sl / a.q_nj;
                j_[ni] = sl - head_[ni] * a.q_nj;
            }
            // 1. RMS-normalise every pixel row over the head's channels, in the accumulators (C/D map: column = lane&15 of each of
            //    the 4 column tiles, row = 4*(lane>>4) + r): square-sum over the head's tiles, then over the 16 lanes of the row.
            //    Done per 32-row group right before its blocks go out (short live ranges: no spills in this cold code).
            auto normalise_rows = [&](f32x4& t0, f32x4& t1, f32x4& t2, f32x4& t3) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // the 16 lanes of a row are one DPP row: pairs, quads (quad_perm), halves (row_half_mirror), the row (row_mirror) -
                    // the same sum tree as an xor butterfly, one v_add_f32 with a DPP operand per step instead of an LDS round trip
                    float s0 = t0[r] * t0[r] + t1[r] * t1[r], s1 = t2[r] * t2[r] + t3[r] * t3[r];
                    s0 += dpp_f32<0xB1>(s0); s1 += dpp_f32<0xB1>(s1);
                    s0 += dpp_f32<0x4E>(s0); s1 += dpp_f32<0x4E>(s1);
                    s0 += dpp_f32<0x141>(s0); s1 += dpp_f32<0x141>(s1);
                    s0 += dpp_f32<0x140>(s0); s1 += dpp_f32<0x140>(s1);
                    if (D == 64) s0 = s1 = s0 + s1;
                    // v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the IEEE sequences: a 1e-7 relative change in a scale that the
                    // tests hold to 2e-6 against vh_qkv_split_x3, and ~50 fewer instructions per row in an issue-bound epilogue
                    const float c0 = ((a.q_nj == 3 && j_[0] == 0) ? a.q_scale : 1.f) * __builtin_amdgcn_rcpf(fmaf(__builtin_amdgcn_sqrtf(s0), rsd, 1e-4f));
                    const float c1 = ((a.q_nj == 3 && j_[1] == 0) ? a.q_scale : 1.f) * __builtin_amdgcn_rcpf(fmaf(__builtin_amdgcn_sqrtf(s1), rsd, 1e-4f));
                    t0[r] *= c0; t1[r] *= c0; t2[r] *= c1; t3[r] *= c1;
                }
            };
            // 2. per 32x32 block: through the wave's LDS patch, then rows (q, k) or columns (v^T) of it go out as 16-byte units
            const int S = a.HW;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                normalise_rows(acc16[2 * mi][0], acc16[2 * mi][1], acc16[2 * mi][2], acc16[2 * mi][3]);
                normalise_rows(acc16[2 * mi + 1][0], acc16[2 * mi + 1][1], acc16[2 * mi + 1][2], acc16[2 * mi + 1][3]);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int row0 = m0 + (wms * MI + mi) * 32;
                    if (row0 >= a.M) continue;                              // (M % 32 == 0: blocks are all-in or all-out)
                    if (colw + ni * 32 >= a.cout) continue;                  // (32-channel heads: cout need not fill the wave's 64 columns)
                    const int head = head_[ni], j = j_[ni];
                    const bool is_q = a.q_nj == 3 && j == 0, is_k = a.q_nj == 3 ? j == 1 : j == 0;
                    const int dbase = D == 64 ? ni * 32 : 0;                // first head channel of this block
                    constexpr int LD = 36;
                    const int rowi = fastdiv(row0, a.div_hw), s0 = row0 - rowi * S;   // 32 | S: the block lies inside one image
                    const int bb = rowi / a.q_rows_per_b, seg = rowi - bb * a.q_rows_per_b;
                    const size_t bhq = (size_t)bb * a.q_heads + head;
                    const int key0 = a.q_koff + seg * S + s0;                  // multiple of 16
                    if (is_q || is_k) {
                        {
                            const int c = l & 15, rb = (l >> 4) * 4;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                patch[(rb + r) * LD + c] = acc16[2 * mi][2 * ni][r];
                                patch[(rb + r) * LD + 16 + c] = acc16[2 * mi][2 * ni + 1][r];
                                patch[(16 + rb + r) * LD + c] = acc16[2 * mi + 1][2 * ni][r];
                                patch[(16 + rb + r) * LD + 16 + c] = acc16[2 * mi + 1][2 * ni + 1][r];
                            }
                        }
                        // one 64-bit address per lane and block (scalar block part + this lane's row / channel); the four row groups follow at 8 rows
                        const int cg = l & 7, rsub = l >> 3;
                        const int d0 = dbase + 4 * cg;
                        float* qp = a.q + (bhq * S + s0) * D + (rsub * D + d0);
                        unsigned short* kp = a.qk + (bhq * a.q_klp + key0) * D * 2 + ((rsub * D + (d0 & ~7)) * 2 + (d0 & 7));
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float4 v = *reinterpret_cast<const float4*>(&patch[(rsub + 8 * i) * LD + 4 * cg]);
                            if (is_q) {
                                *reinterpret_cast<float4*>(qp + 8 * i * D) = v;
                            } else {
                                unsigned H0, H1, L0, L1;
                                bf16_split_pair(v.x, v.y, H0, L0);
                                bf16_split_pair(v.z, v.w, H1, L1);
                                *reinterpret_cast<uint2*>(kp + 16 * i * D) = make_uint2(H0, H1);
                                *reinterpret_cast<uint2*>(kp + 16 * i * D + 8) = make_uint2(L0, L1);
                            }
                        }
                    } else {
                        // V^T [bh][d][hl][klp], positions permuted inside 16-key groups (bits 2,3 swapped).  Straight from the
                        // accumulators: in the C/D map of the 16x16 MFMA a lane already holds 4 consecutive keys (rows 4q..4q+3,
                        // q = lane>>4) of one channel (column lane&15), and the permutation keeps those four together at positions
                        // 4*((q&1)*2 + (q>>1)) - one 8-byte store per tile and half, no LDS transpose.
                        const int q4 = l >> 4, qs = ((q4 & 1) << 1) | (q4 >> 1);
                        const size_t klp = (size_t)a.q_klp;
                        unsigned short* vb = a.qv + (bhq * D + dbase) * 2 * klp + key0 + ((size_t)((l & 15) * 2) * klp + 4 * qs);     // (scalar block part + lane part)
#pragma unroll
                        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                            for (int tb = 0; tb < 2; ++tb) {
                                const f32x4 tv = acc16[2 * mi + ta][2 * ni + tb];
                                unsigned H0, H1, L0, L1;
                                bf16_split_pair(tv[0], tv[1], H0, L0);
                                bf16_split_pair(tv[2], tv[3], H1, L1);
                                unsigned short* vp = vb + (size_t)(32 * tb) * klp + 16 * ta;      // channel dbase + 16*tb + (l & 15), keys key0 + 16*ta + 4*qs
                                *reinterpret_cast<uint2*>(vp) = make_uint2(H0, H1);
                                *reinterpret_cast<uint2*>(vp + klp) = make_uint2(L0, L1);
                            }
                    }
                }
            }
            return;
        }
    }
    ConvK e = a;
    if (a.ksplit > 1) {                                    // raw partial sums; vh_conv's reducer applies the epilogue
        e.epi = VH_EPI_STORE;
        e.clip = 0.f;
        e.out = a.scratch + (size_t)ks * a.M * a.cout;
        e.out_s8 = nullptr;
    }
    if constexpr (M16) {
        // the residual / cvec values of block b are requested PD blocks before it is written out (conv_epilogue_prefetch): the
        // fragment registers are dead by now, so a few blocks' worth of values fit
        constexpr int NB = MI * NI;
        constexpr int PD = VH_EPI_PD < NB ? VH_EPI_PD : NB;
        // Blocks wholly inside the output take the branch-free read-out specialised on the epilogue kind (conv_epilogue_block_fast);
        // edge blocks (ragged M or Cout) and unaligned channel counts keep the generic one.  Both are wave-uniform choices.
        const bool fast_ok = (e.cout & 3) == 0 && (e.epi != VH_EPI_SCALE_SILU || (e.cvec_ld & 3) == 0);
        auto run = [&](auto epic) __attribute__((always_inline)) {
            constexpr int EPI = decltype(epic)::value;
            auto inside = [&](int b) { return fast_ok && m0 + (wm * MI + b / NI) * 32 + 32 <= e.M && n0 + (wn * NI + b % NI) * 32 + 32 <= e.cout; };
            auto prefetch = [&](int b) __attribute__((always_inline)) {
                const int r0 = m0 + (wm * MI + b / NI) * 32, c0 = n0 + (wn * NI + b % NI) * 32;
                return inside(b) ? conv_epilogue_prefetch_fast<EPI>(e, r0, c0, l) : conv_epilogue_prefetch(e, r0, c0, l);
            };
            EpiAux ring[PD];
#pragma unroll
            for (int b = 0; b < PD; ++b) ring[b] = prefetch(b);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int mi = b / NI, ni = b % NI;
                const EpiAux cur = ring[b % PD];
                if (b + PD < NB) ring[b % PD] = prefetch(b + PD);
                const int r0 = m0 + (wm * MI + mi) * 32, c0 = n0 + (wn * NI + ni) * 32;
                if (r0 >= e.M || c0 >= e.cout) continue;     // wholly outside (a 64-pixel level in a 256-row tile): nothing to walk
                if (inside(b))
                    conv_epilogue_block_fast<EPI>(e, acc16[2 * mi][2 * ni], acc16[2 * mi][2 * ni + 1], acc16[2 * mi + 1][2 * ni],
                                                  acc16[2 * mi + 1][2 * ni + 1], r0, c0, patch, l, cur);
                else
                    conv_epilogue_tiles16_lds(e, acc16[2 * mi][2 * ni], acc16[2 * mi][2 * ni + 1], acc16[2 * mi + 1][2 * ni],
                                              acc16[2 * mi + 1][2 * ni + 1], r0, c0, patch, l, &cur);
#ifdef VH_CLOCK
                if (a.dbg && t == 0 && b < 6) { const unsigned long long ck_b = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F);
                                                a.dbg[(size_t)blockIdx.x * 12 + 4 + b] = ck_b - ck_l1; }
#endif
            }
        };
        if (e.epi == VH_EPI_MPSUM) run(std::integral_constant<int, VH_EPI_MPSUM>{});
        else if (e.epi == VH_EPI_SCALE_SILU) run(std::integral_constant<int, VH_EPI_SCALE_SILU>{});
        else run(std::integral_constant<int, VH_EPI_STORE>{});
        VH_CLOCK_EXIT();
    } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                conv_epilogue_tile_lds(e, acc[mi][ni], m0 + (wm * MI + mi) * 32, n0 + (wn * NI + ni) * 32, patch, l);
    }
}

#if VH_CONV_TU != 1
// split-K reducer: sums the ksplit partial-sum slabs and applies the epilogue; one thread per 4 output channels
__global__ __launch_bounds__(256) void conv_splitk_reduce(const ConvK a, long long total4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c4 = (a.cout + 3) >> 2;
    const int gm = (int)(i / c4), gn = (int)(i - (long long)gm * c4) * 4;
    float y[4] = {0.f, 0.f, 0.f, 0.f};
    const size_t slab = (size_t)a.M * a.cout;
    const float* p = a.scratch + (size_t)gm * a.cout + gn;
    for (int s = 0; s < a.ksplit; ++s, p += slab) {
        if ((a.cout & 3) == 0) {
            const float4 v = *reinterpret_cast<const float4*>(p);
            y[0] += v.x; y[1] += v.y; y[2] += v.z; y[3] += v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (gn + j < a.cout) y[j] += p[j];
        }
    }
    conv_epilogue_vec4(a, gm, gn, y);
}

#endif

}  // namespace

#define VH_LAUNCH(T, WM, WN, MI_, NI_, CH_) hipLaunchKernelGGL((conv_x3_glds<T, WM, WN, MI_, NI_, true, CH_>), dim3(grid), dim3(512), 0, s, k)
#define VH_LAUNCH_TAIL(WM, WN, MI_, NI_, CH_, OCC_) hipLaunchKernelGGL((conv_x3_glds<9, WM, WN, MI_, NI_, true, CH_, OCC_, true>), dim3(grid), dim3(512), 0, s, k)
#define VH_LAUNCH_OCC4(T, WM, WN, MI_, NI_, CH_) hipLaunchKernelGGL((conv_x3_glds<T, WM, WN, MI_, NI_, true, CH_, 4>), dim3(grid), dim3(512), 0, s, k)
#define VH_LAUNCH_CFG(T, CH_)                                  \
    do {                                                       \
        if (cfg == 5) VH_LAUNCH_OCC4(T, 8, 1, 1, 2, CH_);      \
        else if (cfg == 3) VH_LAUNCH(T, 8, 1, 2, 2, CH_);      \
        else if (cfg == 2) VH_LAUNCH(T, 4, 2, 4, 2, CH_);      \
        else if (cfg == 1) VH_LAUNCH(T, 2, 4, 4, 2, CH_);      \
        else VH_LAUNCH(T, 4, 2, 2, 2, CH_);                    \
    } while (0)

// cfg: 0 = 256x128 tile, 1 = 256x256, 2 = 512x128, 3 = 512x64, 5 = 256x64 (two workgroups per CU), 7 = 256x192 (3x3 only)
#if VH_CONV_TU != 9
void vh_conv_x3_launch_1tap(const vhconv::ConvK& k, int cfg, unsigned grid, hipStream_t s) { VH_LAUNCH_CFG(1, false); }
#else
void vh_conv_x3_launch_1tap(const vhconv::ConvK& k, int cfg, unsigned grid, hipStream_t s);
#endif

#if VH_CONV_TU != 9
int vh_diag_conv1() { return VH_DIAG_FLAG; }
#endif
#if VH_CONV_TU != 1
int vh_diag_conv3() { return VH_DIAG_FLAG; }
void vh_conv_x3_patch_launch(vhconv::ConvK k, hipStream_t s);      // conv_patch.hip
// Patch-resident kernel (conv_patch.hip): does it run these (validated) arguments?  1 yes, 0 no, -1 = VH_TILE_PATCH16 forced on ineligible ones.
int vh_conv_patch_choice(const vh_conv_args& a, long long M, long long* pwgs_out) {
    // (the kernel addresses its inputs through 32-bit offsets in 16-byte units: M * c / 4 < 2^32)
    const bool narrow = a.cout <= 16 && a.epi == VH_EPI_STORE && !a.src1 && !a.out_s8 && !a.sink[0].ptr && !a.sink[1].ptr && a.out;   // UNet.out_conv
    const bool patch_ok = a.taps == 9 && (a.cout % 32 == 0 || narrow) && a.epi != VH_EPI_QKV &&
                          a.prec == VH_PREC_BF16X3 && a.kernel == VH_CONV_GLDS256 &&
                          (double)M * a.c0 / 4.0 < 4294967296.0 && (double)M * a.c1 / 4.0 < 4294967296.0 && (double)M * a.c2 / 4.0 < 4294967296.0;
    if (a.tile == VH_TILE_PATCH16 && !patch_ok) return -1;
    const long long ptiles = (long long)a.rows * ((a.h + 15) / 16) * ((a.w + 15) / 16);
    const bool tailp = a.src1 != nullptr && !a.src_f32;
    // fp32 main-loop sources: the launch pays 6-10 % (64- / 96-channel blocks fetch the next chunk's pieces two K-tiles ahead; 128-channel blocks have
    // no registers for that and load all five at the boundary) and the vh_split pass it replaces costs 2-3x that.  Knob: 2 (default) every block width,
    // 1 the look-ahead widths only, 0 never (forced tiles always convert)
    if (a.src_f32) {
        const int sk = vh_knob(VH_KNOB_CONV_SRC_F32);
        const bool lookahead = a.cout == 64 || (a.cout % 96 == 0 && a.cout % 128 != 0);
        if (sk == 0 || narrow || (sk == 1 && !lookahead && a.tile != VH_TILE_PATCH16)) return a.tile == VH_TILE_PATCH16 && (sk == 0 || narrow) ? -1 : 0;
    }
    const int tf = vh_knob(VH_KNOB_CONV_TAIL_F32);
    if (a.tail_f32 && (tf == 0 || vh_knob(VH_KNOB_CONV_PATCH_TAIL) == 1)) return a.tile == VH_TILE_PATCH16 ? -1 : 0;      // (fp32 tails exist in the wave-private form only)
    const bool wide_ok = !tailp || vh_knob(VH_KNOB_CONV_PATCH_TAIL) != 1;                                           // (the staged tail exists for 64-channel blocks only)
    const bool n96 = a.cout > 64 && wide_ok && a.cout % 96 == 0 && a.cout % 128 != 0;                               // 96-channel blocks (Cout = 192)
    const long long pwgs = narrow ? ptiles : ptiles * (n96 ? a.cout / 96 : (a.cout > 64 && wide_ok) ? (a.cout + 127) / 128 : (a.cout + 63) / 64);      // workgroups the patch kernel would launch
    if (pwgs_out) *pwgs_out = pwgs;
    const int pknob = vh_knob(VH_KNOB_CONV_PATCH);
    // Size rule, from same-device A/B against the tile the rules of vh_conv_x3_glds_dispatch pick (profiles/r04_ab_conv_patch_vs_glds.txt): the patch
    // kernel leads by +20..29 % at Cout = 64, +13..20 % at Cout = 128 (256^2 / 512^2 / 1024^2), +3..10 % at Cout = 256 / 384 down to 64^2, ties at
    // 32^2 x 512 and loses where its 16x16-pixel tiles do not fit the image (16^2: 0.6-0.87x, 8^2: 0.38x), on small grids (no split-K: 0.35-0.67x at
    // 32 tiles); Cout = 192 runs as two 96-channel blocks (+9..20 % over the 256x192 tile).  With a tail segment (wave-private staging, any block
    // width): +15..24 % at Cout = 64, +11..12 % at 128, +9 % at 192, +7 % at 384, ties (1.00-1.02x) at 256 and 512, which stay on conv_x3_glds.
    // An fp32 tail (tail_f32) costs the launch 1.4-4 % and saves vh_split a third of its bytes (10x that): taken wherever the S8 tail is, and at
    // Cout = 256 from 64^2 up, where the convolution then ties with conv_x3_glds' S8 tail (1.00x) and the split saving is the gain.
    const int mres = a.h < a.w ? a.h : a.w;
    // Narrow outputs (Cout <= 16: the 3-channel out_conv) take a 16-column instantiation: on the 256x64 tile the layer is bound by MFMAs on 61 idle columns
    const bool patch_rule = pwgs >= 256 && mres >= 32 && (narrow || a.cout == 64 || a.cout % 128 == 0 || (n96 && vh_knob(VH_KNOB_CONV_PATCH96) != 0)) &&
                            (!tailp || a.cout <= 192 || (a.cout == 384 && wide_ok) || (a.tail_f32 && (tf == 2 || (a.cout == 256 && mres >= 64)))) && (mres >= 64 || a.cout <= 256) &&
                            (!a.up || a.cout <= 384);      // (`up`: +9..10 % at 128 channels, +5..8 % at 192, +1..3 % at 256 / 384, 0.99x at 512)
    return (patch_ok && (a.tile == VH_TILE_PATCH16 || (a.tile == VH_TILE_AUTO && (pknob > 0 || (pknob < 0 && patch_rule))))) ? 1 : 0;
}

// Entry used by vh_conv for prec == VH_PREC_BF16X3 && kernel == VH_CONV_GLDS (arguments already validated).
int vh_conv_x3_glds_dispatch(vh_ctx* ctx, const vh_conv_args& a, ConvK k, double flops, double bytes) {
    const long long M = k.M;
    {
        long long pwgs = 0;
        const int pc = vh_conv_patch_choice(a, M, &pwgs);
        if (pc < 0) return vh_fail(VH_EINVAL, "vh_conv: VH_TILE_PATCH16 needs a 3x3 convolution with cout %% 32 == 0 or a plain fp32 store of cout <= 16 (got taps %d, cout %d)", a.taps, a.cout);
        if ((a.sink[0].ptr || a.sink[1].ptr) && pc != 1)
            return vh_fail(VH_EINVAL, "vh_conv: S8 sinks given, but these arguments do not take the patch-resident kernel (ask vh_conv_takes_patch first)");
        if (a.src_f32 && pc != 1)
            return vh_fail(VH_EINVAL, "vh_conv: src_f32 given, but these arguments do not take the patch-resident kernel (ask vh_conv_takes_patch first)");
        if (a.tail_f32 && pc != 1)
            return vh_fail(VH_EINVAL, "vh_conv: tail_f32 given, but these arguments do not take the patch-resident kernel (ask vh_conv_takes_patch first)");
        if (pc == 1) {
            if (pwgs >= (1LL << 31)) return vh_fail(VH_EINVAL, "vh_conv: grid too large");
            k.ksplit = 1; k.scratch = nullptr; k.korder = 1; k.stagger = vh_knob(VH_KNOB_CONV_PATCH_DELAY); k.dbg = vh_debug_ptr();
            return vh_dispatch(ctx, VH_TAG_CONV3, flops, bytes, [k](hipStream_t s) -> int {
                vh_conv_x3_patch_launch(k, s);
                return vh_check_launch("conv_x3_patch");
            });
        }
    }
    // 256x256 tiles when Cout allows it and the grid still gives every CU (256) a workgroup; otherwise 256x128
    // (round 3, late: from 128 workgroups on when the K loop is long - 216+ K-tiles: half a round of the bigger wave tile then beats a full
    //  round of 256x128 tiles by 3..11 %, growing with K; at 108-144 K-tiles it loses 1..8 %; profiles/r03_ab_conv_tile_choice_small_launches.txt)
    const long long Tw = ((M + 255) / 256) * (a.cout / 256);
    bool wide = a.cout % 256 == 0 && (Tw >= 256 || (Tw >= 128 && a.k_pad / BK >= 216));
    // 512x128 tiles (same 128x64 wave tile as the wide config) when only the narrow N fits and M is large
    // (round 3: also between 256 and 511 such tiles when that is no more rounds of the chip, weighted by the tile's work (2 x a 256x128 tile at
    //  ~0.88 of its time per flop), than the 256x128 tiles would take - 256 tall tiles are ONE round where 512 narrow ones are two: +12..14 % at
    //  32x64x64 pixels, Cout = 128, profiles/r03_ab_conv_tile_choice_small_launches.txt)
    const long long Tt = ((M + 511) / 512) * ((a.cout + 127) / 128), Tn = ((M + 255) / 256) * ((a.cout + 127) / 128);
    bool tall = !wide && (Tt >= 512 || (Tt >= 256 && (double)((Tt + 255) / 256) * 1.76 <= (double)((Tn + 255) / 256)));
    // 512x64 tiles (8 waves of 64x64) for Cout <= 64 at large M - the full-resolution layers of the super-resolution net,
    // where a 128-wide tile would spend half of its MFMAs and B traffic on zero columns
    // (round 3, late: at ANY M - below 512 such tiles the 128-wide tiles this used to fall back to still spend half their MFMAs on zero columns:
    //  256x64 is +7..45 % on 1-4 rows of 256x256 and on the 3-channel output convolutions, profiles/r03_ab_conv_tile_choice_small_launches.txt)
    bool slim = a.cout <= 64;
    if (a.tile != VH_TILE_AUTO) {                          // caller's choice (tests sweep every shape on small problems)
        if (a.tile == VH_TILE_256x256 && a.cout % 256) return vh_fail(VH_EINVAL, "vh_conv: VH_TILE_256x256 needs cout %% 256 == 0 (got %d)", a.cout);
        if ((a.tile == VH_TILE_512x64 || a.tile == VH_TILE_256x64) && a.cout > 64) return vh_fail(VH_EINVAL, "vh_conv: VH_TILE_512x64 / VH_TILE_256x64 need cout <= 64 (got %d)", a.cout);
        wide = a.tile == VH_TILE_256x256; tall = a.tile == VH_TILE_512x128; slim = a.tile == VH_TILE_512x64 || a.tile == VH_TILE_256x64;
    }
    // 256x192 tiles (wave tile 64x96) for 3x3 convolutions whose Cout is a multiple of 192 but not of 128 (the 192-channel level of the
    // super-resolution UNet): a 128-wide tile would run its second column of tiles half empty (256 / 192 = 1.33x the MFMA work)
    const bool n192 = a.taps == 9 && (a.tile == VH_TILE_256x192 ||
                      (a.tile == VH_TILE_AUTO && a.cout % 192 == 0 && !slim && (a.cout % 128 != 0 || (!wide && !tall)) &&
                       ((M + 255) / 256) * (a.cout / 192) >= 64));      // (below a quarter of the chip the launch is latency-bound: more, smaller tiles)
    // (Cout = 384 wherever the pixels are too few for 512-row tiles: at 65536 pixels - the guidance net's 64x64 level in C2 - 256 x 2 tiles of
    //  256x192 are two full rounds of the chip where 256 x 3 tiles of 256x128 are three, +10 %; at 8192-16384 pixels - the 16x16 level of the
    //  reference's base@64 preset at batch 32 - the bigger wave tile alone is worth +6..25 %, growing with K; profiles/r03_ab_conv_256x192_tile.txt,
    //  r03_ab_conv_tile_choice_small_launches.txt)
    if (n192) wide = tall = slim = false;
    // 256x64 with two workgroups per CU instead of 512x64 with one: +7..15 % on every Cout <= 64 layer measured (3x3 with 18-54 K-tiles, 1x1;
    // fp32 or S8 output; profiles/r03_ab_conv_slim2.txt) - one workgroup's prologue, epilogue and turnaround run under the other's K loop.
    // Knob "conv_slim2": -1 (default) always, 0 never (the 512x64 tile stays reachable for A/B and through vh_conv_args.tile).
    const int slim2_knob = vh_knob(VH_KNOB_CONV_SLIM2);
    const bool slim2 = slim && (a.tile == VH_TILE_256x64 || (a.tile == VH_TILE_AUTO && slim2_knob != 0));
    const int BN = n192 ? 192 : wide ? 256 : slim ? 64 : 128, BMt = (slim2 || (slim && a.src1)) ? 256 : (tall || slim) ? 512 : 256;
    const long long MT = (M + BMt - 1) / BMt, NT = (a.cout + BN - 1) / BN;
    if (MT * NT >= (1LL << 31)) return vh_fail(VH_EINVAL, "vh_conv: grid too large");
    k.NT = (int)NT;
    // split-K for grids that leave most of the chip idle (low-resolution levels / small batches): slice the K loop over
    // `ksplit` workgroups per tile, partial sums through a scratch slab, epilogue in a small reducer launch
    const int KTall = a.k_pad / BK;
    int ksplit = 1;
    if (a.scratch && MT * NT < 256 && KTall >= 16 && a.epi != VH_EPI_QKV) {
        // pick the slice count that minimises (rounds of 256 workgroups) x (K per slice), with a small charge per slice for the
        // reducer's extra traffic: e.g. 128 tiles -> 2 slices (one full round), not 3 (a full and a half-empty round)
        // (up to 16 slices of >= 4 K-tiles when the tiles alone would occupy an eighth of the chip or less: the 8x8 / 16x16 levels of the
        //  reference's 64x64 preset at batch 1, where a launch is ~25 us of fixed costs and the K loop is all that can shrink)
        const int smax = MT * NT <= 32 ? (int)std::min<long long>(16, KTall / 4) : (int)std::min<long long>(8, KTall / 8);
        double best = 1e30;
        for (int ks = 1; ks <= smax; ++ks) {
            if ((size_t)ks * (size_t)M * a.cout > a.scratch_floats) break;
            // time ~ rounds of 256 workgroups x (K-tiles per slice + the tile's fixed prologue / epilogue, worth ~14 K-tiles) + a charge per slice for
            // the partial slabs (fitted to profiles/r03_ab_conv_ksplit_rule.txt; without the fixed term 96 tiles took 5 slices in two rounds
            // where 2 slices in one round are 10 % faster)
            const double rounds = (double)((MT * NT * ks + 255) / 256);
            const double cost = rounds * (1.0 / ks + 14.0 / KTall) + 0.01 * (ks - 1);
            if (cost < best - 1e-9) { best = cost; ksplit = ks; }
        }
    }
    if (const int forced = vh_knob(VH_KNOB_CONV_KSPLIT); forced > 0 && a.scratch && a.epi != VH_EPI_QKV &&
        (size_t)forced * (size_t)M * a.cout <= a.scratch_floats && forced <= KTall) ksplit = forced;      // A/B runs only
    k.ksplit = ksplit;
    k.scratch = a.scratch;
    k.dbg = vh_debug_ptr();
    // chunk-major K order when a tap's sweep over the input does not stay in the XCD's 4 MB L2: with tap-major order every tap then
    // re-fetches its input from beyond L2 (the 256 MB Infinity Cache for inputs below ~200 MB, HBM above).  Round 2 switched at 150 MB
    // (measured +7..10 % at 0.5-1 GB inputs, +1.6 % at 0.2 GB, -2 % at 67 MB in isolation); round 3 switches at 60 MB: inside the network
    // the step time is unchanged to 0.1 % (319.0 vs 318.6 ms, profiles/r03_ab_conv_korder_threshold.txt) and the convolution family's
    // fabric traffic drops from 1.54x to 1.26x its algorithmic bytes (1067 -> 873 MB per launch, rocprofv3 FETCH_SIZE / WRITE_SIZE).
    // vh_conv_args.korder overrides the size rule; the process-wide knobs "conv_korder_mb" (threshold) and "conv_korder" (0 tap / 1 chunk,
    // overrides everything) exist for A/B runs.
    const int korder_env = vh_knob(VH_KNOB_CONV_KORDER);
    const bool big_input = (double)M * a.cin_pad * 4.0 > 1e6 * (double)vh_knob(VH_KNOB_CONV_KORDER_MB);
    const int korder_arg = a.korder == VH_KORDER_TAP ? 0 : a.korder == VH_KORDER_CHUNK ? 1 : (big_input ? 1 : 0);
    k.korder = (a.taps == 9 && !a.up && a.cin_pad > BK) ? (korder_env >= 0 ? korder_env : korder_arg) : 0;
    const unsigned grid = (unsigned)(MT * NT * ksplit);
    const int taps = a.taps;
    const bool has_tail = a.src1 != nullptr;               // (validated by vh_conv: bf16x3, 3x3, no `up`)
    const int cfg = n192 ? 7 : (slim2 || (slim && has_tail)) ? 5 : slim ? 3 : tall ? 2 : wide ? 1 : 0;       // (a tail launch with Cout <= 64 always takes the 256x64 tile)
    if (has_tail && (cfg == 1 || cfg == 2) && a.cin_pad > BK) k.korder = 1;
    const int stagger_env = vh_knob(VH_KNOB_CONV_STAGGER);
    k.stagger = stagger_env >= 0 ? stagger_env : (a.stagger == 1 ? 1 : 0);       // default: off (see the kernel's note on `late`)
    const bool chunk = k.korder != 0;
    return vh_dispatch(ctx, taps == 9 ? VH_TAG_CONV3 : VH_TAG_CONV1, flops, bytes, [k, taps, cfg, chunk, grid](hipStream_t s) -> int {
        if (taps == 9 && k.c1 > 0) {
            // tail instantiations: 256x128 and 256x64 in both K orders; 512x128 and 256x256 chunk-major only (the dispatcher forces that
            // order for them: tap-major they are 2 registers over the limit, and no launch the size rule makes would take them tap-major)
            if (cfg == 7) { if (chunk) VH_LAUNCH_TAIL(4, 2, 2, 3, true, 2); else VH_LAUNCH_TAIL(4, 2, 2, 3, false, 2); }
            else if (cfg == 2) VH_LAUNCH_TAIL(4, 2, 4, 2, true, 2);
            else if (cfg == 1) VH_LAUNCH_TAIL(2, 4, 4, 2, true, 2);
            else if (cfg == 5) { if (chunk) VH_LAUNCH_TAIL(8, 1, 1, 2, true, 4); else VH_LAUNCH_TAIL(8, 1, 1, 2, false, 4); }
            else { if (chunk) VH_LAUNCH_TAIL(4, 2, 2, 2, true, 2); else VH_LAUNCH_TAIL(4, 2, 2, 2, false, 2); }
        }
        else if (taps == 9 && cfg == 7) { if (chunk) VH_LAUNCH(9, 4, 2, 2, 3, true); else VH_LAUNCH(9, 4, 2, 2, 3, false); }
        else if (taps == 9) { if (chunk) VH_LAUNCH_CFG(9, true); else VH_LAUNCH_CFG(9, false); }
        else vh_conv_x3_launch_1tap(k, cfg, grid, s);
        if (k.ksplit > 1) {
            const long long total4 = (long long)k.M * ((k.cout + 3) / 4);
            hipLaunchKernelGGL(conv_splitk_reduce, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, k, total4);
        }
        return vh_check_launch("conv_x3_glds");
    });
}
#endif  // VH_CONV_TU != 1
