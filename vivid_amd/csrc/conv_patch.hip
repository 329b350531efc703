// bf16x3 3x3 convolution for Cout <= 64 with the input PATCH resident in LDS (gfx950).
//
// Same operation and epilogues as conv_x3_glds (reference training/models.py:123-126, F.conv2d(x, w, padding=1), with the fused
// y*c + mp_silu :175-176 / mp_sum + clip :184,:204 epilogues); this is the kernel of the full-resolution 64-channel layers of the
// super-resolution UNet (SRXAttnUNet :575-582 at 256x256 / 1024x1024), where conv_x3_glds' 256x64 tile is bound by OPERAND DELIVERY, not
// by the matrix pipe: per 32-channel K-tile it moves 32 KB of activations + 8 KB of weights L2 -> LDS for 768 CU-cycles of MFMA work
// (104 GB/s per CU at full MFMA rate against the ~70 GB/s a CU gathers from L2, MI355X_MICROARCH.md "Indexed rows: gather into LDS"),
// because every input pixel is staged nine times, once per tap.
//
// Here one workgroup owns a 16x16-pixel output tile of one image.  Per 32-channel chunk its (16+2)x(16+2) input patch is staged ONCE
// (41 KB, LDS-DMA) and the nine taps read their A fragments from it in place - a tap is a constant offset into the patch - so the
// activations cross L2 -> LDS 1.27x instead of 9x; the weights stream through two 8 KB stages as before.  58 KB of LDS and <= 128
// VGPRs: two workgroups per CU, one's patch loads / epilogue run under the other's MFMAs.
//   patch image : [pixel p = py*18 + px][8 units of 16 B], unit order hl*4 + chunk, unit' = unit ^ (p & 7) applied on the SOURCE side
//                 of the DMA; with that swizzle the 16-pixel ds_read_b128 fragment reads are conflict-free for all three dx
//                 alignments (brute-forced over the instruction's four 16-lane groups; (p>>1)&7, conv_x3_glds' swizzle, is 2-way
//                 conflicted whenever the row of 16 pixels does not start at a multiple of 4)
//   wave w      : output rows 2w, 2w+1 of the tile (two 16-pixel M tiles) x 64 output channels: 24 MFMAs per K-tile
//   K order     : chunk-major (the nine taps of channels 0-31, then of 32-63, ...) - different from both orders of conv_x3_glds only in
//                 rounding (fp32 accumulation order)
//   tail segment: the 1-tap second source of a fused conv_res1 + conv_skip (vh_conv_args.src1) needs no halo and no reuse across
//                 taps: it runs conv_x3_glds' loop (a 256-pixel A tile per K-tile, two LDS stages) in the space of the patch.  (A first
//                 form loaded its A fragments global -> registers, 32 contiguous bytes per lane: 16 different lines per 16 lanes, paced
//                 by the vector L1's tag rate - +3..10 % where the staged form's main loop gives +20..25 %.)
#include "conv_common.h"
#include <type_traits>

namespace {
using namespace vhconv;

constexpr int PT = 16, PP = PT + 2, PPIX = PP * PP;        // tile edge, patch pitch, patch pixels (324)
constexpr int PSLOTS = 41 * 64;                             // 16-byte LDS slots of the patch: 324*8 = 2592, rounded up to whole wave-instructions (2624)

__device__ __forceinline__ void glds16p(const float4* gsrc, unsigned lds_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_addr) : "memory", "m0");
#endif
}
__device__ __forceinline__ void wait_dma_p() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
// all but the N youngest vector-memory operations of this wave have completed (they complete in issue order)
template <int N>
__device__ __forceinline__ void wait_dma_but() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
#endif
}

// residual (MPSUM) / cvec (SCALE_SILU) values of one 32-pixel x 32-channel block, requested ahead of its read-out
struct PAux { float4 v[4]; float rs[4]; };

// Block = the wave's two image rows (yb, yb+1) x 16 pixels from x0, channels c0..c0+31.  Lane (cg = lane&7, rsub = lane>>3) handles 4
// consecutive channels of pixels rl = rsub + 8i, i = 0..3: image row yb + (i>>1), column x0 + rsub + 8*(i&1).
template <int EPI>
__device__ __forceinline__ PAux patch_prefetch(const ConvK& a, int img, int yb, int x0, int c0, int lane) {
    PAux x;
    const int cg = lane & 7, rsub = lane >> 3, gn = c0 + 4 * cg;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        x.rs[i] = 0.f;
        const int yy = yb + (i >> 1), xx = x0 + rsub + 8 * (i & 1);
        if (EPI == VH_EPI_MPSUM) {
            if (yy < a.h && xx < a.w) {
                // res_up: the residual is the block input of an `up` block, at half resolution, nearest-replicated (resample 'up', training/models.py:60-61)
                const size_t gm = a.res_up ? ((size_t)img * (a.h >> 1) + (yy >> 1)) * (a.w >> 1) + (xx >> 1) : ((size_t)img * a.h + yy) * a.w + xx;
                x.v[i] = *reinterpret_cast<const float4*>(a.res + gm * a.cout + gn);
                x.rs[i] = a.res_scale ? a.ta * a.res_scale[gm] : a.ta;
            }
        } else if (EPI == VH_EPI_SCALE_SILU) {
            if (i == 0) x.v[0] = *reinterpret_cast<const float4*>(a.cvec + (size_t)img * a.cvec_ld + gn);      // one image per tile: one row of cvec
        }
    }
    return x;
}

template <int EPI>
__device__ __forceinline__ void patch_block_out(const ConvK& a, const f32x4 t00, const f32x4 t01, const f32x4 t10, const f32x4 t11,
                                                int img, int yb, int x0, int c0, float* patch, int lane, const PAux& aux) {
    constexpr int LD = 36;
    {
        const int c = lane & 15, rb = (lane >> 4) * 4;      // C/D map of the 16x16 MFMA: column = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            patch[(rb + r) * LD + c] = t00[r];
            patch[(rb + r) * LD + 16 + c] = t01[r];
            patch[(16 + rb + r) * LD + c] = t10[r];
            patch[(16 + rb + r) * LD + 16 + c] = t11[r];
        }
    }
    const int cg = lane & 7, rsub = lane >> 3;
    const int sub = 4 * (cg & 1);                             // position of the lane's 4 channels inside their 8-channel S8 chunk
    const bool odd = sub != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(&patch[(rsub + 8 * i) * LD + 4 * cg]);
        float y[4] = {v.x, v.y, v.z, v.w};
        if (EPI == VH_EPI_SCALE_SILU) {
            const float c[4] = {aux.v[0].x, aux.v[0].y, aux.v[0].z, aux.v[0].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = mp_silu_dev(y[j] * c[j]);
        } else if (EPI == VH_EPI_MPSUM) {
            const float rv[4] = {aux.v[i].x, aux.v[i].y, aux.v[i].z, aux.v[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                y[j] = rv[j] * aux.rs[i] + y[j] * a.tb;
                if (a.clip > 0.f) y[j] = fminf(fmaxf(y[j], -a.clip), a.clip);
            }
        } else if (a.clip > 0.f) {
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = fminf(fmaxf(y[j], -a.clip), a.clip);
        }
        const int yy = yb + (i >> 1), xx = x0 + rsub + 8 * (i & 1);
        const bool ok = yy < a.h && xx < a.w;                  // (ragged image edges: the tile hangs over)
        const size_t gm = ((size_t)img * a.h + yy) * a.w + xx;
        const size_t eo = gm * a.cout + c0 + 4 * cg;
        if (a.out && ok) *reinterpret_cast<float4*>(a.out + eo) = make_float4(y[0], y[1], y[2], y[3]);
        // S8 store of the lane's 4 channels = half of one 8-channel chunk [hi x8 | lo x8]: the lane pair (even, odd) swaps one 8-byte piece so that
        // the even lane writes the whole hi half and the odd lane the whole lo half - one 16-byte store each (conv_common.h s8_store_half_chunk)
        auto s8_pair_store = [&](unsigned short* chunk, const float (&v)[4]) __attribute__((always_inline)) {
            unsigned h[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h[j] = bf16_rn_bits(v[j]);
                lo[j] = bf16_rn_bits(v[j] - __uint_as_float(h[j] << 16));
            }
            const unsigned H0 = h[0] | (h[1] << 16), H1 = h[2] | (h[3] << 16), L0 = lo[0] | (lo[1] << 16), L1 = lo[2] | (lo[3] << 16);
            const unsigned r0 = __shfl_xor(odd ? H0 : L0, 1), r1 = __shfl_xor(odd ? H1 : L1, 1);
            if (ok) *reinterpret_cast<uint4*>(chunk + (odd ? 8 : 0)) = odd ? make_uint4(r0, r1, L0, L1) : make_uint4(H0, H1, r0, r1);
        };
        if (a.out_s8) s8_pair_store(a.out_s8 + (eo - sub) * 2, y);
        // sinks (vh_s8_sink): the same values as vh_split would derive from the fp32 result - scale, mp_silu, hi/lo split - into a channel range of a wider tensor
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!a.sk_ptr[k]) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = y[j] * a.sk_scale[k];
                if (a.sk_silu[k]) v[j] = mp_silu_dev(v[j]);
            }
            s8_pair_store(a.sk_ptr[k] + (gm * a.sk_ct[k] + (size_t)(a.sk_off[k] + c0 + 4 * cg - sub)) * 2, v);
        }
    }
}

// a * b rounded to fp32, never contracted into a following add / subtract (HIP's __fmul_rn is a plain `*` that -ffp-contract=fast may fuse:
// the fp32 tail must produce split_k's bits, whose product is rounded before the hi / lo split subtracts from it)
__device__ __forceinline__ float mul_rounded(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}

// (a, b) -> packed bf16 hi pair and lo pair of the hi + lo split (one v_cvt_pk_bf16_f32 per pair; the roundings of split_k / bf16_rn_bits)
__device__ __forceinline__ void bf16_split_pair_p(float a, float b, unsigned& H, unsigned& L) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t h; h[0] = (__bf16)a; h[1] = (__bf16)b;
    H = __builtin_bit_cast(unsigned, h);
    bf16x2_t q; q[0] = (__bf16)(a - __uint_as_float(H << 16)); q[1] = (__bf16)(b - __uint_as_float(H & 0xFFFF0000u));
    L = __builtin_bit_cast(unsigned, q);
}

// PF: the next chunk's patch is requested during this chunk's taps (two pieces by LDS-DMA into a landing pad, three into registers - the
// loop cannot hold all five) and moved into place at the boundary, instead of being fetched there.  Same-device A/B: the boundary itself
// shrinks from 2.0 to 0.45 us (s_memtime stamps), worth +3 % with two chunks (64 input channels: one boundary, short K loop) and -2.5 %
// with four or six (the extra LDS traffic and the address work sit in a K loop that is MFMA-paced while both workgroups of the CU are in
// theirs): the launcher takes it for cin_pad <= 64 only.
// NJ: 16-channel N tiles per wave - 4 (Cout <= 64), 8 (128 channels per workgroup: 64 accumulator registers, 16 KB weight stages, 73 KB of LDS) or
// 6 (96: Cout = 192 as two blocks, where 128-channel blocks would leave the second half empty) or 1 (Cout <= 16, plain store: UNet.out_conv's 3
// channels - on a 64-wide tile 61 of 64 MFMA columns are idle and the layer is MFMA-bound on them; 16 columns leave it to the patch's LDS reads).
// TAIL: 0 no second source; 1 the 1-tap tail through two 32 KB LDS-DMA stages (NJ = 4 only: 80 KB with the weights); 2 through registers and a
// wave-private 4 KB LDS area (any NJ: each wave's 32 pixels are read by that wave alone, so their staging needs no barrier and no second stage);
// 3 = 2 with fp32 sources (vh_conv_args.tail_f32): the registers the pieces pass through anyway are where mp_cat's scale and the bf16 hi / lo split
// are applied, so no raw S8 form of the concat has to exist in memory.
// RS ("register-staged sources", vh_conv_args.src_f32): the main loop's input is not an S8 tensor but the fp32 tensors a producer would have made it
// from - one or two NHWC sources as a channel concat, each times its mp_cat weight, optionally through mp_silu (conv_res0 of a decoder block reads
// mp_silu(mp_cat(x, skip)), training/models.py:78-84, :174).  At every chunk boundary the patch goes global -> registers -> (scale, mp_silu, bf16
// hi / lo split: vh_split's operations and roundings) -> LDS instead of by LDS-DMA: the vh_split pass and its S8 tensor do not exist.
template <int NJ, int TAIL, bool PF, bool RS = false>
__global__ __launch_bounds__(512, 4) void conv_x3_patch(const ConvK a) {
    static_assert(!RS || (TAIL == 0 && !PF && NJ >= 4), "register-staged sources: plain main loop only");
    static_assert(NJ == 1 || NJ == 4 || NJ == 6 || NJ == 8, "16 (narrow outputs: the 3-channel out_conv), 64, 96 or 128 output channels per workgroup");
    static_assert(!(NJ == 1 && TAIL != 0), "the narrow form has no tail segment");
    static_assert(TAIL >= 0 && TAIL <= 3, "tail modes 0-3");
    static_assert(!(TAIL == 1 && NJ > 4), "the tail's two 32 KB stages and 32 KB of weight stages do not fit half a CU's LDS");
    static_assert(!(PF && NJ > 4), "the landing pad of the next chunk's pieces and 32 KB of weight stages do not fit half a CU's LDS");
    constexpr int BN = NJ * 16, RB = (BN + 63) / 64;         // output channels per workgroup; weight DMA pieces per wave and K-tile (stage = RB * 64 rows)
    // One LDS region for activations: the resident patch (PSLOTS 16-byte slots, 41 KB) and, behind it, the landing pad of the next chunk's
    // LDS-DMA pieces (PF); the TAIL variant re-uses the whole region as two 32 KB stages of its 1-tap segment (80 KB with the
    // weights: exactly half a CU's LDS) - and the epilogue as eight per-wave transpose patches.
    constexpr int PXL = 2;                                   // next chunk's pieces 0..PXL-1 (and wave 0's sixth) go through the landing pad, the rest through registers
    constexpr int XSLOTS = PXL * 512 + 64;
    constexpr int ASLOTS = TAIL == 1 ? 4096 : PSLOTS + (PF ? XSLOTS : 0);
    static_assert(PSLOTS + XSLOTS <= 4096, "landing pad must fit behind the patch");
    __shared__ float4 sA[ASLOTS];
    __shared__ float4 sB[2][RB * 64 * 8];                    // weight K-tiles, two stages (16 / 32 KB)
    float4* const sP = sA;
    float4* const sX = sA + PSLOTS;
    (void)sX;

#ifdef VH_CLOCK   // diagnostic build (make variant ... DEFS=-DVH_CLOCK=1; tools/clock_probe.py): s_memtime stamps of wave 0 into vh_debug_ptr()
#define PCK(var) const unsigned long long var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F)
    PCK(ck_e0);
    unsigned long long ck_bsum = 0, ck_ksum = 0;
#endif
    // Desynchronise the two workgroups of a CU (knob "conv_patch_delay", a.stagger = units of 2048 cycles): both start together and would
    // run in lock step - both in their prologue, both in their K loop sharing the matrix pipe, both in their epilogue.  The second
    // workgroup of every CU in the launch's first round (ids 256..511 under round-robin placement; a guess that costs nothing where it is
    // wrong) starts late, so that one's patch loads / stores run under the other's MFMAs from then on.
    if (a.stagger > 0 && blockIdx.x >= 256u && blockIdx.x < 512u)
        for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(32);
    const int t = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), l = t & 63;
    // workgroup -> (pixel tile, block of BN output channels): the N blocks of one pixel tile are neighbours (they stage the same patch: L2)
    const unsigned tile0 = xcd_tile_id();
    const int nb = a.NT > 1 ? (int)(tile0 % (unsigned)a.NT) : 0;
    const unsigned tile = a.NT > 1 ? tile0 / (unsigned)a.NT : tile0;
    const int img = fastdiv((int)tile, a.div_ptiles);
    const int trem = (int)tile - img * (a.pty * a.ptx);
    const int tyi = fastdiv(trem, a.div_ptx), txi = trem - tyi * a.ptx;
    const int y0 = tyi * PT, x0 = txi * PT;
    const float4* zp = reinterpret_cast<const float4*>(a.zeros);

    // ---- patch staging map: slot = j*512 + t -> pixel p = slot>>3 (py = p/18, px = p%18), LDS unit u' = slot&7 holds logical unit u' ^ (p&7)
    // rounds j = 0..4 cover slots 0..2559 with all eight waves; slots 2560..2623 (pixels 320..323 + padding) are wave 0's sixth piece
    constexpr int PR = 6;
    // Source of piece j of this thread for the chunk at 16-byte-unit offset cu: recomputed whenever a chunk is requested (~12 VALU per piece,
    // once per nine K-tiles) instead of held across the K loop - the loop has no registers to spare (<= 128 for two workgroups per CU).
    const float4* src4 = reinterpret_cast<const float4*>(a.src0);
    const unsigned cu4 = (unsigned)(a.c0 >> 2);                        // 16-byte units per pixel
    // `up` (resample(x, mode="up") with the default [1,1] filter fused into the loader, training/models.py:60-61, :167): patch pixel (yy, xx) of the
    // upsampled image is source pixel (yy>>1, xx>>1) of the half-size tensor; zero padding applies at the UPSAMPLED border
    const int Ws = a.up ? (a.w >> 1) : a.w;
    const unsigned img_off = (unsigned)img * (unsigned)(a.up ? (a.HW >> 2) : a.HW);
    auto piece_src = [&](int j, unsigned cu, int tq) __attribute__((always_inline)) -> const float4* {
        // tq == threadIdx.x, passed through an opaque copy by the callers inside the K loop: otherwise hipcc hoists this whole computation
        // out of the loop and keeps its results live (LICM), which is exactly the register cost it is here to avoid
        const int slot = j * 512 + (j == PR - 1 ? (tq & 63) : tq);       // (the sixth piece exists for wave 0 only: its lane id is its slot offset)
        const int p = slot >> 3, up = slot & 7;
        const int py = (p * 3641) >> 16, px = p - py * PP;                  // p / 18 for p < 4096
        const int yy = y0 - 1 + py, xx = x0 - 1 + px;
        const bool inside = p < PPIX && (unsigned)yy < (unsigned)a.h && (unsigned)xx < (unsigned)a.w;
        const int u = up ^ (p & 7), s8u = (u & 3) * 2 + (u >> 2);          // LDS unit order hl*4 + chunk -> S8 order chunk*2 + hl
        // 32-bit offsets in 16-byte units (the dispatcher checks M * c0 / 4 < 2^32)
        const int sy = a.up ? (yy >> 1) : yy, sx = a.up ? (xx >> 1) : xx;
        return inside ? src4 + (size_t)((img_off + (unsigned)(sy * Ws + sx)) * cu4 + (unsigned)s8u + cu) : zp;
    };
    unsigned ldsP_w = 0, ldsB_w = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    ldsP_w = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)&sP[w * 64]);
    ldsB_w = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)&sB[0][w * 64]);
#endif
    auto load_patch = [&](int cu) __attribute__((always_inline)) {          // cu: channel offset in 16-byte units
        int tq = t;
        asm volatile("" : "+v"(tq));
#pragma unroll
        for (int j = 0; j < PR - 1; ++j) glds16p(piece_src(j, (unsigned)cu, tq), ldsP_w + j * 8192u);
        if (w == 0) glds16p(piece_src(PR - 1, (unsigned)cu, tq), ldsP_w + (PR - 1) * 8192u);
    };

    // RS: chunk c of the concat, global fp32 -> registers -> scale / mp_silu / split -> the patch image.  Slot j*512 + t = (patch pixel p, channel quad
    // cq): 16 bytes = channels 4cq .. 4cq+3 of the chunk (8 lanes = the pixel's 128 bytes); hi pairs go to unit cq>>1, half cq&1, lo pairs to unit 4 + (cq>>1)
    // Two halves: fetch (global -> registers; with RSP two K-tiles ahead of the boundary, so the latency runs under taps 7 and 8) and convert (-> LDS).
    constexpr bool RSP = RS && NJ < 8;                         // look-ahead fetch where the K loop has 24 registers to spare (64 / 96-channel blocks)
    f32x4 rr[RSP ? PR : 1];
    (void)rr;
    auto rs_coords = [&](int j, int tq, int& p, int& cq) __attribute__((always_inline)) {
        const int slot = j * 512 + (j == PR - 1 ? (tq & 63) : tq);
        p = slot >> 3; cq = slot & 7;
    };
    auto fetch_f32 = [&](int c) __attribute__((always_inline)) {
        int tq = t;
        asm volatile("" : "+v"(tq));
        const int n0c = a.c0 >> 5;
        const bool first = c < n0c;
        const float4* s4 = reinterpret_cast<const float4*>(first ? a.src0 : a.src1);
        const unsigned cs4 = (unsigned)((first ? a.c0 : a.c1) >> 2), cu = (unsigned)(first ? c : c - n0c) * 8u;
        auto one = [&](int j) __attribute__((always_inline)) {
            int p, cq;
            rs_coords(j, tq, p, cq);
            const int py = (p * 3641) >> 16, px = p - py * PP;
            const int yy = y0 - 1 + py, xx = x0 - 1 + px;
            const bool inside = p < PPIX && (unsigned)yy < (unsigned)a.h && (unsigned)xx < (unsigned)a.w;
            const int sy = a.up ? (yy >> 1) : yy, sx = a.up ? (xx >> 1) : xx;
            rr[j] = *reinterpret_cast<const f32x4*>(inside ? s4 + (size_t)((img_off + (unsigned)(sy * Ws + sx)) * cs4 + (unsigned)cq + cu) : zp);
        };
#pragma unroll
        for (int j = 0; j < PR - 1; ++j) one(j);
        if (w == 0) one(PR - 1);
    };
    auto convert_f32 = [&](int c) __attribute__((always_inline)) {
        int tq = t;
        asm volatile("" : "+v"(tq));
        const float sc = c < (a.c0 >> 5) ? a.scale0 : a.scale1;
        uint2* const sP2 = reinterpret_cast<uint2*>(sP);
        auto one = [&](int j) __attribute__((always_inline)) {
            int p, cq;
            rs_coords(j, tq, p, cq);
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k] = mul_rounded(rr[j][k], sc);
                if (a.pro == VH_PRO_SILU) v[k] = mp_silu_dev(v[k]);
            }
            unsigned H0, L0, H1, L1;
            bf16_split_pair_p(v[0], v[1], H0, L0);
            bf16_split_pair_p(v[2], v[3], H1, L1);
            const int ih = (p * 8 + ((cq >> 1) ^ (p & 7))) * 2 + (cq & 1);       // 8-byte half of hi unit cq>>1 (swizzled p & 7); its lo unit is 4 units on: ^ 8 in halves
            if (p < PPIX + 4) { sP2[ih] = make_uint2(H0, H1); sP2[ih ^ 8] = make_uint2(L0, L1); }
        };
#pragma unroll
        for (int j = 0; j < PR - 1; ++j) one(j);
        if (w == 0) one(PR - 1);
    };
    // without look-ahead (128-channel blocks: no registers to carry the pieces across K-tiles): at the boundary, in groups of RSG pieces - all loads of
    // a group in flight together (one latency per group, not per piece: in-kernel stamps, 5.2 us per boundary piece by piece against 1.8 us by LDS-DMA)
    constexpr int RSG = 5;
    auto stage_f32 = [&](int c) __attribute__((always_inline)) {
        if constexpr (RSP) { fetch_f32(c); convert_f32(c); return; }
        int tq = t;
        asm volatile("" : "+v"(tq));
        const int n0c = a.c0 >> 5;
        const bool first = c < n0c;
        const float4* s4 = reinterpret_cast<const float4*>(first ? a.src0 : a.src1);
        const unsigned cs4 = (unsigned)((first ? a.c0 : a.c1) >> 2), cu = (unsigned)(first ? c : c - n0c) * 8u;
        const float sc = first ? a.scale0 : a.scale1;
        uint2* const sP2 = reinterpret_cast<uint2*>(sP);
        auto load1 = [&](int j) __attribute__((always_inline)) -> f32x4 {
            int p, cq;
            rs_coords(j, tq, p, cq);
            const int py = (p * 3641) >> 16, px = p - py * PP;
            const int yy = y0 - 1 + py, xx = x0 - 1 + px;
            const bool inside = p < PPIX && (unsigned)yy < (unsigned)a.h && (unsigned)xx < (unsigned)a.w;
            const int sy = a.up ? (yy >> 1) : yy, sx = a.up ? (xx >> 1) : xx;
            return *reinterpret_cast<const f32x4*>(inside ? s4 + (size_t)((img_off + (unsigned)(sy * Ws + sx)) * cs4 + (unsigned)cq + cu) : zp);
        };
        auto conv1 = [&](int j, const f32x4 r) __attribute__((always_inline)) {
            int p, cq;
            rs_coords(j, tq, p, cq);
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k] = mul_rounded(r[k], sc);
                if (a.pro == VH_PRO_SILU) v[k] = mp_silu_dev(v[k]);
            }
            unsigned H0, L0, H1, L1;
            bf16_split_pair_p(v[0], v[1], H0, L0);
            bf16_split_pair_p(v[2], v[3], H1, L1);
            const int ih = (p * 8 + ((cq >> 1) ^ (p & 7))) * 2 + (cq & 1);
            if (p < PPIX + 4) { sP2[ih] = make_uint2(H0, H1); sP2[ih ^ 8] = make_uint2(L0, L1); }
        };
#pragma unroll
        for (int g0 = 0; g0 < PR - 1; g0 += RSG) {
            f32x4 r[RSG];
#pragma unroll
            for (int k = 0; k < RSG; ++k) if (g0 + k < PR - 1) r[k] = load1(g0 + k);
            asm volatile("" ::: "memory");                     // (keep the group's loads ahead of its conversions)
#pragma unroll
            for (int k = 0; k < RSG; ++k) if (g0 + k < PR - 1) conv1(g0 + k, r[k]);
        }
        if (w == 0) conv1(PR - 1, load1(PR - 1));
    };

    // ---- weight staging (as conv_x3_glds): row = output channel w*8 + (l>>3), LDS unit l&7, swizzle (row>>1)&7 on the source side
    const int KU = a.k_pad >> 2;
    const int uslotB = (l & 7) ^ (((w & 1) << 2) | (l >> 4));
    const int uB = (uslotB & 3) * 2 + (uslotB >> 2);
    unsigned pbo[RB], pbv = 0;                                 // weight row of this lane per piece: 32-bit offset from a.wt in 16-byte units (+ a valid bit): half the registers of pointers
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int rloc = j * 64 + w * 8 + (l >> 3), gnB = nb * BN + rloc;
        const bool valid = rloc < BN && gnB < a.cout;
        pbo[j] = valid ? (unsigned)gnB * (unsigned)KU + (unsigned)uB : 0u;
        pbv |= valid ? (1u << j) : 0u;
    }
    auto issueB = [&](int st, int ku) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < RB; ++j) glds16p(((pbv >> j) & 1u) ? a.wt + (size_t)(pbo[j] + (unsigned)ku) : zp, ldsB_w + (unsigned)st * (unsigned)(RB * 8192) + j * 8192u);
    };

    f32x4 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    const int l15 = l & 15, kg = l >> 4;
    const int pbase0 = (2 * w + 1) * PP + l15 + 1;             // patch pixel of (tile row 2w, column l15) for the centre tap; row 2w+1 is PP further
    const int brow16 = l15 * 8, u16h = kg ^ (l15 >> 1);

    auto mfma3 = [&](f32x4& c, const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl) __attribute__((always_inline)) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
    };
    // one K-tile: tap offset `off` (patch pixels) of the resident chunk against weight stage st
    auto compute = [&](int st, int off) __attribute__((always_inline)) {
        // A fragments of both M tiles first (16 registers), then one N tile of weights at a time (8): the all-B-first order of conv_x3_glds
        // holds 32 registers of weights, which this kernel needs for the next chunk's patch (PF)
        bf16x8 ah[2], al[2];
        int pb = pbase0;
        asm volatile("" : "+v"(pb));       // opaque: otherwise hipcc computes the 18 (tap, M tile) fragment addresses ahead of the loop and keeps them live (spills at NJ = 8)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = pb + i * PP + off;
            const int ih = p * 8 + (kg ^ (p & 7));
            ah[i] = *reinterpret_cast<const bf16x8*>(&sP[ih]);
            al[i] = *reinterpret_cast<const bf16x8*>(&sP[ih ^ 4]);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&sB[st][brow16 + j * 128 + u16h]);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&sB[st][(brow16 + j * 128 + u16h) ^ 4]);     // lo unit = hi unit ^ 4
#pragma unroll
            for (int i = 0; i < 2; ++i) mfma3(acc[i][j], ah[i], al[i], bh, bl);
        }
    };

    // ---- main loop: chunks of 32 channels; per chunk the patch is staged once and the nine taps run against it -------------------------
    const int nch = a.cin_pad >> 5;
    const int ntail = TAIL == 3 ? ((a.c1 + a.c2) >> 5) : TAIL != 0 ? (a.c1 >> 5) : 0;
#ifdef VH_CLOCK
    PCK(ck_p1);
#endif
    if constexpr (RS) { issueB(0, 0); stage_f32(0); } else { load_patch(0); issueB(0, 0); }
    wait_dma_p();
    __syncthreads();
#ifdef VH_CLOCK
    PCK(ck_m0);
    const unsigned long long ck_r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    int st = 0;
    f32x4 nx[PR - 1 - PXL];                                    // PF: the next chunk's register-staged pieces of this thread, requested at tap 0 of the chunk before
    unsigned ldsX_w = 0, ldsX6 = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (PF) {
        ldsX_w = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)&sX[w * 64]);
        ldsX6 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)&sX[PXL * 512]);
    }
#endif
    for (int c = 0; c < nch; ++c) {
        if (c > 0) {                                           // (the last barrier of the previous chunk retired every read of its patch)
#ifdef VH_CLOCK
            PCK(ck_b0);
#endif
            if constexpr (PF) {
#pragma unroll
                for (int j = 0; j < PXL; ++j) sP[j * 512 + t] = sX[j * 512 + t];   // (each thread moves the slots its own DMA filled; waited for by tap 1's vmcnt(0))
#pragma unroll
                for (int j = PXL; j < PR - 1; ++j) *reinterpret_cast<f32x4*>(&sP[j * 512 + t]) = nx[j - PXL];
                if (w == 0) sP[(PR - 1) * 512 + l] = sX[PXL * 512 + l];
            } else if constexpr (RSP) {
                convert_f32(c);                                // (fetched during taps 7 and 8 of the chunk before)
            } else if constexpr (RS) {
                stage_f32(c);
                wait_dma_p();
            } else {
                load_patch(c * 8);
                wait_dma_p();
            }
            __syncthreads();
#ifdef VH_CLOCK
            PCK(ck_b1);
            ck_bsum += ck_b1 - ck_b0;
#endif
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            // weights of the next K-tile: next tap of this chunk, tap 0 of the next chunk, or the first tail tile
            if (tap < 8) issueB(st ^ 1, ((tap + 1) * a.cin_pad + c * 32) >> 2);
            else if (c + 1 < nch) issueB(st ^ 1, ((c + 1) * 32) >> 2);
            else if (TAIL != 0 && ntail > 0) issueB(st ^ 1, (9 * a.cin_pad) >> 2);
            const bool rsp = RSP && tap == 7 && c + 1 < nch;
            if (rsp) fetch_f32(c + 1);                         // plain loads behind this K-tile's weight DMA (in-order queue): the tile waits for all but these
            const bool pf = PF && tap == 0 && c + 1 < nch;
            if (pf) {
                // plain loads, behind this K-tile's weight DMA in the wave's (in-order) memory queue: the tile waits for all but these
                const unsigned cu = (unsigned)(c + 1) * 8u;
                int tq = t;
                asm volatile("" : "+v"(tq));
#pragma unroll
                for (int j = 0; j < PXL; ++j) glds16p(piece_src(j, cu, tq), ldsX_w + j * 8192u);
                if (w == 0) glds16p(piece_src(PR - 1, cu, tq), ldsX6);
#pragma unroll
                for (int j = PXL; j < PR - 1; ++j) nx[j - PXL] = *reinterpret_cast<const f32x4*>(piece_src(j, cu, tq));
            }
            compute(st, dy * PP + dx);
#ifdef VH_CLOCK
            PCK(ck_k0);
#endif
            if (pf) wait_dma_but<PR - 1 - PXL>();              // (the register-staged loads are the youngest; the LDS-DMA pieces before them are waited for with the weights)
            else if (rsp) { if (w == 0) wait_dma_but<PR>(); else wait_dma_but<PR - 1>(); }
            else wait_dma_p();
            __syncthreads();
#ifdef VH_CLOCK
            PCK(ck_k1);
            ck_ksum += ck_k1 - ck_k0;
#endif
            st ^= 1;
        }
    }
    if constexpr (TAIL == 2) {
        // 1-tap tail over the second source, wave-private: a wave's 32 pixels x 32 channels (4 KB) are fetched by that wave itself - coalesced, 8 lanes
        // per 128-byte pixel line, one K-tile ahead into registers - written to its own corner of the (now free) patch space and read back as MFMA
        // fragments: the LDS pass is only the lane transpose, there is nothing to publish, so the A path needs no barrier and no second stage
        // (the weights keep their two shared stages and the per-K-tile barrier).
        if (ntail > 0) {
            // (lane coordinates re-derived from the hardware counter: carried across the main loop they cost it a register it does not have at NJ = 8)
            const int lt = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), l15t = lt & 15, kgt = lt >> 4;
            const int brow16t = l15t * 8, u16ht = kgt ^ (l15t >> 1);
            const float4* s14 = reinterpret_cast<const float4*>(a.src1);
            const unsigned c14 = (unsigned)(a.c1 >> 2);
            f32x4 nx[4];
            // piece j of this lane: pixel q = (lane>>3) + 8j of the wave's 32 (tile row 2w + (q>>4), column q & 15), LDS unit lane & 7.  Its source is
            // recomputed per K-tile from an opaque copy of the lane id (cf. piece_src): five registers the NJ = 8 loop does not have
            auto fetchA = [&](unsigned cu) __attribute__((always_inline)) {
                int lq = lt;
                asm volatile("" : "+v"(lq));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = (lq >> 3) + 8 * j, up = lq & 7;
                    const int yy = y0 + 2 * w + (q >> 4), xx = x0 + (q & 15);
                    const bool inside = yy < a.h && xx < a.w;
                    const int u = up ^ (q & 7), s8u = (u & 3) * 2 + (u >> 2);
                    nx[j] = *reinterpret_cast<const f32x4*>(inside ? s14 + (size_t)((img_off + (unsigned)(yy * a.w + xx)) * c14 + (unsigned)s8u + cu) : zp);
                }
            };
            fetchA(0u);
            float4* const mine = sA + w * 256;                                // this wave's [32 pixels][8 units], swizzled q & 7 like the patch
            for (int c = 0; c < ntail; ++c) {
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(&mine[j * 64 + lt]) = nx[j];      // slot (q, up) = q * 8 + up = j * 64 + l
                if (c + 1 < ntail) {
                    issueB(st ^ 1, (9 * a.cin_pad + (c + 1) * 32) >> 2);
                    fetchA((unsigned)(c + 1) * 8u);
                }
                bf16x8 ah[2], al[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int q = i * 16 + l15t;
                    const int ih = q * 8 + (kgt ^ (q & 7));
                    ah[i] = *reinterpret_cast<const bf16x8*>(&mine[ih]);
                    al[i] = *reinterpret_cast<const bf16x8*>(&mine[ih ^ 4]);
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&sB[st][brow16t + j * 128 + u16ht]);
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&sB[st][(brow16t + j * 128 + u16ht) ^ 4]);     // lo unit = hi unit ^ 4
#pragma unroll
                    for (int i = 0; i < 2; ++i) mfma3(acc[i][j], ah[i], al[i], bh, bl);
                }
                if (c + 1 < ntail) wait_dma_but<4>();                        // the next weight tile has landed; the four fragment loads behind it stay in flight
                else wait_dma_p();
                __syncthreads();
                st ^= 1;
            }
        }
    }
    if constexpr (TAIL == 3) {
        // TAIL == 2 with fp32 sources: K-tiles 0 .. c1/32-1 of the tail come from src1 (fp32 NHWC, c1 channels, times scale1), the rest from src2
        // (c2 channels, times scale2) - mp_cat(x, skip) itself (training/models.py:78-84).  Lane (pixel q = (lane>>3) + 8j, quad cq = lane & 7) loads
        // channels 4cq..4cq+3 of the K-tile (8 lanes = the pixel's 128 bytes), scales, splits (the operations and roundings of split_k's raw form)
        // and writes 8 bytes of hi unit cq>>1 and 8 bytes of lo unit 4 + (cq>>1) of its wave's private [32 pixels][8 units] area.
        if (ntail > 0) {
            const int lt = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), l15t = lt & 15, kgt = lt >> 4;
            const int brow16t = l15t * 8, u16ht = kgt ^ (l15t >> 1);
            const int n1 = a.c1 >> 5;                                       // K-tiles of the first source
            f32x4 nx[4];
            auto fetchA = [&](int c) __attribute__((always_inline)) {
                int lq = lt;
                asm volatile("" : "+v"(lq));
                const bool first = c < n1;
                const float4* s4 = reinterpret_cast<const float4*>(first ? a.src1 : a.src2);
                const unsigned cs4 = (unsigned)((first ? a.c1 : a.c2) >> 2), cu = (unsigned)(first ? c : c - n1) * 8u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = (lq >> 3) + 8 * j, cq = lq & 7;
                    const int yy = y0 + 2 * w + (q >> 4), xx = x0 + (q & 15);
                    const bool inside = yy < a.h && xx < a.w;
                    nx[j] = *reinterpret_cast<const f32x4*>(inside ? s4 + (size_t)((img_off + (unsigned)(yy * a.w + xx)) * cs4 + (unsigned)cq + cu) : zp);
                }
            };
            fetchA(0);
            uint2* const mine2 = reinterpret_cast<uint2*>(sA + w * 256);    // this wave's [32 pixels][8 units of 16 B], as 8-byte halves
            const float4* const mine = sA + w * 256;
            for (int c = 0; c < ntail; ++c) {
                const float sc = c < n1 ? a.scale1 : a.scale2;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = (lt >> 3) + 8 * j, cq = lt & 7;
                    unsigned H0, L0, H1, L1;
                    bf16_split_pair_p(mul_rounded(nx[j][0], sc), mul_rounded(nx[j][1], sc), H0, L0);
                    bf16_split_pair_p(mul_rounded(nx[j][2], sc), mul_rounded(nx[j][3], sc), H1, L1);
                    const int slot = q * 8 + ((cq >> 1) ^ (q & 7));           // hi unit cq>>1 of pixel q (swizzled like the patch); its lo unit is slot ^ 4
                    mine2[slot * 2 + (cq & 1)] = make_uint2(H0, H1);
                    mine2[(slot ^ 4) * 2 + (cq & 1)] = make_uint2(L0, L1);
                }
                if (c + 1 < ntail) {
                    issueB(st ^ 1, (9 * a.cin_pad + (c + 1) * 32) >> 2);
                    fetchA(c + 1);
                }
                bf16x8 ah[2], al[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int q = i * 16 + l15t;
                    const int ih = q * 8 + (kgt ^ (q & 7));
                    ah[i] = *reinterpret_cast<const bf16x8*>(&mine[ih]);
                    al[i] = *reinterpret_cast<const bf16x8*>(&mine[ih ^ 4]);
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&sB[st][brow16t + j * 128 + u16ht]);
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&sB[st][(brow16t + j * 128 + u16ht) ^ 4]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) mfma3(acc[i][j], ah[i], al[i], bh, bl);
                }
                if (c + 1 < ntail) wait_dma_but<4>();
                else wait_dma_p();
                __syncthreads();
                st ^= 1;
            }
        }
    }
    if constexpr (TAIL == 1) {
        // 1-tap tail over the second source (c1 channels, same 256 pixels, no halo): conv_x3_glds' loop - per K-tile a 256-row x 32-channel
        // A tile (row r = tile pixel (r>>4, r&15), swizzle r&7) and a weight tile staged by LDS-DMA into the stage not being read.
        if (ntail > 0) {
            const float4* s14 = reinterpret_cast<const float4*>(a.src1);
            const unsigned c14 = (unsigned)(a.c1 >> 2);
            unsigned to[4], tin = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = j * 64 + (t >> 3), up = t & 7;
                const int yy = y0 + (r >> 4), xx = x0 + (r & 15);
                const bool inside = yy < a.h && xx < a.w;
                const int u = up ^ (r & 7), s8u = (u & 3) * 2 + (u >> 2);
                to[j] = inside ? (img_off + (unsigned)(yy * a.w + xx)) * c14 + (unsigned)s8u : 0u;
                tin |= inside ? (1u << j) : 0u;
            }
            auto issueA = [&](int sa, unsigned cu) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) glds16p(((tin >> j) & 1u) ? s14 + (size_t)(to[j] + cu) : zp, ldsP_w + (unsigned)sa * 32768u + j * 8192u);
            };
            issueA(0, 0u);                                     // (the main loop's last barrier retired every read of the patch; tile 0's weights are already in sB[st])
            wait_dma_p();
            __syncthreads();
            const int rbase = (2 * w) * 16 + l15;
            for (int c = 0; c < ntail; ++c) {
                const int sa = c & 1;
                if (c + 1 < ntail) {
                    issueB(st ^ 1, (9 * a.cin_pad + (c + 1) * 32) >> 2);
                    issueA(sa ^ 1, (unsigned)(c + 1) * 8u);
                }
                bf16x8 ah[2], al[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = rbase + i * 16;
                    const int ih = sa * 2048 + r * 8 + (kg ^ (r & 7));
                    ah[i] = *reinterpret_cast<const bf16x8*>(&sA[ih]);
                    al[i] = *reinterpret_cast<const bf16x8*>(&sA[ih ^ 4]);
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&sB[st][brow16 + j * 128 + u16h]);
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&sB[st][(brow16 + j * 128 + u16h) ^ 4]);     // lo unit = hi unit ^ 4
#pragma unroll
                    for (int i = 0; i < 2; ++i) mfma3(acc[i][j], ah[i], al[i], bh, bl);
                }
                wait_dma_p();
                __syncthreads();
                st ^= 1;
            }
        }
    }
#ifdef VH_CLOCK
    PCK(ck_m1);
    const unsigned long long ck_r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    unsigned long long* dbg = a.dbg && blockIdx.x < 16384u && t == 0 ? a.dbg + (size_t)blockIdx.x * 12 : nullptr;
    if (dbg) { dbg[0] = ck_m1 - ck_m0; dbg[1] = ck_r1 - ck_r0; dbg[2] = ck_m0 - ck_e0; dbg[10] = ck_p1 - ck_e0; dbg[11] = ck_m0 - ck_p1;
               dbg[4] = ck_bsum; dbg[5] = ck_ksum;
               dbg[6] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID: wave [3:0], simd [5:4], pipe [7:6], cu [11:8], sh [12], se [15:13]
               dbg[7] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
               dbg[8] = ck_r0; dbg[9] = ck_r1; }
#endif
    // ---- epilogue: the wave's two 32x32 blocks through its LDS transpose patch (the loop's last barrier retired every read of sP) ----
    float* patch = reinterpret_cast<float*>(&sP[0]) + w * (32 * 36);
    const int yb = y0 + 2 * w;
    if (yb >= a.h) return;
    // (the lane id re-derived from the hardware counter: carried over from the prologue it costs the K loop a register it does not have)
    const int le = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if constexpr (NJ == 1) {
        // narrow output (Cout <= 16, VH_EPI_STORE): straight from the accumulators - C/D map of the 16x16 tile: column = lane & 15, pixels
        // 4 * (lane >> 4) + r of tile row 2w + i; 12 B per pixel at Cout = 3, no transpose worth making
        const int col = le & 15, xq = x0 + 4 * (le >> 4);
        if (col < a.cout) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int yy = yb + i;
                if (yy >= a.h) continue;
                float* orow = a.out + ((size_t)img * a.HW + (size_t)yy * a.w) * a.cout + col;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][0][r];
                    if (a.clip > 0.f) v = fminf(fmaxf(v, -a.clip), a.clip);
                    if (xq + r < a.w) orow[(size_t)(xq + r) * a.cout] = v;
                }
            }
        }
        return;
    }
    auto run = [&](auto epic) __attribute__((always_inline)) {
        constexpr int EPI = decltype(epic)::value;
        const int cb = nb * BN;                                // first output channel of this workgroup
        if constexpr (NJ < 8) {
            // the residual / cvec values of block b + 1 are requested before block b is written out
            PAux cur = patch_prefetch<EPI>(a, img, yb, x0, cb, le);
#pragma unroll
            for (int ni = 0; ni < NJ / 2; ++ni) {
                PAux nxt = cur;
                if (ni + 1 < NJ / 2 && cb + (ni + 1) * 32 < a.cout) nxt = patch_prefetch<EPI>(a, img, yb, x0, cb + (ni + 1) * 32, le);
                if (cb + ni * 32 < a.cout)                     // (cout % 32 == 0: blocks are all-in or all-out)
                    patch_block_out<EPI>(a, acc[0][2 * ni], acc[0][2 * ni + 1], acc[1][2 * ni], acc[1][2 * ni + 1], img, yb, x0, cb + ni * 32, patch, le, cur);
                cur = nxt;
            }
        } else {
            // (64 accumulator registers: no room to hold a second block's values in flight - 24 B of scratch with them)
#pragma unroll
            for (int ni = 0; ni < NJ / 2; ++ni) {
                if (cb + ni * 32 >= a.cout) continue;
                const PAux cur = patch_prefetch<EPI>(a, img, yb, x0, cb + ni * 32, le);
                patch_block_out<EPI>(a, acc[0][2 * ni], acc[0][2 * ni + 1], acc[1][2 * ni], acc[1][2 * ni + 1], img, yb, x0, cb + ni * 32, patch, le, cur);
            }
        }
    };
    if (a.epi == VH_EPI_MPSUM) run(std::integral_constant<int, VH_EPI_MPSUM>{});
    else if (a.epi == VH_EPI_SCALE_SILU) run(std::integral_constant<int, VH_EPI_SCALE_SILU>{});
    else run(std::integral_constant<int, VH_EPI_STORE>{});
#ifdef VH_CLOCK
    { PCK(ck_x); if (dbg) dbg[3] = ck_x - ck_m1; }
#endif
}

}  // namespace

int vh_diag_conv_patch() { return VH_DIAG_FLAG; }

// Launch of the patch-resident kernel (arguments validated by vh_conv / chosen by vh_conv_x3_glds_dispatch): 3x3 (with or without `up`),
// cout % 32 == 0 or <= 16, epilogue STORE / SCALE_SILU / MPSUM.  One workgroup per 16x16 output tile per image and block of 64 (with a tail segment)
// or 128 output channels.
void vh_conv_x3_patch_launch(vhconv::ConvK k, hipStream_t s) {
    k.ptx = (k.w + PT - 1) / PT;
    k.pty = (k.h + PT - 1) / PT;
    k.div_ptx = vhconv::fastdiv_make((unsigned)k.ptx);
    k.div_ptiles = vhconv::fastdiv_make((unsigned)(k.ptx * k.pty));
    // 128 (or 96) output channels per workgroup beyond 64; with a tail segment the wave-private form (TAIL = 2) serves every block width,
    // knob "conv_patch_tail": 2 (default) always, 1 the LDS-DMA staged tail where it exists (64-channel blocks)
    if (k.cout <= 16) {                                      // narrow outputs (validated by vh_conv_patch_choice: plain store, no tail, no S8 forms)
        k.NT = 1;
        const unsigned grid = (unsigned)((long long)(k.M / k.HW) * k.ptx * k.pty);
        if (k.cin_pad <= 64) hipLaunchKernelGGL((conv_x3_patch<1, 0, true>), dim3(grid), dim3(512), 0, s, k);
        else hipLaunchKernelGGL((conv_x3_patch<1, 0, false>), dim3(grid), dim3(512), 0, s, k);
        return;
    }
    const int tmode = (k.c1 > 0 && !k.src_f32) ? (k.tail_f32 ? 3 : vh_knob(VH_KNOB_CONV_PATCH_TAIL) == 1 ? 1 : 2) : 0;
    const bool wide = k.cout > 64 && tmode != 1;
    const bool n96 = wide && k.cout % 96 == 0 && k.cout % 128 != 0;
    const int bn = n96 ? 96 : wide ? 128 : 64;
    k.NT = (k.cout + bn - 1) / bn;
    const unsigned grid = (unsigned)((long long)(k.M / k.HW) * k.ptx * k.pty * k.NT);
    const bool pf = k.cin_pad <= 64 && bn == 64;
    if (k.src_f32) {                                         // register-staged fp32 sources (no tail, no look-ahead patch; validated by vh_conv)
        if (n96) hipLaunchKernelGGL((conv_x3_patch<6, 0, false, true>), dim3(grid), dim3(512), 0, s, k);
        else if (wide) hipLaunchKernelGGL((conv_x3_patch<8, 0, false, true>), dim3(grid), dim3(512), 0, s, k);
        else hipLaunchKernelGGL((conv_x3_patch<4, 0, false, true>), dim3(grid), dim3(512), 0, s, k);
        return;
    }
#define VH_PATCH_LAUNCH(NJ_, T_, PF_) hipLaunchKernelGGL((conv_x3_patch<NJ_, T_, PF_>), dim3(grid), dim3(512), 0, s, k)
    if (tmode == 0) {
        if (n96) VH_PATCH_LAUNCH(6, 0, false); else if (wide) VH_PATCH_LAUNCH(8, 0, false); else if (pf) VH_PATCH_LAUNCH(4, 0, true); else VH_PATCH_LAUNCH(4, 0, false);
    } else if (tmode == 1) {
        if (pf) VH_PATCH_LAUNCH(4, 1, true); else VH_PATCH_LAUNCH(4, 1, false);
    } else if (tmode == 2) {
        if (n96) VH_PATCH_LAUNCH(6, 2, false); else if (wide) VH_PATCH_LAUNCH(8, 2, false); else if (pf) VH_PATCH_LAUNCH(4, 2, true); else VH_PATCH_LAUNCH(4, 2, false);
    } else {
        if (n96) VH_PATCH_LAUNCH(6, 3, false); else if (wide) VH_PATCH_LAUNCH(8, 3, false); else if (pf) VH_PATCH_LAUNCH(4, 3, true); else VH_PATCH_LAUNCH(4, 3, false);
    }
#undef VH_PATCH_LAUNCH
}
