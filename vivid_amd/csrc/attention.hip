// Flash-style attention on fp32 MFMA for gfx950 (CDNA4).
//
// Replaces F.scaled_dot_product_attention(q, k, v) of the reference
// (training/models.py:198, :305; explicit einsum+softmax in the snapshot,
// experiments/code/training/models.py:190-191, 273-280): no mask, no dropout,
// scale 1/sqrt(D) (folded into q together with log2(e) by vh_qkv_split).
//
// One workgroup = 4 waves = 4*QT*32 queries of one (batch, head); keys/values stream
// through LDS in tiles of 64.  Per 32-key sub-tile each wave computes
//     S^T[key][q] = K[key][:] . Q[q][:]       A = K tile (LDS),  B = Q (registers)
// so the query sits on the MFMA lane and a query's 32 logits are the lane's 16
// accumulator registers plus its partner lane (lane^32): the online-softmax row
// statistics need one cross-lane exchange, and P never leaves registers —
//     O^T[d][q] += V^T[d][key] * P^T[key][q]  A = V^T tile (LDS), B = P (registers)
// takes the S^T accumulators directly as B operands (k order = the accumulator row map,
// matched by the order V^T is read in).
// LDS images: K as float4 groups of 4 consecutive d per key, [d/4][key ^ (d/4 & 7)];
// V transposed, [d][key] with a 68-float row stride (conflict-free ds_read_b128).
#include "ctx.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int KT = 64;        // keys per LDS tile
constexpr int VS = KT + 4;    // V^T row stride (floats)

struct AttnK {
    const float* q; const float* k; const float* v; float* out;
    int heads, s, kl, c;
    float n_zero;
};

template <int D, int QT>
__global__ __launch_bounds__(256, 2) void attn_fwd_f32(const AttnK a) {
    constexpr int D4 = D / 4;
    __shared__ float4 sK[D4 * KT];
    __shared__ float sV[D * VS];

    const int t = threadIdx.x;
    const int wv = t >> 6, l = t & 63, lr = l & 31, hh = l >> 5;
    const int bh = blockIdx.y;
    const int b = bh / a.heads, hd = bh - b * a.heads;
    const int q0 = blockIdx.x * (4 * QT * 32) + wv * (QT * 32);

    const float* Qb = a.q + (size_t)bh * a.s * D;
    const float* Kb = a.k + (size_t)bh * a.kl * D;
    const float* Vb = a.v + (size_t)bh * a.kl * D;

    // Q fragments: step j = 4*kg + jj uses d = 8*kg + 4*hh + jj
    float qf[QT][D / 2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qrow = q0 + qt * 32 + lr;
#pragma unroll
        for (int kg = 0; kg < D / 8; ++kg) {
            float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qrow < a.s) v4 = *reinterpret_cast<const float4*>(Qb + (size_t)qrow * D + kg * 8 + hh * 4);
            qf[qt][kg * 4 + 0] = v4.x; qf[qt][kg * 4 + 1] = v4.y;
            qf[qt][kg * 4 + 2] = v4.z; qf[qt][kg * 4 + 3] = v4.w;
        }
    }

    f32x16 oacc[QT][D / 32];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qt][dt][r] = 0.f;
    float mrow[QT], lsum[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        // zero-logit phantom keys (unconditional guidance net): start from max 0, sum n_zero
        mrow[qt] = a.n_zero > 0.f ? 0.f : -1e30f;
        lsum[qt] = a.n_zero;
    }

    // staging map: 16 float4 (D=64) or 8 (D=32) per key row; KT*D4 float4 per tile
    constexpr int PER_T = KT * D4 / 256;    // 4 (D=64) or 2 (D=32)
    constexpr int KSTEP = 256 / D4;         // keys covered per pass
    const int sd4 = t % D4, skey = t / D4;
    float4 rk[PER_T], rv[PER_T];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int key = k0 + skey + i * KSTEP;
            const bool ok = key < a.kl;
            rk[i] = ok ? *reinterpret_cast<const float4*>(Kb + (size_t)key * D + sd4 * 4) : zero4;
            rv[i] = ok ? *reinterpret_cast<const float4*>(Vb + (size_t)key * D + sd4 * 4) : zero4;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int key = skey + i * KSTEP;
            sK[sd4 * KT + (key ^ (sd4 & 7))] = rk[i];
            sV[(sd4 * 4 + 0) * VS + key] = rv[i].x;
            sV[(sd4 * 4 + 1) * VS + key] = rv[i].y;
            sV[(sd4 * 4 + 2) * VS + key] = rv[i].z;
            sV[(sd4 * 4 + 3) * VS + key] = rv[i].w;
        }
    };

    const int ntiles = (a.kl + KT - 1) / KT;
    load_tile(0);
    for (int tile = 0; tile < ntiles; ++tile) {
        const int k0 = tile * KT;
        __syncthreads();              // previous tile's readers are done
        store_tile();
        __syncthreads();
        if (tile + 1 < ntiles) load_tile(k0 + KT);

#pragma unroll
        for (int ks = 0; ks < KT / 32; ++ks) {
            // ---- S^T = K Q^T -------------------------------------------------
            f32x16 sacc[QT];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[qt][r] = 0.f;
#pragma unroll
            for (int kg = 0; kg < D / 8; ++kg) {
                const int d4 = kg * 2 + hh;
                const float4 kf = sK[d4 * KT + ((ks * 32 + lr) ^ (d4 & 7))];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float kv = jj == 0 ? kf.x : jj == 1 ? kf.y : jj == 2 ? kf.z : kf.w;
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        sacc[qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv, qf[qt][kg * 4 + jj], sacc[qt], 0, 0, 0);
                }
            }
            // ---- online softmax (per query = per lane, pairs with lane^32) ------
            const int kbase = k0 + ks * 32 + 4 * hh;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                float mx = -1e30f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase + (r & 3) + 8 * (r >> 2);
                    if (key >= a.kl) sacc[qt][r] = -INFINITY;
                    mx = fmaxf(mx, sacc[qt][r]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float mnew = fmaxf(mrow[qt], mx);
                const float alpha = __builtin_amdgcn_exp2f(mrow[qt] - mnew);
                float rs = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(sacc[qt][r] - mnew);
                    sacc[qt][r] = p;
                    rs += p;
                }
                rs += __shfl_xor(rs, 32);
                lsum[qt] = lsum[qt] * alpha + rs;
                mrow[qt] = mnew;
#pragma unroll
                for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[qt][dt][r] *= alpha;
            }
            // ---- O^T += V^T P^T ---------------------------------------------------
#pragma unroll
            for (int dt = 0; dt < D / 32; ++dt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 vf = *reinterpret_cast<const float4*>(&sV[(dt * 32 + lr) * VS + ks * 32 + 8 * g + 4 * hh]);
#pragma unroll
                    for (int cidx = 0; cidx < 4; ++cidx) {
                        const float vv = cidx == 0 ? vf.x : cidx == 1 ? vf.y : cidx == 2 ? vf.z : vf.w;
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt)
                            oacc[qt][dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, sacc[qt][4 * g + cidx], oacc[qt][dt], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- normalise and store: out[b][q][hd*D + d], d = dt*32 + (r&3) + 8*(r>>2) + 4*hh
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qrow = q0 + qt * 32 + lr;
        if (qrow >= a.s) continue;
        const float inv = 1.0f / lsum[qt];
        float* op = a.out + ((size_t)b * a.s + qrow) * a.c + hd * D;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 o;
                o.x = oacc[qt][dt][4 * g + 0] * inv; o.y = oacc[qt][dt][4 * g + 1] * inv;
                o.z = oacc[qt][dt][4 * g + 2] * inv; o.w = oacc[qt][dt][4 * g + 3] * inv;
                *reinterpret_cast<float4*>(op + dt * 32 + 8 * g + 4 * hh) = o;
            }
    }
}

}  // namespace

extern "C" int vh_attention(vh_ctx* ctx, const vh_attention_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_attention: null args");
    const vh_attention_args a = *p;
    VH_REQUIRE(a.q && a.k && a.v && a.out, "vh_attention: null tensor");
    VH_REQUIRE(a.d == 32 || a.d == 64, "vh_attention: head dim %d unsupported (32 or 64; the reference's presets use 64, 32 in the SR UNet)", a.d);
    VH_REQUIRE(a.b > 0 && a.heads > 0 && a.s > 0 && a.kl > 0, "vh_attention: bad geometry");
    VH_REQUIRE(vh_aligned16(a.q) && vh_aligned16(a.k) && vh_aligned16(a.v) && vh_aligned16(a.out), "vh_attention: pointers must be 16-byte aligned");
    VH_REQUIRE((long long)a.b * a.heads < 65536, "vh_attention: b*heads too large for grid.y");
    VH_REQUIRE(a.n_zero_keys >= 0.f, "vh_attention: negative n_zero_keys");
    AttnK k{a.q, a.k, a.v, a.out, a.heads, a.s, a.kl, a.heads * a.d, a.n_zero_keys};
    const int d = a.d;
    const int qt = a.s >= 256 ? 2 : 1;
    const dim3 grid((a.s + 128 * qt - 1) / (128 * qt), a.b * a.heads);
    const double bh = (double)a.b * a.heads;
    const double flops = 4.0 * bh * a.s * a.kl * a.d;
    const double bytes = 4.0 * bh * a.d * (2.0 * a.s + 2.0 * a.kl);
    return vh_dispatch(ctx, VH_TAG_ATTN, flops, bytes, [k, d, qt, grid](hipStream_t s) -> int {
        if (d == 64 && qt == 2) hipLaunchKernelGGL((attn_fwd_f32<64, 2>), grid, dim3(256), 0, s, k);
        else if (d == 64) hipLaunchKernelGGL((attn_fwd_f32<64, 1>), grid, dim3(256), 0, s, k);
        else if (qt == 2) hipLaunchKernelGGL((attn_fwd_f32<32, 2>), grid, dim3(256), 0, s, k);
        else hipLaunchKernelGGL((attn_fwd_f32<32, 1>), grid, dim3(256), 0, s, k);
        return vh_check_launch("attn_fwd_f32");
    });
}
