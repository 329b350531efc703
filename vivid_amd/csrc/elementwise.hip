// HBM-bound helper kernels of the VIVID denoiser for gfx950: weight preparation, pixel norm
// (+2x2 mean pooling), q/k/v split + head norm, noise/pose embedding, batched emb_linear,
// input assembly, preconditioned output, depth-warp Fourier features, sampler update.
// Reference lines are cited at each entry point in include/vivid_hip.h.
#include "ctx.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int t = threadIdx.x;
    if ((t & 63) == 0) red[t >> 6] = v;
    __syncthreads();
    const float r = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return r;
}

__device__ __forceinline__ float mp_silu_dev(float v) {
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * v);
    return v * __builtin_amdgcn_rcpf(1.0f + e) * (1.0f / 0.596f);
}

__device__ __forceinline__ unsigned bf16_rn_bits(float v) {      // v_cvt_pk_bf16_f32: round to nearest even
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v);
}
__device__ __forceinline__ void split_bf16(float v, unsigned& hi, unsigned& lo) {
    hi = bf16_rn_bits(v);
    lo = bf16_rn_bits(v - __uint_as_float(hi << 16));
}

// (a, b) -> packed bf16 hi pair and lo pair: one v_cvt_pk_bf16_f32 per pair, the same values as two split_bf16 calls
__device__ __forceinline__ void split_bf16_pair(float a, float b, unsigned& H, unsigned& L) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t h; h[0] = (__bf16)a; h[1] = (__bf16)b;
    H = __builtin_bit_cast(unsigned, h);
    bf16x2_t q; q[0] = (__bf16)(a - __uint_as_float(H << 16)); q[1] = (__bf16)(b - __uint_as_float(H & 0xFFFF0000u));
    L = __builtin_bit_cast(unsigned, q);
}
// 8 values -> the 32-byte S8 chunk [hi x8 | lo x8]
__device__ __forceinline__ void store_s8_chunk(uint4* o, const float (&e)[8]) {
    unsigned H[4], L[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_bf16_pair(e[2 * j], e[2 * j + 1], H[j], L[j]);
    o[0] = make_uint4(H[0], H[1], H[2], H[3]);
    o[1] = make_uint4(L[0], L[1], L[2], L[3]);
}

// ---------------------------------------------------------------- weight prep
__global__ __launch_bounds__(256) void prep_weight_k(vh_prep_weight_args a) {
    __shared__ float red[4];
    const int o = blockIdx.x, t = threadIdx.x;
    const int fan = a.cin * a.taps;
    const float* w = a.w + (size_t)o * fan;
    float ss = 0.f;
    for (int i = t; i < fan; i += 256) ss += w[i] * w[i];
    ss = block_sum_256(ss, red);
    const float gain = a.gain_ptr ? *a.gain_ptr : a.gain_value;
    const float rs = rsqrtf((float)fan);
    const float scale = gain * rs / (1e-4f + sqrtf(ss) * rs);
    for (int k = t; k < a.k_pad; k += 256) {
        const int tap = k / a.cin_pad, ci = k - tap * a.cin_pad;
        float v = 0.f;
        if (tap < a.taps && ci < a.cin) v = w[ci * a.taps + tap] * scale;
        if (a.split == 2) {
            unsigned hi, lo;
            split_bf16(v, hi, lo);
            // row of output channel o: k_stride K elements (0 = k_pad); this weight's K range starts at k_off (a fused 3x3 + 1x1 weight is
            // written by two calls: the 3x3 part at 0, the 1x1 part behind it)
            const int kd = a.k_off + k;
            unsigned short* ws = reinterpret_cast<unsigned short*>(a.wt) + (size_t)(a.dst_col0 + o) * (a.k_stride ? a.k_stride : a.k_pad) * 2;
            ws[(size_t)(kd >> 3) * 16 + (kd & 7)] = (unsigned short)hi;
            ws[(size_t)(kd >> 3) * 16 + 8 + (kd & 7)] = (unsigned short)lo;
        } else if (a.split) {
            unsigned hi, lo;
            split_bf16(v, hi, lo);
            unsigned short* ws = reinterpret_cast<unsigned short*>(a.wt);
            const size_t u = (size_t)(k >> 3) * 2;
            ws[((u + 0) * a.dst_cols + a.dst_col0 + o) * 8 + (k & 7)] = (unsigned short)hi;
            ws[((u + 1) * a.dst_cols + a.dst_col0 + o) * 8 + (k & 7)] = (unsigned short)lo;
        } else {
            a.wt[((size_t)(k >> 2) * a.dst_cols + a.dst_col0 + o) * 4 + (k & 3)] = v;
        }
    }
}

// ---------------------------------------------------------------- pixel norm (+pool)
// one wave per output pixel; channels in float4
__global__ __launch_bounds__(256) void pixnorm_k(vh_pixnorm_args a, long long npix) {
    const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= npix) return;
    const int lane = threadIdx.x & 63;
    const int c4 = a.c >> 2;
    const float4* src[4];
    int nsrc = 1;
    if (a.pool) {
        const int hw = a.h * a.w;
        const int img = (int)(p / hw), rem = (int)(p - (long long)img * hw);
        const int y = rem / a.w, x = rem - y * a.w;
        const int wi = 2 * a.w;
        const size_t base = ((size_t)img * 2 * a.h + 2 * y) * wi + 2 * x;
        src[0] = reinterpret_cast<const float4*>(a.in + base * a.c);
        src[1] = reinterpret_cast<const float4*>(a.in + (base + 1) * a.c);
        src[2] = reinterpret_cast<const float4*>(a.in + (base + wi) * a.c);
        src[3] = reinterpret_cast<const float4*>(a.in + (base + wi + 1) * a.c);
        nsrc = 4;
    } else {
        src[0] = reinterpret_cast<const float4*>(a.in + (size_t)p * a.c);
        src[1] = src[2] = src[3] = src[0];
    }
    float4* dst = a.out ? reinterpret_cast<float4*>(a.out + (size_t)p * a.c) : nullptr;
    auto fetch = [&](int i) {
        float4 v = src[0][i];
        if (nsrc == 4) {
            const float4 b = src[1][i], c = src[2][i], d = src[3][i];
            v.x = 0.25f * (v.x + b.x + c.x + d.x); v.y = 0.25f * (v.y + b.y + c.y + d.y);
            v.z = 0.25f * (v.z + b.z + c.z + d.z); v.w = 0.25f * (v.w + b.w + c.w + d.w);
        }
        return v;
    };
    float scale = 1.f;
    if (a.norm) {
        float ss = 0.f;
        for (int i = lane; i < c4; i += 64) {
            const float4 v = fetch(i);
            ss = fmaf(v.w, v.w, fmaf(v.z, v.z, fmaf(v.y, v.y, fmaf(v.x, v.x, ss))));
        }
        ss = wave_sum(ss);
        scale = 1.0f / (1e-4f + sqrtf(ss) * rsqrtf((float)a.c));
    }
    unsigned short* s8 = a.out_s8 ? static_cast<unsigned short*>(a.out_s8) + (size_t)p * a.c * 2 : nullptr;
    if (a.scale_out && lane == 0) a.scale_out[p] = scale;
    for (int i = lane; i < c4; i += 64) {
        float4 v = fetch(i);
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        if (dst) dst[i] = v;
        if (s8) {       // mp_silu(out), bf16 hi/lo: chunk of 8 channels = [hi x8 | lo x8], this float4 is half of one
            const float e[4] = {mp_silu_dev(v.x), mp_silu_dev(v.y), mp_silu_dev(v.z), mp_silu_dev(v.w)};
            unsigned h[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split_bf16(e[j], h[j], l[j]);
            unsigned short* q = s8 + (size_t)(i >> 1) * 16 + (i & 1) * 4;
            *reinterpret_cast<uint2*>(q) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
            *reinterpret_cast<uint2*>(q + 8) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
        }
    }
}

// Register-resident form for c <= 512: a wave takes PPW groups of 64/LPP pixels (LPP lanes per pixel: 16 or 32 for
// c <= 64 / 128 so that no lane idles), issues all of their loads up front (the one-pixel-per-wave form above has one
// dependent load -> reduce -> store chain per wave: latency-bound) and reads every value once.
template <int NV, int LPP, bool POOL, int PPW>
__global__ __launch_bounds__(256) void pixnorm_reg_k(vh_pixnorm_args a, long long npix) {
    constexpr int PW = 64 / LPP;                                   // pixels side by side in a wave
    const long long p0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (PPW * PW);
    if (p0 >= npix) return;
    const int lane = threadIdx.x & 63;
    const int li = lane & (LPP - 1), sub = lane / LPP;
    const int c4 = a.c >> 2;
    float4 v[PPW][NV];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const long long pq = p0 + q * PW + sub;
        const long long p = pq < npix ? pq : npix - 1;
        const float4* s0;
        size_t o1 = 0, o2 = 0, o3 = 0;
        if (POOL) {
            const int hw = a.h * a.w;
            const int img = (int)(p / hw), rem = (int)(p - (long long)img * hw);
            const int y = rem / a.w, x = rem - y * a.w;
            const int wi = 2 * a.w;
            const size_t base = ((size_t)img * 2 * a.h + 2 * y) * wi + 2 * x;
            s0 = reinterpret_cast<const float4*>(a.in + base * a.c);
            o1 = (size_t)c4; o2 = (size_t)wi * c4; o3 = o2 + c4;
        } else {
            s0 = reinterpret_cast<const float4*>(a.in + (size_t)p * a.c);
        }
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int i = li + LPP * n;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < c4) {
                t = s0[i];
                if (POOL) {
                    const float4 b = s0[i + o1], c = s0[i + o2], d = s0[i + o3];
                    t.x = 0.25f * (t.x + b.x + c.x + d.x); t.y = 0.25f * (t.y + b.y + c.y + d.y);
                    t.z = 0.25f * (t.z + b.z + c.z + d.z); t.w = 0.25f * (t.w + b.w + c.w + d.w);
                }
            }
            v[q][n] = t;
        }
    }
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const long long p = p0 + q * PW + sub;
        float scale = 1.f;
        if (a.norm) {
            float ss = 0.f;
#pragma unroll
            for (int n = 0; n < NV; ++n)       // explicit fma chain: the sum does not depend on what the compiler chooses to contract in this build
                ss = fmaf(v[q][n].w, v[q][n].w, fmaf(v[q][n].z, v[q][n].z, fmaf(v[q][n].y, v[q][n].y, fmaf(v[q][n].x, v[q][n].x, ss))));
#pragma unroll
            for (int o = LPP / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
            scale = 1.0f / (1e-4f + sqrtf(ss) * rsqrtf((float)a.c));
        }
        if (p >= npix) continue;
        float4* dst = a.out ? reinterpret_cast<float4*>(a.out + (size_t)p * a.c) : nullptr;
        unsigned short* s8 = a.out_s8 ? static_cast<unsigned short*>(a.out_s8) + (size_t)p * a.c * 2 : nullptr;
        if (a.scale_out && li == 0) a.scale_out[p] = scale;
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int i = li + LPP * n;
            if (i >= c4) continue;
            float4 t = v[q][n];
            t.x *= scale; t.y *= scale; t.z *= scale; t.w *= scale;
            if (dst) dst[i] = t;
            if (s8) {
                // this float4 is one half of an 8-channel chunk [hi x8 | lo x8] and the lane next door (li ^ 1, same pixel: LPP and c/4 are
                // even) holds the other: the pair swaps one 8-byte piece through DPP (quad_perm [1,0,3,2]) so that the even lane writes the
                // hi half and the odd lane the lo half - one 16-byte store per lane instead of two 8-byte ones
                unsigned H0, H1, L0, L1;
                split_bf16_pair(mp_silu_dev(t.x), mp_silu_dev(t.y), H0, L0);
                split_bf16_pair(mp_silu_dev(t.z), mp_silu_dev(t.w), H1, L1);
                const bool odd = i & 1;
                const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? H0 : L0), 0xB1, 0xF, 0xF, true);
                const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? H1 : L1), 0xB1, 0xF, 0xF, true);
                unsigned short* qd = s8 + (size_t)(i >> 1) * 16 + (odd ? 8 : 0);
                *reinterpret_cast<uint4*>(qd) = odd ? make_uint4(r0, r1, L0, L1) : make_uint4(H0, H1, r0, r1);
            }
        }
    }
}

// ---------------------------------------------------------------- fp32 -> S8 split (+concat, scale, silu)
// one thread per 8-channel chunk of one pixel
__global__ __launch_bounds__(256) void split_k(vh_split_args a, long long total, unsigned div_mul, unsigned div_shr) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int nch = a.c_pad >> 3;
    // chunk index -> (pixel, chunk of the pixel): a multiply-high when the index fits 31 bits (every launch of the networks here), the
    // 64-bit division otherwise
    const long long pix = div_mul ? (long long)(__umulhi((unsigned)i, div_mul) >> div_shr) : i / nch;
    const int c = (int)(i - pix * nch) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    const float* sp = nullptr;
    float sc = 1.f;
    if (c < a.c0) { sp = a.src0 + (size_t)pix * a.c0 + c; sc = a.scale0; }
    else if (a.src1 && c - a.c0 < a.c1) { sp = a.src1 + (size_t)pix * a.c1 + (c - a.c0); sc = a.scale1; }
    if (sp) {
        const float4 p0 = *reinterpret_cast<const float4*>(sp), p1 = *reinterpret_cast<const float4*>(sp + 4);
        v[0] = p0.x; v[1] = p0.y; v[2] = p0.z; v[3] = p0.w; v[4] = p1.x; v[5] = p1.y; v[6] = p1.z; v[7] = p1.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= sc;
    // output chunk index: dense, or inside rows of out_c_total channels from channel out_c_off on
    const size_t oi = a.out_c_total ? (size_t)pix * (size_t)(a.out_c_total >> 3) + (size_t)((a.out_c_off + c) >> 3) : (size_t)i;
    if (a.out_raw) store_s8_chunk(reinterpret_cast<uint4*>(static_cast<unsigned short*>(a.out_raw) + oi * 16), v);
    if (!a.out) return;
    if (a.pro == VH_PRO_SILU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = mp_silu_dev(v[j]);
    }
    store_s8_chunk(reinterpret_cast<uint4*>(static_cast<unsigned short*>(a.out) + oi * 16), v);
}

// ---------------------------------------------------------------- q/k/v split + head norm
// a group of D lanes handles one (row, s, head)
template <int D>
__global__ __launch_bounds__(256) void qkv_split_k(vh_qkv_split_args a, long long ngroups) {
    constexpr int GPB = 256 / D;
    const long long gid = (long long)blockIdx.x * GPB + threadIdx.x / D;
    if (gid >= ngroups) return;     // D divides 64: a whole shuffle group exits together
    const int d = threadIdx.x % D;
    const int head = (int)(gid % a.heads);
    const long long pix = gid / a.heads;
    const int s = (int)(pix % a.s), row = (int)(pix / a.s);
    const int bb = row / a.rows_per_b, seg = row - bb * a.rows_per_b;
    const float* in = a.in + ((size_t)pix * a.heads * D + (size_t)head * D + d) * a.nj;
    const float rsd = rsqrtf((float)D);
    for (int j = 0; j < a.nj; ++j) {
        const float v = in[j];
        float ss = v * v;
#pragma unroll
        for (int o = D / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        float y = v / (1e-4f + sqrtf(ss) * rsd);
        const bool is_q = (a.nj == 3 && j == 0);
        if (is_q) {
            a.q[(((size_t)bb * a.heads + head) * a.s + s) * D + d] = y * a.qscale;
        } else {
            float* dst = ((a.nj == 3) ? (j == 1) : (j == 0)) ? a.k : a.v;
            dst[(((size_t)bb * a.heads + head) * a.kl + a.koff + seg * a.s + s) * D + d] = y;
        }
    }
}

// ---------------------------------------------------------------- embedding
__global__ __launch_bounds__(256) void embed_k(vh_embed_args a) {
    __shared__ float sf[1024];
    __shared__ float sg[64];
    const int r = blockIdx.x, t = threadIdx.x;
    const float sigma = a.sigma[(size_t)r * a.sigma_stride];
    const float cn = (logf(sigma) * 0.25f) * a.time_scale;
    for (int i = t; i < a.cnoise; i += 256) {
        const float ph = __fadd_rn(__fmul_rn(cn, a.freqs[i]), a.phases[i]);
        sf[i] = cosf(ph) * 1.41421356237309515f;
    }
    const bool lab = !a.raw && a.geometry && a.label_dim > 0;
    if (lab)
        for (int i = t; i < a.label_dim; i += 256) sg[i] = a.geometry[(size_t)r * a.label_dim + i] * a.geometry_scale;
    __syncthreads();
    const float tb = a.label_balance;
    const float inv = rsqrtf((1.f - tb) * (1.f - tb) + tb * tb);
    for (int co = t; co < a.cemb; co += 256) {
        float e1 = 0.f;
        for (int k = 0; k < a.cnoise; ++k) e1 += sf[k] * a.w_noise[((size_t)(k >> 2) * a.cemb + co) * 4 + (k & 3)];
        float e = e1;
        if (lab) {
            float e2 = 0.f;
            for (int k = 0; k < a.label_dim; ++k) e2 += sg[k] * a.w_label[((size_t)(k >> 2) * a.cemb + co) * 4 + (k & 3)];
            e = (e1 + tb * (e2 - e1)) * inv;
        }
        a.emb[(size_t)r * a.cemb + co] = a.raw ? e : mp_silu_dev(e);
    }
}

// ---------------------------------------------------------------- small batched linear
__global__ __launch_bounds__(256) void linear_k(vh_linear_args a) {
    __shared__ float se[1024];
    const int r = blockIdx.y, t = threadIdx.x;
    for (int i = t; i < a.cemb; i += 256) se[i] = a.emb[(size_t)r * a.cemb + i];
    __syncthreads();
    const int o = blockIdx.x * 256 + t;
    if (o >= a.cols) return;
    const float4* w = reinterpret_cast<const float4*>(a.wt);
    float acc = 0.f;
    for (int k4 = 0; k4 < (a.cemb >> 2); ++k4) {
        const float4 wv = w[(size_t)k4 * a.cols + o];
        acc += se[4 * k4] * wv.x + se[4 * k4 + 1] * wv.y + se[4 * k4 + 2] * wv.z + se[4 * k4 + 3] * wv.w;
    }
    a.out[(size_t)r * a.cols + o] = acc + a.bias;
}

// ---------------------------------------------------------------- input assembly
__global__ __launch_bounds__(256) void assemble_k(vh_assemble_args a, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % a.c_pad);
    const long long pix = i / a.c_pad;
    const int hw = a.h * a.w;
    const int r = (int)(pix / hw), p = (int)(pix - (long long)r * hw);
    float v = 0.f;
    int c0 = 0;
    bool done = false;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (s < a.nseg && !done) {
            const vh_segment sg = a.seg[s];
            if (c < c0 + sg.c) {
                const int cc = c - c0;
                const size_t sr = (size_t)r * sg.row_mul;
                v = sg.kind == 0 ? sg.ptr[(sr * sg.c_src + cc) * hw + p] : sg.ptr[(sr * hw + p) * sg.c_src + cc];
                if (sg.scale_cin) {
                    const float sig = a.sigma[sr];
                    v *= 1.0f / sqrtf(a.sigma_data * a.sigma_data + sig * sig);
                }
                done = true;
            }
            c0 += sg.c;
        }
    }
    if (!done && c == c0) v = 1.f;      // the constant-ones (bias) channel
    a.out[i] = v;
}

// ---------------------------------------------------------------- preconditioned output
__global__ __launch_bounds__(256) void precond_out_k(vh_precond_out_args a, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int hw = a.h * a.w;
    const int p = (int)(i % hw);
    const long long rc = i / hw;
    const int c = (int)(rc % a.c), r = (int)(rc / a.c);
    const size_t sr = (size_t)r * a.row_mul;
    const float sig = a.sigma[sr];
    const float sd = a.sigma_data;
    const float den = sig * sig + sd * sd;
    const float c_skip = sd * sd / den;
    const float c_out = sig * sd / sqrtf(den);
    const float x = a.x[(sr * a.c + c) * hw + p];
    const float f = a.f[((size_t)r * hw + p) * a.fc + c];
    a.out[i] = c_skip * x + c_out * f;
}

// ---------------------------------------------------------------- depth-warp Fourier features
// thread per (row, pixel, channel 0..127): channel = 64*axis + k
__global__ __launch_bounds__(256) void warp_features_k(vh_warp_args a, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ch = (int)(i & 127);
    const long long pix = i >> 7;
    const int ss = a.s * a.s;
    const int r = (int)(pix / ss), p = (int)(pix - (long long)r * ss);
    const int yi = p / a.s, xi = p - yi * a.s;
    if (a.nonzero_flag && *a.nonzero_flag == 0.f) {      // all-zero source: zero grids (training/models.py:647-648)
        a.grid_feat[i] = 0.f;
        a.warp_feat[i] = 0.f;
        if (a.uv_out && ch < 2) a.uv_out[pix * 2 + ch] = 0.f;
        return;
    }
    const int axis = ch >> 6, kf = ch & 63;
    const float g0 = yi + 0.5f, g1 = xi + 0.5f;     // meshgrid 'ij': coordinate 0 = row index
    float g[20];
#pragma unroll
    for (int j = 0; j < 20; ++j) g[j] = a.geometry[(size_t)r * 20 + j] * a.std[j] + a.mean[j];
    // unproject with K_src: K^-1 [g0, g1, 1]
    const float depth = a.depth[((size_t)r * a.src_c + a.depth_ch) * ss + p];
    const float X = (g0 - g[14]) / g[12] * depth, Y = (g1 - g[15]) / g[13] * depth, Z = depth;
    // inverse of [R t; 0 1]: R^-1 (w - t)
    const float r00 = g[0], r01 = g[1], r02 = g[2], t0 = g[3];
    const float r10 = g[4], r11 = g[5], r12 = g[6], t1 = g[7];
    const float r20 = g[8], r21 = g[9], r22 = g[10], t2 = g[11];
    const float c00 = r11 * r22 - r12 * r21, c01 = r02 * r21 - r01 * r22, c02 = r01 * r12 - r02 * r11;
    const float c10 = r12 * r20 - r10 * r22, c11 = r00 * r22 - r02 * r20, c12 = r02 * r10 - r00 * r12;
    const float c20 = r10 * r21 - r11 * r20, c21 = r01 * r20 - r00 * r21, c22 = r00 * r11 - r01 * r10;
    const float idet = 1.0f / (r00 * c00 + r01 * c10 + r02 * c20);
    const float dx = X - t0, dy = Y - t1, dz = Z - t2;
    const float wx = (c00 * dx + c01 * dy + c02 * dz) * idet;
    const float wy = (c10 * dx + c11 * dy + c12 * dz) * idet;
    const float wz = (c20 * dx + c21 * dy + c22 * dz) * idet;
    // project with K_tgt and divide
    float u = (g[16] * wx + g[18] * wz) / wz;
    float v = (g[17] * wy + g[19] * wz) / wz;
    if (u != u) u = 0.f;
    if (v != v) v = 0.f;
    if (a.uv_out && ch == 0) { a.uv_out[pix * 2] = u; a.uv_out[pix * 2 + 1] = v; }
    const float f = a.freqs[kf], ph = a.phases[kf];
    const float cg = axis == 0 ? g0 : g1;
    const float cw = axis == 0 ? u : v;
    a.grid_feat[i] = cosf(__fadd_rn(__fmul_rn(cg, f), ph)) * 1.41421356237309515f;
    a.warp_feat[i] = cosf(__fadd_rn(__fmul_rn(cw, f), ph)) * 1.41421356237309515f;
}

// ---------------------------------------------------------------- sampler update
__global__ __launch_bounds__(256) void sampler_step_k(vh_sampler_step_args a, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long r = i / (long long)a.row_elems, e = i - r * (long long)a.row_elems;
    const size_t xi = (size_t)r * a.row_mul * a.row_elems + e;
    float D = a.d_cond[i];
    if (a.d_ref) {
        const float ref = a.d_ref[i];
        D = ref + a.guidance * (D - ref);
    }
    const float xh = a.x_hat[xi];
    float xn;
    if (!a.x_probe) {
        const float d = (xh - D) / a.t_hat;
        a.d_cur[i] = d;
        xn = xh + (a.t_next - a.t_hat) * d;
    } else {
        const float dp = (a.x_probe[xi] - D) / a.t_next;
        xn = xh + (a.t_next - a.t_hat) * (0.5f * a.d_cur[i] + 0.5f * dp);
    }
    for (int j = 0; j < a.row_mul; ++j) a.x_next[xi + (size_t)j * a.row_elems] = xn;
}

// ---------------------------------------------------------------- pixel codec
__global__ __launch_bounds__(256) void codec_k(vh_codec_args a) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    if (!a.decode) {
        static_cast<float*>(a.out)[i] = static_cast<const unsigned char*>(a.in)[i] / 127.5f - 1.0f;
    } else {
        const float v = fminf(fmaxf(static_cast<const float*>(a.in)[i] * 127.5f + 128.0f, 0.0f), 255.0f);
        static_cast<unsigned char*>(a.out)[i] = (unsigned char)v;
    }
}

// ---------------------------------------------------------------- add_depth: one workgroup per sample
__global__ __launch_bounds__(256) void add_depth_k(vh_add_depth_args a) {
    __shared__ float red[4];
    const int r = blockIdx.x, t = threadIdx.x;
    const int hw = a.h * a.w;
    const float* d = a.depth + (size_t)r * hw;
    float* o = a.out + (size_t)r * (a.c + 1) * hw;
    const float* s = a.src + (size_t)r * a.c * hw;
    for (int i = t; i < a.c * hw; i += 256) o[i] = s[i];
    float mx = -INFINITY;
    if (a.inv_norm) {
        for (int i = t; i < hw; i += 256) mx = fmaxf(mx, 1.0f / d[i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if ((t & 63) == 0) red[t >> 6] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
    for (int i = t; i < hw; i += 256) {
        float v = d[i];
        if (a.inv_norm) v = ((1.0f / v) / mx - 0.4947f) / 0.2294f;
        o[(size_t)a.c * hw + i] = v;
    }
}

// ---------------------------------------------------------------- bilinear resize (optionally anti-aliased)
__device__ __forceinline__ float tri(float x) { x = fabsf(x); return x < 1.f ? 1.f - x : 0.f; }

// cubic convolution weights of the 4 taps at offsets -1, 0, 1, 2 for fraction t (A = -0.75: aten's get_cubic_upsample_coefficients)
__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

__global__ __launch_bounds__(256) void resize_k(vh_resize_args a, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int xo = (int)(i % a.wout);
    const long long r = i / a.wout;
    const int yo = (int)(r % a.hout);
    const long long pl = r / a.hout;
    const float sy = (float)a.hin / a.hout, sx = (float)a.win / a.wout;
    const float* src = a.in + (size_t)pl * a.hin * a.win;
    float acc = 0.f;
    const float ay = a.hout > 1 ? (float)(a.hin - 1) / (a.hout - 1) : 0.f, ax = a.wout > 1 ? (float)(a.win - 1) / (a.wout - 1) : 0.f;
    if (a.mode == VH_RESIZE_BICUBIC) {
        // aten upsample_bicubic2d: 4x4 taps around floor(coordinate), indices clamped to the border, A = -0.75
        const float fy = a.align_corners ? yo * ay : (yo + 0.5f) * sy - 0.5f;
        const float fx = a.align_corners ? xo * ax : (xo + 0.5f) * sx - 0.5f;
        const float fy0 = floorf(fy), fx0 = floorf(fx);
        const float ty = fy - fy0, tx = fx - fx0;
        const int iy = (int)fy0, ix = (int)fx0;
        float wy[4], wx[4];
        cubic_coeffs(ty, wy);
        cubic_coeffs(tx, wx);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int y = min(max(iy - 1 + j, 0), a.hin - 1);
            float row = 0.f;
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) row += wx[i2] * src[y * a.win + min(max(ix - 1 + i2, 0), a.win - 1)];
            acc += wy[j] * row;
        }
    } else if (!a.antialias) {
        float fy = a.align_corners ? yo * ay : (yo + 0.5f) * sy - 0.5f, fx = a.align_corners ? xo * ax : (xo + 0.5f) * sx - 0.5f;
        fy = fmaxf(fy, 0.f); fx = fmaxf(fx, 0.f);
        const int y0 = min((int)fy, a.hin - 1), x0 = min((int)fx, a.win - 1);
        const int y1 = min(y0 + 1, a.hin - 1), x1 = min(x0 + 1, a.win - 1);
        const float ly = fy - y0, lx = fx - x0;
        acc = (1.f - ly) * ((1.f - lx) * src[y0 * a.win + x0] + lx * src[y0 * a.win + x1]) +
              ly * ((1.f - lx) * src[y1 * a.win + x0] + lx * src[y1 * a.win + x1]);
    } else {
        // aten upsample_bilinear2d_aa: support = max(scale, 1); taps [xmin, xmin+xsize), weights tri((j+xmin-center+0.5)/s)
        const float supy = fmaxf(sy, 1.f), supx = fmaxf(sx, 1.f);
        const float cy = sy * (yo + 0.5f), cx = sx * (xo + 0.5f);
        const int ymin = max((int)(cy - supy + 0.5f), 0), ymax = min((int)(cy + supy + 0.5f), a.hin);
        const int xmin = max((int)(cx - supx + 0.5f), 0), xmax = min((int)(cx + supx + 0.5f), a.win);
        float wy_sum = 0.f, wx_sum = 0.f;
        for (int y = ymin; y < ymax; ++y) wy_sum += tri((y - cy + 0.5f) / supy);
        for (int x = xmin; x < xmax; ++x) wx_sum += tri((x - cx + 0.5f) / supx);
        for (int y = ymin; y < ymax; ++y) {
            const float wy = tri((y - cy + 0.5f) / supy) / wy_sum;
            float row = 0.f;
            for (int x = xmin; x < xmax; ++x) row += tri((x - cx + 0.5f) / supx) / wx_sum * src[y * a.win + x];
            acc += wy * row;
        }
    }
    if (a.ch_scale) {
        const int c = (int)(pl % a.channels);
        acc = acc * a.ch_scale[c] + a.ch_bias[c];
    }
    a.out[i] = acc;
}

inline unsigned blocks_for(long long n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

extern "C" int vh_prep_weight(vh_ctx* ctx, const vh_prep_weight_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_prep_weight: null args");
    const vh_prep_weight_args a = *p;
    VH_REQUIRE(a.w && a.wt, "vh_prep_weight: null tensor");
    VH_REQUIRE(a.cout > 0 && a.cin > 0 && (a.taps == 1 || a.taps == 9), "vh_prep_weight: bad shape");
    VH_REQUIRE(a.cin_pad >= a.cin && a.cin_pad % 4 == 0, "vh_prep_weight: cin_pad %d", a.cin_pad);
    VH_REQUIRE(a.k_pad % 32 == 0 && a.k_pad >= a.taps * a.cin_pad, "vh_prep_weight: k_pad %d", a.k_pad);
    VH_REQUIRE(a.dst_col0 >= 0 && a.dst_col0 + a.cout <= a.dst_cols, "vh_prep_weight: destination columns out of range");
    VH_REQUIRE((a.k_off == 0 && a.k_stride == 0) || (a.split == 2 && a.k_off >= 0 && a.k_off % 32 == 0 && a.k_stride % 32 == 0 && a.k_off + a.k_pad <= a.k_stride),
               "vh_prep_weight: k_off / k_stride place a K range inside a longer row of a split = 2 weight (multiples of 32, k_off + k_pad <= k_stride)");
    return vh_dispatch(ctx, VH_TAG_PREP, 0.0, 4.0 * ((double)a.cout * a.cin * a.taps + (double)a.k_pad * a.cout), [a](hipStream_t s) -> int {
        hipLaunchKernelGGL(prep_weight_k, dim3(a.cout), dim3(256), 0, s, a);
        return vh_check_launch("prep_weight_k");
    });
}

extern "C" int vh_split(vh_ctx* ctx, const vh_split_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_split: null args");
    const vh_split_args a = *p;
    VH_REQUIRE(a.src0 && (a.out || a.out_raw), "vh_split: null tensor");
    VH_REQUIRE(a.c0 > 0 && a.c0 % 8 == 0, "vh_split: c0 must be a positive multiple of 8 (got %d)", a.c0);
    VH_REQUIRE(a.src1 ? (a.c1 > 0 && a.c1 % 8 == 0) : a.c1 == 0, "vh_split: bad c1 %d", a.c1);
    VH_REQUIRE(a.c_pad % 32 == 0 && a.c_pad >= a.c0 + a.c1, "vh_split: c_pad %d must be a multiple of 32 >= %d", a.c_pad, a.c0 + a.c1);
    VH_REQUIRE(a.npix > 0 && (a.pro == VH_PRO_NONE || a.pro == VH_PRO_SILU), "vh_split: bad arguments");
    VH_REQUIRE((a.out_c_total == 0 && a.out_c_off == 0) || (a.out_c_total % 32 == 0 && a.out_c_off % 32 == 0 && a.out_c_off >= 0 && a.out_c_off + a.c_pad <= a.out_c_total),
               "vh_split: out_c_total / out_c_off must be multiples of 32 with out_c_off + c_pad <= out_c_total (got %d, %d, c_pad %d)", a.out_c_total, a.out_c_off, a.c_pad);
    VH_REQUIRE(vh_aligned16(a.src0) && vh_aligned16(a.src1) && vh_aligned16(a.out) && vh_aligned16(a.out_raw), "vh_split: pointers must be 16-byte aligned");
    const long long total = a.npix * (a.c_pad / 8);
    return vh_dispatch(ctx, VH_TAG_SPLIT, 0.0, 4.0 * (double)a.npix * ((double)a.c0 + a.c1 + a.c_pad * ((a.out ? 1 : 0) + (a.out_raw ? 1 : 0))), [a, total](hipStream_t s) -> int {
        // magic numbers of n / (c_pad / 8) for 0 <= n < 2^31 (as conv_common.h's FastDiv); 0 = take the division
        unsigned mul = 0, shr = 0;
        const unsigned d = (unsigned)(a.c_pad / 8);
        if (total < (1ll << 31)) {                     // (d >= 4: c_pad is a multiple of 32)
            unsigned l = 0;
            while ((1ull << l) < d) ++l;
            mul = (unsigned)(((1ull << (31 + l)) / d) + 1); shr = l - 1;
        }
        hipLaunchKernelGGL(split_k, dim3(blocks_for(total, 256)), dim3(256), 0, s, a, total, mul, shr);
        return vh_check_launch("split_k");
    });
}

extern "C" int vh_pixnorm(vh_ctx* ctx, const vh_pixnorm_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_pixnorm: null args");
    const vh_pixnorm_args a = *p;
    VH_REQUIRE(a.in && (a.out || (a.out_s8 && a.scale_out && a.norm)), "vh_pixnorm: null tensor (out may be NULL only when out_s8 and scale_out are given)");
    VH_REQUIRE(a.rows > 0 && a.h > 0 && a.w > 0 && a.c > 0 && a.c % 4 == 0, "vh_pixnorm: bad geometry (c must be a multiple of 4)");
    VH_REQUIRE(vh_aligned16(a.in) && vh_aligned16(a.out) && vh_aligned16(a.out_s8), "vh_pixnorm: pointers must be 16-byte aligned");
    VH_REQUIRE(!a.out_s8 || a.c % 32 == 0, "vh_pixnorm: S8 output needs c %% 32 == 0 (got %d)", a.c);
    const long long npix = (long long)a.rows * a.h * a.w;
    return vh_dispatch(ctx, VH_TAG_PIXNORM, 0.0, 4.0 * (double)npix * a.c * ((a.pool ? 4.0 : 1.0) + (a.out ? 1.0 : 0.0) + (a.out_s8 ? 1.0 : 0.0)), [a, npix](hipStream_t s) -> int {
        const int c4 = a.c >> 2;
        const bool inplace_pool = a.pool && a.in == a.out;       // (never used by the engine; the register form reads neighbours late)
#define VH_PIX(NV_, LPP_)                                                                                                         \
        do {                                                                                                                      \
            constexpr int PW = 64 / LPP_;                                                                                         \
            if (a.pool) hipLaunchKernelGGL((pixnorm_reg_k<NV_, LPP_, true, 2>), dim3(blocks_for(npix, 8 * PW)), dim3(256), 0, s, a, npix);  \
            else hipLaunchKernelGGL((pixnorm_reg_k<NV_, LPP_, false, 4>), dim3(blocks_for(npix, 16 * PW)), dim3(256), 0, s, a, npix);        \
        } while (0)
        if (inplace_pool || c4 > 128) hipLaunchKernelGGL(pixnorm_k, dim3(blocks_for(npix, 4)), dim3(256), 0, s, a, npix);
        else if (c4 <= 16) VH_PIX(1, 16);
        else if (c4 <= 32) VH_PIX(1, 32);
        else if (c4 <= 64) VH_PIX(1, 64);
        else VH_PIX(2, 64);
#undef VH_PIX
        return vh_check_launch("pixnorm_k");
    });
}

extern "C" int vh_qkv_split(vh_ctx* ctx, const vh_qkv_split_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_qkv_split: null args");
    const vh_qkv_split_args a = *p;
    VH_REQUIRE(a.in && a.k && a.v, "vh_qkv_split: null tensor");
    VH_REQUIRE(a.nj == 2 || (a.nj == 3 && a.q), "vh_qkv_split: nj must be 2 or 3 (3 needs q)");
    VH_REQUIRE(a.d == 32 || a.d == 64, "vh_qkv_split: head dim %d unsupported (32 or 64)", a.d);
    VH_REQUIRE(a.rows > 0 && a.s > 0 && a.heads > 0 && a.rows_per_b > 0 && a.rows % a.rows_per_b == 0, "vh_qkv_split: bad geometry");
    VH_REQUIRE(a.koff >= 0 && a.koff + a.rows_per_b * a.s <= a.kl, "vh_qkv_split: keys do not fit (koff %d + %d*%d > kl %d)", a.koff, a.rows_per_b, a.s, a.kl);
    const long long ng = (long long)a.rows * a.s * a.heads;
    const int d = a.d;
    return vh_dispatch(ctx, VH_TAG_QKVSPLIT, 0.0, 8.0 * (double)ng * d * a.nj, [a, ng, d](hipStream_t s) -> int {
        if (d == 64) hipLaunchKernelGGL(qkv_split_k<64>, dim3(blocks_for(ng, 4)), dim3(256), 0, s, a, ng);
        else hipLaunchKernelGGL(qkv_split_k<32>, dim3(blocks_for(ng, 8)), dim3(256), 0, s, a, ng);
        return vh_check_launch("qkv_split_k");
    });
}

extern "C" int vh_embed(vh_ctx* ctx, const vh_embed_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_embed: null args");
    const vh_embed_args a = *p;
    VH_REQUIRE(a.sigma && a.freqs && a.phases && a.w_noise && a.emb, "vh_embed: null tensor");
    VH_REQUIRE(a.cnoise > 0 && a.cnoise <= 1024 && a.cnoise <= a.w_noise_kpad, "vh_embed: cnoise %d", a.cnoise);
    VH_REQUIRE(a.label_dim >= 0 && a.label_dim <= 64, "vh_embed: label_dim %d", a.label_dim);
    VH_REQUIRE(!(a.geometry && a.label_dim > 0) || (a.w_label && a.label_dim <= a.w_label_kpad), "vh_embed: w_label missing");
    VH_REQUIRE(a.rows > 0 && a.cemb > 0, "vh_embed: bad geometry");
    return vh_dispatch(ctx, VH_TAG_EMBED, 2.0 * a.rows * (double)a.cemb * (a.cnoise + a.label_dim), 4.0 * (double)a.cemb * (a.cnoise + a.label_dim + a.rows), [a](hipStream_t s) -> int {
        hipLaunchKernelGGL(embed_k, dim3(a.rows), dim3(256), 0, s, a);
        return vh_check_launch("embed_k");
    });
}

extern "C" int vh_linear(vh_ctx* ctx, const vh_linear_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_linear: null args");
    const vh_linear_args a = *p;
    VH_REQUIRE(a.emb && a.wt && a.out, "vh_linear: null tensor");
    VH_REQUIRE(a.cemb > 0 && a.cemb <= 1024 && a.cemb % 4 == 0 && a.cemb <= a.k_pad, "vh_linear: cemb %d", a.cemb);
    VH_REQUIRE(a.rows > 0 && a.rows < 65536 && a.cols > 0, "vh_linear: bad geometry");
    VH_REQUIRE(vh_aligned16(a.wt), "vh_linear: weights must be 16-byte aligned");
    return vh_dispatch(ctx, VH_TAG_EMBED, 2.0 * a.rows * (double)a.cemb * a.cols, 4.0 * ((double)a.cemb * a.cols + (double)a.rows * (a.cols + a.cemb)), [a](hipStream_t s) -> int {
        hipLaunchKernelGGL(linear_k, dim3(blocks_for(a.cols, 256), a.rows), dim3(256), 0, s, a);
        return vh_check_launch("linear_k");
    });
}

extern "C" int vh_assemble(vh_ctx* ctx, const vh_assemble_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_assemble: null args");
    const vh_assemble_args a = *p;
    VH_REQUIRE(a.out && a.nseg >= 1 && a.nseg <= 4, "vh_assemble: bad args");
    int ctot = 0;
    bool need_sigma = false;
    for (int i = 0; i < a.nseg; ++i) {
        VH_REQUIRE(a.seg[i].ptr && a.seg[i].c > 0 && a.seg[i].c_src >= a.seg[i].c && a.seg[i].row_mul >= 1 && (a.seg[i].kind == 0 || a.seg[i].kind == 1), "vh_assemble: bad segment %d", i);
        ctot += a.seg[i].c;
        need_sigma |= a.seg[i].scale_cin != 0;
    }
    VH_REQUIRE(!need_sigma || a.sigma, "vh_assemble: sigma missing");
    VH_REQUIRE(a.c_pad >= ctot + 1 && a.c_pad % 4 == 0, "vh_assemble: c_pad %d too small for %d channels + ones", a.c_pad, ctot);
    VH_REQUIRE(a.rows > 0 && a.h > 0 && a.w > 0, "vh_assemble: bad geometry");
    const long long total = (long long)a.rows * a.h * a.w * a.c_pad;
    return vh_dispatch(ctx, VH_TAG_ASSEMBLE, 0.0, 8.0 * (double)total, [a, total](hipStream_t s) -> int {
        hipLaunchKernelGGL(assemble_k, dim3(blocks_for(total, 256)), dim3(256), 0, s, a, total);
        return vh_check_launch("assemble_k");
    });
}

extern "C" int vh_precond_out(vh_ctx* ctx, const vh_precond_out_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_precond_out: null args");
    const vh_precond_out_args a = *p;
    VH_REQUIRE(a.x && a.f && a.sigma && a.out, "vh_precond_out: null tensor");
    VH_REQUIRE(a.rows > 0 && a.c > 0 && a.h > 0 && a.w > 0 && a.fc >= a.c && a.row_mul >= 1, "vh_precond_out: bad geometry");
    const long long total = (long long)a.rows * a.c * a.h * a.w;
    return vh_dispatch(ctx, VH_TAG_ASSEMBLE, 0.0, 12.0 * (double)total, [a, total](hipStream_t s) -> int {
        hipLaunchKernelGGL(precond_out_k, dim3(blocks_for(total, 256)), dim3(256), 0, s, a, total);
        return vh_check_launch("precond_out_k");
    });
}

extern "C" int vh_warp_features(vh_ctx* ctx, const vh_warp_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_warp_features: null args");
    const vh_warp_args a = *p;
    VH_REQUIRE(a.depth && a.geometry && a.freqs && a.phases && a.grid_feat && a.warp_feat, "vh_warp_features: null tensor");
    VH_REQUIRE(a.rows > 0 && a.s > 0 && a.depth_ch >= 0 && a.depth_ch < a.src_c, "vh_warp_features: bad geometry");
    const long long total = (long long)a.rows * a.s * a.s * 128;
    return vh_dispatch(ctx, VH_TAG_WARP, 0.0, 8.0 * (double)total, [a, total](hipStream_t s) -> int {
        hipLaunchKernelGGL(warp_features_k, dim3(blocks_for(total, 256)), dim3(256), 0, s, a, total);
        return vh_check_launch("warp_features_k");
    });
}

extern "C" int vh_codec(vh_ctx* ctx, const vh_codec_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_codec: null args");
    const vh_codec_args a = *p;
    VH_REQUIRE(a.in && a.out && a.n > 0, "vh_codec: bad arguments");
    return vh_dispatch(ctx, VH_TAG_ASSEMBLE, 0.0, 5.0 * (double)a.n, [a](hipStream_t s) -> int {
        hipLaunchKernelGGL(codec_k, dim3(blocks_for((long long)a.n, 256)), dim3(256), 0, s, a);
        return vh_check_launch("codec_k");
    });
}

extern "C" int vh_add_depth(vh_ctx* ctx, const vh_add_depth_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_add_depth: null args");
    const vh_add_depth_args a = *p;
    VH_REQUIRE(a.src && a.depth && a.out, "vh_add_depth: null tensor");
    VH_REQUIRE(a.rows > 0 && a.c > 0 && a.h > 0 && a.w > 0, "vh_add_depth: bad geometry");
    return vh_dispatch(ctx, VH_TAG_ASSEMBLE, 0.0, 8.0 * (double)a.rows * (a.c + 1) * a.h * a.w, [a](hipStream_t s) -> int {
        hipLaunchKernelGGL(add_depth_k, dim3(a.rows), dim3(256), 0, s, a);
        return vh_check_launch("add_depth_k");
    });
}

extern "C" int vh_resize(vh_ctx* ctx, const vh_resize_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_resize: null args");
    const vh_resize_args a = *p;
    VH_REQUIRE(a.in && a.out, "vh_resize: null tensor");
    VH_REQUIRE(a.planes > 0 && a.hin > 0 && a.win > 0 && a.hout > 0 && a.wout > 0, "vh_resize: bad geometry");
    VH_REQUIRE(a.mode == VH_RESIZE_BILINEAR || a.mode == VH_RESIZE_BICUBIC, "vh_resize: bad mode %d", a.mode);
    VH_REQUIRE(!a.antialias || (a.mode == VH_RESIZE_BILINEAR && !a.align_corners), "vh_resize: antialias exists for bilinear, align_corners = 0 only");
    VH_REQUIRE(!a.ch_scale || (a.ch_bias && a.channels > 0 && a.planes % a.channels == 0), "vh_resize: per-channel affine needs ch_bias and channels dividing planes");
    const long long total = (long long)a.planes * a.hout * a.wout;
    return vh_dispatch(ctx, VH_TAG_ASSEMBLE, 0.0, 4.0 * ((double)total + (double)a.planes * a.hin * a.win), [a, total](hipStream_t s) -> int {
        hipLaunchKernelGGL(resize_k, dim3(blocks_for(total, 256)), dim3(256), 0, s, a, total);
        return vh_check_launch("resize_k");
    });
}

extern "C" int vh_resize_bilinear(vh_ctx* ctx, const vh_resize_args* p) {
    if (p && p->mode != VH_RESIZE_BILINEAR) return vh_fail(VH_EINVAL, "vh_resize_bilinear: mode must be VH_RESIZE_BILINEAR (use vh_resize)");
    return vh_resize(ctx, p);
}

extern "C" int vh_sampler_step(vh_ctx* ctx, const vh_sampler_step_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_sampler_step: null args");
    const vh_sampler_step_args a = *p;
    VH_REQUIRE(a.x_hat && a.d_cond && a.d_cur && a.x_next, "vh_sampler_step: null tensor");
    VH_REQUIRE(a.rows > 0 && a.row_mul >= 1 && a.row_elems > 0, "vh_sampler_step: bad geometry");
    VH_REQUIRE(a.t_hat != 0.f && (!a.x_probe || a.t_next != 0.f), "vh_sampler_step: division by a zero noise level");
    const long long total = (long long)a.rows * (long long)a.row_elems;
    return vh_dispatch(ctx, VH_TAG_SAMPLER, 0.0, 4.0 * (double)total * (3.0 + a.row_mul + (a.d_ref ? 1.0 : 0.0)), [a, total](hipStream_t s) -> int {
        hipLaunchKernelGGL(sampler_step_k, dim3(blocks_for(total, 256)), dim3(256), 0, s, a, total);
        return vh_check_launch("sampler_step_k");
    });
}
