// Kernels beside the denoiser proper (gfx950): general 2x FIR resampling for non-default `resample_filter`s, the
// all-zero-source test behind the depth-warp shortcut, and the FID / PSNR statistics of calculate_metrics.py
// (fp64 feature moments on v_mfma_f64_16x16x4_f64, per-image PSNR sums).  Reference lines are cited at each
// entry point in include/vivid_hip.h.
#include "ctx.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));

namespace {

inline unsigned blocks_for(long long n, int per) { return (unsigned)((n + per - 1) / per); }

// ---------------------------------------------------------------- resample (training/models.py:48-61), NHWC
// down: out[y][x] = sum_{i,j} g[i] g[j] in[2y - pad + i][2x - pad + j]           g = f / sum(f), zero outside
// up  : out[Y][X] = sum_{y,i: 2y+i-pad = Y} sum_{x,j: 2x+j-pad = X} 4 g[i] g[j] in[y][x]
// one thread per (output pixel, 4 channels)
__global__ __launch_bounds__(256) void resample_k(vh_resample_args a, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c4n = a.c >> 2;
    const int c4 = (int)(idx % c4n);
    const long long pix = idx / c4n;
    const int ho = a.up ? a.h * 2 : a.h / 2, wo = a.up ? a.w * 2 : a.w / 2;
    const int xo = (int)(pix % wo);
    const long long r1 = pix / wo;
    const int yo = (int)(r1 % ho), img = (int)(r1 / ho);
    const int L = a.ntaps, pad = (L - 1) / 2;
    const float4* in = reinterpret_cast<const float4*>(a.in) + (size_t)img * a.h * a.w * c4n + c4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!a.up) {
        for (int i = 0; i < L; ++i) {
            const int y = 2 * yo - pad + i;
            if ((unsigned)y >= (unsigned)a.h) continue;
            for (int j = 0; j < L; ++j) {
                const int x = 2 * xo - pad + j;
                if ((unsigned)x >= (unsigned)a.w) continue;
                const float wgt = a.taps[i] * a.taps[j];
                const float4 v = in[((size_t)y * a.w + x) * c4n];
                acc.x += wgt * v.x; acc.y += wgt * v.y; acc.z += wgt * v.z; acc.w += wgt * v.w;
            }
        }
    } else {
        for (int i = (yo + pad) & 1; i < L; i += 2) {
            const int y2 = yo + pad - i;                       // = 2y
            if (y2 < 0 || (y2 >> 1) >= a.h) continue;
            for (int j = (xo + pad) & 1; j < L; j += 2) {
                const int x2 = xo + pad - j;
                if (x2 < 0 || (x2 >> 1) >= a.w) continue;
                const float wgt = 4.f * a.taps[i] * a.taps[j];
                const float4 v = in[((size_t)(y2 >> 1) * a.w + (x2 >> 1)) * c4n];
                acc.x += wgt * v.x; acc.y += wgt * v.y; acc.z += wgt * v.z; acc.w += wgt * v.w;
            }
        }
    }
    reinterpret_cast<float4*>(a.out)[idx] = acc;
}

// ---------------------------------------------------------------- "is this tensor all zero?" (training/models.py:647)
// flag[0] = 1.0f if any of the first `c_used` channels of any row of an NCHW tensor is non-zero (the flag is cleared by a
// memset node ahead of the launch; every thread that sees a non-zero value stores the same 1.0f)
__global__ __launch_bounds__(256) void nonzero_flag_k(vh_nonzero_args a, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long per_row = (long long)a.c_used * a.hw;
    const long long r = i / per_row, e = i - r * per_row;
    if (a.in[(size_t)r * a.c_total * a.hw + e] != 0.f) *a.flag = 1.0f;
}

// ---------------------------------------------------------------- fp64 feature moments (calculate_metrics.py:158-172)
// outer[fa][fb] += A^T B,  sum_a[fa] += column sums of A;  A [n][fa], B [n][fb] fp32 row-major, products and sums in fp64
// (exactly features.to(float64) of the reference).  One workgroup = 4 waves = a 64 x 64 block of `outer`; wave w owns rows
// 16w..16w+15 and four 16-column tiles: v_mfma_f64_16x16x4_f64, K = the batch rows, 4 per instruction.
// A fragment: lane l holds A[m = l&15][k = l>>4]; B: B[k = l>>4][n = l&15]; C/D: col = l&15, row = (l>>4) + 4*reg
// (cdna_hip_programming.md 3: the f64 map differs from the f32 one).
__global__ __launch_bounds__(256) void moments_k(vh_moments_args a) {
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const int i0 = blockIdx.y * 64 + w * 16, j0 = blockIdx.x * 64;
    const int m = l & 15, kq = l >> 4;
    f64x4 acc[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[jt][r] = 0.0;
    const int ia = i0 + m;
    for (int n0 = 0; n0 < a.n; n0 += 4) {
        const int n = n0 + kq;
        const bool nok = n < a.n;
        const double av = (nok && ia < a.fa) ? (double)a.a[(size_t)n * a.fa + ia] : 0.0;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const int jb = j0 + jt * 16 + m;
            const double bv = (nok && jb < a.fb) ? (double)a.b[(size_t)n * a.fb + jb] : 0.0;
            acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[jt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
        const int j = j0 + jt * 16 + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + kq + 4 * r;
            if (i < a.fa && j < a.fb) a.outer[(size_t)i * a.fb + j] += acc[jt][r];
        }
    }
    // column sums of A: the workgroups of the first block column add their 64 rows' worth
    if (a.sum_a && blockIdx.x == 0 && t < 64) {
        const int i = blockIdx.y * 64 + t;
        if (i < a.fa) {
            double s = 0.0;
            for (int n = 0; n < a.n; ++n) s += (double)a.a[(size_t)n * a.fa + i];
            a.sum_a[i] += s;
        }
    }
}

// ---------------------------------------------------------------- PSNR (calculate_metrics.py:147)
// per_image[i] = 10 log10(255^2 / mean((x_i - y_i)^2)); one workgroup per image, uint8 or fp32 inputs.  The per-image values are
// then added to acc[0] by ONE lane in index order (psnr_fold_k): the accumulator is bit-reproducible from run to run, which a
// double atomicAdd in arrival order was not.
template <class T>
__global__ __launch_bounds__(256) void psnr_k(const T* x, const T* y, long long elems, double* per_image) {
    __shared__ double red[4];
    const int t = threadIdx.x;
    const T* xi = x + (size_t)blockIdx.x * elems;
    const T* yi = y + (size_t)blockIdx.x * elems;
    double s = 0.0;
    for (long long i = t; i < elems; i += 256) {
        const float d = (float)xi[i] - (float)yi[i];
        s += (double)(d * d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((t & 63) == 0) red[t >> 6] = s;
    __syncthreads();
    if (t == 0) {
        const double mse = (red[0] + red[1] + red[2] + red[3]) / (double)elems;
        per_image[blockIdx.x] = 10.0 * log10(255.0 * 255.0 / mse);
    }
}

__global__ void psnr_fold_k(const double* per_image, int images, double* acc) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = acc[0];
        for (int i = 0; i < images; ++i) s += per_image[i];
        acc[0] = s;
    }
}

// ---------------------------------------------------------------- NHWC <-> NCHW (fp32), 32 x 32 tiles through LDS
// per row (image): in [hw][c] -> out [c][hw] (to_nchw) or the other way round; both sides coalesced
__global__ __launch_bounds__(256) void layout_k(vh_layout_args a, int tiles_a, int tiles_b) {
    __shared__ float tile[32][33];
    // A = the input's slow axis, B = its fast axis: to_nchw reads [hw][c] (A = hw, B = c), else [c][hw] (A = c, B = hw)
    const int nA = a.to_nchw ? a.hw : a.c, nB = a.to_nchw ? a.c : a.hw;
    const int t = blockIdx.x % (tiles_a * tiles_b), row = blockIdx.x / (tiles_a * tiles_b);
    const int ta = t / tiles_b, tb = t - ta * tiles_b;
    const float* in = a.in + (size_t)row * a.hw * a.c;
    float* out = a.out + (size_t)row * a.hw * a.c;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ia = ta * 32 + ly + 8 * k, ib = tb * 32 + lx;
        if (ia < nA && ib < nB) tile[ly + 8 * k][lx] = in[(size_t)ia * nB + ib];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ib = tb * 32 + ly + 8 * k, ia = ta * 32 + lx;
        if (ia < nA && ib < nB) out[(size_t)ib * nA + ia] = tile[lx][ly + 8 * k];
    }
}

// out = a + s * b with two roundings (torch's `a + s * b` is a multiply kernel and an add kernel); a == NULL: s * b; b == NULL: the constant s
__global__ __launch_bounds__(256) void axpy_k(vh_axpy_args a) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    float v = a.b ? __fmul_rn(a.s, a.b[i]) : a.s;
    if (a.a) v = __fadd_rn(a.a[i], v);
    a.out[i] = v;
}

}  // namespace

extern "C" int vh_layout(vh_ctx* ctx, const vh_layout_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_layout: null args");
    const vh_layout_args a = *p;
    VH_REQUIRE(a.in && a.out && a.in != a.out, "vh_layout: null tensor (or in == out)");
    VH_REQUIRE(a.rows > 0 && a.c > 0 && a.hw > 0, "vh_layout: bad geometry");
    const int nA = a.to_nchw ? a.hw : a.c, nB = a.to_nchw ? a.c : a.hw;
    const int tiles_a = (nA + 31) / 32, tiles_b = (nB + 31) / 32;
    VH_REQUIRE((long long)a.rows * tiles_a * tiles_b < (1LL << 31), "vh_layout: grid too large");
    return vh_dispatch(ctx, VH_TAG_ASSEMBLE, 0.0, 8.0 * (double)a.rows * a.c * a.hw, [a, tiles_a, tiles_b](hipStream_t s) -> int {
        hipLaunchKernelGGL(layout_k, dim3((unsigned)(a.rows * tiles_a * tiles_b)), dim3(256), 0, s, a, tiles_a, tiles_b);
        return vh_check_launch("layout_k");
    });
}

extern "C" int vh_axpy(vh_ctx* ctx, const vh_axpy_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_axpy: null args");
    const vh_axpy_args a = *p;
    VH_REQUIRE(a.out && a.n > 0, "vh_axpy: null output");
    return vh_dispatch(ctx, VH_TAG_SAMPLER, 0.0, 4.0 * (double)a.n * (1 + (a.a ? 1 : 0) + (a.b ? 1 : 0)), [a](hipStream_t s) -> int {
        hipLaunchKernelGGL(axpy_k, dim3(blocks_for((long long)a.n, 256)), dim3(256), 0, s, a);
        return vh_check_launch("axpy_k");
    });
}

extern "C" int vh_resample(vh_ctx* ctx, const vh_resample_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_resample: null args");
    const vh_resample_args a = *p;
    VH_REQUIRE(a.in && a.out && vh_aligned16(a.in) && vh_aligned16(a.out), "vh_resample: null or misaligned tensor");
    VH_REQUIRE(a.rows > 0 && a.h > 0 && a.w > 0 && a.c > 0 && a.c % 4 == 0, "vh_resample: bad geometry (c must be a multiple of 4)");
    VH_REQUIRE(a.ntaps >= 2 && a.ntaps <= 8 && a.ntaps % 2 == 0, "vh_resample: the filter must have 2, 4, 6 or 8 taps (got %d)", a.ntaps);
    VH_REQUIRE(a.up || (a.h % 2 == 0 && a.w % 2 == 0), "vh_resample: down needs even input size");
    const long long opix = (long long)a.rows * (a.up ? 4LL * a.h * a.w : (long long)(a.h / 2) * (a.w / 2));
    const long long total = opix * (a.c / 4);
    const double bytes = 4.0 * a.c * ((double)a.rows * a.h * a.w + (double)opix);
    return vh_dispatch(ctx, VH_TAG_PIXNORM, 0.0, bytes, [a, total](hipStream_t s) -> int {
        hipLaunchKernelGGL(resample_k, dim3(blocks_for(total, 256)), dim3(256), 0, s, a, total);
        return vh_check_launch("resample_k");
    });
}

extern "C" int vh_nonzero_flag(vh_ctx* ctx, const vh_nonzero_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_nonzero_flag: null args");
    const vh_nonzero_args a = *p;
    VH_REQUIRE(a.in && a.flag, "vh_nonzero_flag: null tensor");
    VH_REQUIRE(a.rows > 0 && a.c_used > 0 && a.c_used <= a.c_total && a.hw > 0, "vh_nonzero_flag: bad geometry");
    const long long total = (long long)a.rows * a.c_used * a.hw;
    return vh_dispatch(ctx, VH_TAG_ASSEMBLE, 0.0, 4.0 * (double)total, [a, total](hipStream_t s) -> int {
        const hipError_t e = hipMemsetAsync(a.flag, 0, sizeof(float), s);
        if (e != hipSuccess) return vh_fail(VH_EHIP, "vh_nonzero_flag: memset: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(nonzero_flag_k, dim3(blocks_for(total, 256)), dim3(256), 0, s, a, total);
        return vh_check_launch("nonzero_flag_k");
    });
}

extern "C" int vh_moments(vh_ctx* ctx, const vh_moments_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_moments: null args");
    const vh_moments_args a = *p;
    VH_REQUIRE(a.a && a.b && a.outer, "vh_moments: null tensor");
    VH_REQUIRE(a.n > 0 && a.fa > 0 && a.fb > 0, "vh_moments: bad geometry");
    VH_REQUIRE((a.fa + 63) / 64 < 65536, "vh_moments: too many features");
    const dim3 grid((a.fb + 63) / 64, (a.fa + 63) / 64);
    const double flops = 2.0 * a.n * (double)a.fa * a.fb;
    const double bytes = 16.0 * (double)a.fa * a.fb + 4.0 * a.n * ((double)a.fa + a.fb);
    return vh_dispatch(ctx, VH_TAG_SAMPLER, flops, bytes, [a, grid](hipStream_t s) -> int {
        hipLaunchKernelGGL(moments_k, grid, dim3(256), 0, s, a);
        return vh_check_launch("moments_k");
    });
}

extern "C" int vh_psnr_sum(vh_ctx* ctx, const vh_psnr_args* p) {
    if (!p) return vh_fail(VH_EINVAL, "vh_psnr_sum: null args");
    const vh_psnr_args a = *p;
    VH_REQUIRE(a.x && a.y && a.acc && a.per_image, "vh_psnr_sum: null tensor");
    VH_REQUIRE(a.images > 0 && a.elems > 0 && (a.dtype == VH_U8 || a.dtype == VH_F32), "vh_psnr_sum: bad arguments");
    return vh_dispatch(ctx, VH_TAG_SAMPLER, 0.0, (a.dtype == VH_U8 ? 2.0 : 8.0) * a.images * (double)a.elems, [a](hipStream_t s) -> int {
        if (a.dtype == VH_U8)
            hipLaunchKernelGGL(psnr_k<unsigned char>, dim3(a.images), dim3(256), 0, s, static_cast<const unsigned char*>(a.x),
                               static_cast<const unsigned char*>(a.y), (long long)a.elems, a.per_image);
        else
            hipLaunchKernelGGL(psnr_k<float>, dim3(a.images), dim3(256), 0, s, static_cast<const float*>(a.x),
                               static_cast<const float*>(a.y), (long long)a.elems, a.per_image);
        hipLaunchKernelGGL(psnr_fold_k, dim3(1), dim3(64), 0, s, a.per_image, a.images, a.acc);
        return vh_check_launch("psnr_k");
    });
}
