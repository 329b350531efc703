// Bare MFMA loops on random operands held in registers: sustained rate and in-kernel clock of v_mfma_f32_16x16x32_bf16 against
// v_mfma_i32_16x16x64_i8 (MI355X_MICROARCH.md, DVFS give-back items 6 and 7).  One workgroup of 512 threads per CU slot, two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o tools/micro/mfma_rate && tools/micro/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// KIND 0: bf16, 1: int8, 2: bf16 with the A operands re-read from LDS (one ds_read_b128 per 4 MFMAs, the convolution's ratio),
// 3: bf16 with both operands re-read (one per 2 MFMAs)
template <int KIND>
__global__ __launch_bounds__(512, 2) void loop(const uint4* in, float* out, unsigned long long* stamps, int iters) {
    const int t = threadIdx.x + blockIdx.x * 512;
    __shared__ uint4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = in[(i * 7 + blockIdx.x) & 0xffff];
    __syncthreads();
    uint4 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 0xffff]; b[i] = in[(t * 8 + 4 + i) & 0xffff]; }
    f32x4 cf[8]; i32x4 ci[8];
    for (int i = 0; i < 8; ++i) { cf[i] = {0.f, 0.f, 0.f, 0.f}; ci[i] = {0, 0, 0, 0}; }
    const unsigned long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND >= 2) {
            const int base = (threadIdx.x * 5 + it * 64) & 4095;
            a[0] = lds[base]; a[1] = lds[(base + 1024) & 4095];
            if constexpr (KIND == 3) { b[0] = lds[(base + 2048) & 4095]; b[1] = lds[(base + 3072) & 4095]; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (KIND == 0 || KIND >= 2)
                cf[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i & 3]), __builtin_bit_cast(bf16x8, b[(i >> 1) & 3]), cf[i], 0, 0, 0);
            else
                ci[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a[i & 3]), __builtin_bit_cast(i32x4, b[(i >> 1) & 3]), ci[i], 0, 0, 0);
        }
    }
    const unsigned long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += cf[i][r] + (float)ci[i][r];
    out[t] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = m1 - m0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
    const int nwg = 256, iters = 20000;
    uint4* in; float* out; unsigned long long* st;
    hipMalloc(&in, 65536 * 16); hipMalloc(&out, nwg * 512 * 4); hipMalloc(&st, nwg * 16);
    std::vector<unsigned> h(65536 * 4);
    srand(1);
    for (auto& v : h) {                      // finite bf16 pairs / arbitrary int8: random mantissas, exponents near 1
        unsigned lo = (rand() & 0x807f) | 0x3f00, hi = (rand() & 0x807f) | 0x3f00;
        v = lo | (hi << 16);
    }
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int kind = 0; kind < 4; ++kind) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e30f, ms = 0.f;
        for (int rep = 0; rep < 60; ++rep) {         // ~2 s of back-to-back launches so that the clock settles
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(loop<0>, dim3(nwg), dim3(512), 0, 0, in, out, st, iters);
            else if (kind == 1) hipLaunchKernelGGL(loop<1>, dim3(nwg), dim3(512), 0, 0, in, out, st, iters);
            else if (kind == 2) hipLaunchKernelGGL(loop<2>, dim3(nwg), dim3(512), 0, 0, in, out, st, iters);
            else hipLaunchKernelGGL(loop<3>, dim3(nwg), dim3(512), 0, 0, in, out, st, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 40) best = std::min(best, ms);
        }
        std::vector<unsigned long long> hs(nwg * 2);
        hipMemcpy(hs.data(), st, nwg * 16, hipMemcpyDeviceToHost);
        std::vector<double> clk;
        for (int i = 0; i < nwg; ++i) clk.push_back((double)hs[2 * i] / (double)hs[2 * i + 1] * 0.1);
        std::sort(clk.begin(), clk.end());
        const double ops = 2.0 * 16 * 16 * (kind == 1 ? 64 : 32) * 8.0 * iters * 8 /*waves*/ * nwg;
        static const char* names[4] = {"v_mfma_f32_16x16x32_bf16, operands in registers", "v_mfma_i32_16x16x64_i8, operands in registers",
                                       "bf16 + 2 ds_read_b128 per 8 MFMAs", "bf16 + 4 ds_read_b128 per 8 MFMAs"};
        printf("%s: %.3f ms, %.1f T(FL)OP/s, in-kernel clock median %.3f GHz, loop cycles per MFMA and SIMD %.2f\n",
               names[kind], best, ops / best / 1e9, clk[nwg / 2], (double)hs[0] / (8.0 * iters * 2));
    }
    return 0;
}
