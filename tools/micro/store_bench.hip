// Micro-benchmark: how fast can one CU stream stores?  (tools/micro; not part of the library)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_bench.hip -o /tmp/store_bench && /tmp/store_bench
// Each 512-thread workgroup writes a 512 x 128 fp32 tile (256 KB) of a [M][128] matrix the way conv_x3_glds' epilogue does
// (a wave instruction = 8 rows x 128 B, row pitch 512 B) or fully contiguous (1 KB per wave instruction); grid = CUs x rounds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(512) void store_k(float* out, int rows_per_wg) {
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    float* base = out + (size_t)blockIdx.x * rows_per_wg * 128;
    const float4 v = make_float4((float)t, 1.f, 2.f, 3.f);
    if (MODE == 0) {            // epilogue pattern: wave w owns rows [w*64, w*64+64) x 64 columns? no: 128 rows x 64 cols per wave (4x2 waves)
        const int wm = w >> 1, wn = w & 1;
        const int cg = l & 7, rsub = l >> 3;
        for (int blk = 0; blk < 8; ++blk) {                 // 8 blocks of 32 x 32
            const int mi = blk >> 1, ni = blk & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 128 + mi * 32 + rsub + 8 * i, col = wn * 64 + ni * 32 + 4 * cg;
                *reinterpret_cast<float4*>(base + (size_t)row * 128 + col) = v;
            }
        }
    } else {                    // contiguous: each wave instruction writes 1 KB
        for (int i = 0; i < 32; ++i)
            *reinterpret_cast<float4*>(base + ((size_t)(i * 8 + w) * 64 + l) * 4) = v;
    }
}

int main() {
    const int rows_per_wg = 512;
    for (int cus : {128, 256}) {
        for (int rounds : {1, 8}) {
            const int grid = cus * rounds;
            float* out;
            hipMalloc(&out, (size_t)grid * rows_per_wg * 128 * 4);
            for (int mode = 0; mode < 2; ++mode) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                for (int it = 0; it < 3; ++it) {
                    if (mode == 0) hipLaunchKernelGGL(store_k<0>, dim3(grid), dim3(512), 0, 0, out, rows_per_wg);
                    else hipLaunchKernelGGL(store_k<1>, dim3(grid), dim3(512), 0, 0, out, rows_per_wg);
                }
                hipEventRecord(e0);
                const int n = 20;
                for (int it = 0; it < n; ++it) {
                    if (mode == 0) hipLaunchKernelGGL(store_k<0>, dim3(grid), dim3(512), 0, 0, out, rows_per_wg);
                    else hipLaunchKernelGGL(store_k<1>, dim3(grid), dim3(512), 0, 0, out, rows_per_wg);
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                ms /= n;
                const double bytes = (double)grid * rows_per_wg * 128 * 4;
                printf("workgroups %5d (%d per CU on %d CUs) mode %s: %.3f ms  %.2f TB/s  %.1f GB/s per CU  %.2f us per 256 KB tile\n", grid, rounds, cus,
                       mode ? "contiguous" : "epilogue  ", ms, bytes / ms / 1e9, bytes / ms / 1e6 / cus, ms * 1e3 / rounds);
            }
            hipFree(out);
        }
    }
    return 0;
}
