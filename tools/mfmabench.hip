// Micro-benchmark (GPU box): issue rate of v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64 in the instruction
// mixes the covariance kernel uses.   hipcc --offload-arch=gfx950 -O3 tools/mfmabench.hip -o tools/mfmabench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f64x4 = __attribute__((ext_vector_type(4))) double;

// MODE 0: 8 independent accumulators, constant operands, back to back
// MODE 1: a VALU multiply produces the A operand of every MFMA (one shared temp register)
// MODE 2: as 1 plus 8 float64 VALU operations per MFMA (the fold of the fp32 chains)
template <typename REAL, int MODE>
__global__ __launch_bounds__(256) void bench(float* out, int iters, float seed) {
    using acc_t = std::conditional_t<sizeof(REAL) == 4, f32x4, f64x4>;
    acc_t acc[8];
    double tot[8][4];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 4; ++r) {
            acc[i][r] = 0;
            tot[i][r] = 0;
        }
    REAL x[4], w0 = (REAL)seed, w1 = (REAL)(seed * 0.5f);
    for (int j = 0; j < 4; ++j) x[j] = (REAL)(threadIdx.x * 0.001f + j);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                REAL a = x[j];
                if constexpr (MODE >= 1) a = x[j] * (k ? w1 : w0);
                if constexpr (MODE == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) tot[2 * j + k][r] += (double)acc[2 * j + k][r];
                }
                if constexpr (sizeof(REAL) == 4)
                    acc[2 * j + k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x[j], acc[2 * j + k], 0, 0, 0);
                else
                    acc[2 * j + k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x[j], acc[2 * j + k], 0, 0, 0);
            }
        }
        // keep the operands changing so that nothing is hoisted
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = x[j] * (REAL)1.0001f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 4; ++r) s += (float)acc[i][r] + (float)tot[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

template <typename REAL, int MODE>
void run(const char* name, int blocks_per_cu) {
    float* out;
    hipMalloc(&out, 256 * 4 * 256 * 8);
    const int iters = 2000;
    const int grid = 256 * blocks_per_cu;
    bench<REAL, MODE><<<grid, 256>>>(out, 10, 1.f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    bench<REAL, MODE><<<grid, 256>>>(out, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    float cyc;
    hipMemcpy(&cyc, out, 4, hipMemcpyDeviceToHost);
    const double mfma_per_simd = (double)iters * 8 * blocks_per_cu;   // one wave of every block on each SIMD
    printf("%-28s waves/SIMD %d: %8.1f us, wave-cycles per MFMA %.1f, SIMD cycles per MFMA %.1f (%.0f MHz)\n", name,
           blocks_per_cu, ms * 1e3, cyc / (iters * 8.0), cyc / mfma_per_simd, cyc / (ms * 1e3));
    hipFree(out);
}

int main() {
    for (int b : {1, 2, 3, 4}) {
        run<float, 0>("f32 back-to-back", b);
        run<float, 1>("f32 + v_mul per MFMA", b);
        run<float, 2>("f32 + v_mul + 8 f64 VALU", b);
        run<double, 0>("f64 back-to-back", b);
        run<double, 1>("f64 + v_mul per MFMA", b);
    }
    return 0;
}
