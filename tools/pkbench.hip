// Micro-benchmark (GPU box): issue rate of the packed-fp32 instruction forms the covariance kernels are made of, with 1
// and 2 waves per SIMD, and the clock the chip sustains under them.
//   hipcc --offload-arch=gfx950 -O3 tools/pkbench.hip -o tools/pkbench && tools/pkbench
#include <hip/hip_runtime.h>

#include <cstdio>

using v2f = __attribute__((ext_vector_type(2))) float;

// MODE 0: v_fma_f32, 16 independent accumulators
// MODE 1: v_pk_fma_f32 plain, 16 independent accumulator pairs, three distinct VGPR-pair sources
// MODE 2: v_pk_fma_f32 with the op_sel broadcast forms of cov_arith.h (w0 / w1 / hi_swap) + v_pk_mul neg_hi
// MODE 3: as 2, 64 accumulator pairs (the register footprint of the real kernels)
// MODE 4: v_fma_f64, 16 independent accumulators;  MODE 5: v_cvt_f64_f32 + v_mul_f64 + v_fma_f64 in the mix of the float64
//         Hermitian-half kernel (1 : 3 : 9)
template <int MODE>
__global__ __launch_bounds__(256) void bench(float* out, unsigned long long* clk, int iters, float seed) {
    constexpr int NA = MODE == 3 ? 64 : 16;
    v2f acc[NA];
    for (int i = 0; i < NA; ++i) acc[i] = v2f{0.f, 0.f};
    v2f x[8], w = v2f{seed, seed * 0.5f};
    for (int j = 0; j < 8; ++j) x[j] = v2f{threadIdx.x * 0.001f + j, threadIdx.x * 0.002f - j};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < NA; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(x[i & 7].x), "v"(x[(i + r) & 7].y));
        } else if constexpr (MODE == 4) {
            double* d = reinterpret_cast<double*>(acc);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < NA; ++i)
                    asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(*reinterpret_cast<double*>(&x[i & 7])), "v"(*reinterpret_cast<double*>(&x[(i + r + 1) & 7])));
        } else if constexpr (MODE == 5) {
            double* d = reinterpret_cast<double*>(acc);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                double c0, c1, p0, p1, p2;
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(c0) : "v"(x[g].x));
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(c1) : "v"(x[g + 4].y));
                asm volatile("v_mul_f64 %0, %1, %2" : "=v"(p0) : "v"(c0), "v"(c1));
                asm volatile("v_mul_f64 %0, %1, %1" : "=v"(p1) : "v"(c0));
                asm volatile("v_mul_f64 %0, %1, %1" : "=v"(p2) : "v"(c1));
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[(g * 3 + e) & 15]) : "v"(p0), "v"(c0));
                    asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[(g * 3 + e + 5) & 15]) : "v"(p1), "v"(c1));
                    asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[(g * 3 + e + 10) & 15]) : "v"(p2), "v"(c0));
                }
            }
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < NA; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x[i & 7]), "v"(x[(i + r + 1) & 7]));
        } else {
#pragma unroll
            for (int g = 0; g < NA / 8; ++g) {
                v2f p[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(p[e]) : "v"(x[e]), "v"(x[(e + g + 1) & 7]));
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(p[e]) : "v"(x[e]), "v"(x[(e + g + 1) & 7]));
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[g * 8 + e]) : "v"(w), "v"(p[e]));
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[(g * 8 + e + NA / 2) % NA]) : "v"(w), "v"(p[e]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < NA; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = w1 - w0;
    }
}

template <int MODE>
void run(const char* name, int blocks_per_cu, int instr_per_iter) {
    float* out;
    unsigned long long* clk;
    hipMalloc(&out, 256 * 4 * 256 * 8);
    hipMalloc(&clk, 16);
    const int iters = 20000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    bench<MODE><<<256 * blocks_per_cu, 256>>>(out, clk, 100, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    bench<MODE><<<256 * blocks_per_cu, 256>>>(out, clk, iters, 1.f);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double n = (double)iters * instr_per_iter * blocks_per_cu;      // wave-instructions per SIMD
    printf("%-44s %d wave(s)/SIMD: %.3f ms, %.2f ns per instruction and SIMD; block 0: %.2f s_memtime ticks per instruction, wall %.3f ms (100 MHz counter)\n",
           name, blocks_per_cu, ms, ms * 1e6 / n, (double)h[0] / ((double)iters * instr_per_iter), h[1] / 1e5);
    hipFree(out);
    hipFree(clk);
}

int main() {
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run<0>("v_fma_f32 x64", bpc, 64);
        run<1>("v_pk_fma_f32 plain x64", bpc, 64);
        run<2>("pk_mul + 3 pk_fma (op_sel forms) x64", bpc, 64);
        run<3>("same, 64 accumulator pairs x256", bpc, 256);
        run<4>("v_fma_f64 x64", bpc, 64);
        run<5>("cvt_f64_f32 : mul_f64 : fma_f64 = 8 : 12 : 36", bpc, 56);
    }
    return 0;
}
