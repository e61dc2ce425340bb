// Micro-benchmark (GPU box): issue rate of the float64 instruction forms of csrc/kernels_update16r.hip, ONE wave per SIMD
// (64-thread workgroups, one per SIMD) and two.
//   hipcc --offload-arch=gfx950 -O3 tools/r6/dppbench.hip -o /tmp/dppbench && /tmp/dppbench
#include <hip/hip_runtime.h>

#include <cstdio>

// MODE 0: v_fmac_f64 (VOP2), 16 independent accumulators
// MODE 1: v_fmac_f64_dpp row_newbcast, 16 independent accumulators
// MODE 2: v_mov_b64_dpp row_newbcast + v_fmac_f64
// MODE 3: v_fmac_f64_dpp where the DPP source IS the accumulator (the elimination step's form)
// MODE 4: v_fma_f64 chain of 4 dependent (latency)
template <int MODE>
__global__ __launch_bounds__(64) void bench(double* out, unsigned long long* clk, int iters, double seed) {
    double acc[16], x[8];
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    for (int j = 0; j < 8; ++j) x[j] = seed * (j + 1) * 1e-9;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (MODE == 0) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[i]) : "v"(x[i & 7]), "v"(x[(i + r + 1) & 7]));
                if constexpr (MODE == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(x[i & 7]), "v"(x[(i + r + 1) & 7]));
                if constexpr (MODE == 2) {
                    double t;
                    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(x[i & 7]));
                    asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[i]) : "v"(t), "v"(x[(i + r + 1) & 7]));
                }
                if constexpr (MODE == 3) asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(x[(i + r + 1) & 7]));
                if constexpr (MODE == 4) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[0]) : "v"(x[i & 7]), "v"(x[(i + r + 1) & 7]));
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks) {
    double* out;
    unsigned long long* clk;
    hipMalloc(&out, blocks * 64 * 8);
    hipMalloc(&clk, blocks * 8);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 1.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 64 * (MODE == 2 ? 2 : 1);
    printf("%-44s blocks %5d: %7.2f ns per instruction and wave (%.3f ms)\n", name, blocks, ms * 1e6 / n, ms);
    hipFree(out);
    hipFree(clk);
}

int main() {
    for (int blocks : {1024, 2048}) {
        run<0>("v_fmac_f64", blocks);
        run<1>("v_fmac_f64_dpp row_newbcast", blocks);
        run<2>("v_mov_b64_dpp + v_fmac_f64 (per instr)", blocks);
        run<3>("v_fmac_f64_dpp acc = dpp source", blocks);
        run<4>("v_fmac_f64 dependent chain", blocks);
    }
    return 0;
}
