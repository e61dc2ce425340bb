// Device-side helpers shared by several kernels.
#pragma once
#include "oiva_internal.h"

namespace oiva {

constexpr float kEpsR = 1e-15f;  // reference overiva.py:170

// block-wide sum of one double per thread, fixed order (wave shuffle tree, then the waves in order);
// every thread of the block must call it; scratch holds kWaves doubles
__device__ __forceinline__ double block_sum(double v, double* scratch) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    double s = 0.;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) s += scratch[w];
    return s;
}

// gamma_k = mean_t R[t,k]  (reference overiva.py:158) from the per-block partial sums the activation kernel left
// behind R (rsum_offset_floats): a fixed-order sum of T/256 float64 values at wave-uniform addresses (scalar
// loads), so every lane of every kernel obtains the same bits for the price of a few instructions.
__device__ __forceinline__ double gamma_of(const float* __restrict__ R, int T, int K, int k) {
    const double* S = reinterpret_cast<const double*>(R + rsum_offset_floats(T, K));
    const int nblk = rsum_blocks(T);
    double s = 0.;
    for (int b = 0; b < nblk; ++b) s += S[(size_t)b * K + k];
    return s / (double)T;
}

// 1 / max(r / gamma, eps)   (reference overiva.py:159, :170-173) with ginv = 1 / gamma
__device__ __forceinline__ float activation_weight(float r, float ginv) {
    float rn = r * ginv;
    rn = rn < kEpsR ? kEpsR : rn;   // a NaN stays NaN, like r[r < eps] = eps in the reference
    return 1.f / rn;
}

// one M-vector (M complex64 = M*8 bytes, 16-byte aligned for even M) into split re/im registers
template <int M>
__device__ __forceinline__ void load_x(const float2* __restrict__ p, float (&xr)[M], float (&xi)[M]) {
    if constexpr (M % 2 == 0) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
#pragma unroll
        for (int i = 0; i < M / 2; ++i) {
            const float4 v = p4[i];
            xr[2 * i] = v.x;
            xi[2 * i] = v.y;
            xr[2 * i + 1] = v.z;
            xi[2 * i + 1] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const float2 v = p[i];
            xr[i] = v.x;
            xi[i] = v.y;
        }
    }
}

// ---- fp32 / fp64 matrix-core tile of shape 16x16x4 (A, B: one value per lane; C/D: 4 values per lane) ----
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f64x4 = __attribute__((ext_vector_type(4))) double;

template <typename REAL>
struct Mfma;
template <>
struct Mfma<float> {
    using acc_t = f32x4;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);     // exact fp32 fmaf chain over the 4 k
    }
    // C/D layout: lane l, register r -> row; the column is l & 15
    static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};
template <>
struct Mfma<double> {
    using acc_t = f64x4;
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }   // NOT the fp32 map
};

}  // namespace oiva
