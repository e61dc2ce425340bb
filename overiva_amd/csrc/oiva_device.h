// Device-side helpers shared by several kernels.
#pragma once
#include "oiva_internal.h"

#include <type_traits>
#include <utility>

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

// ---- register-level lane exchanges (no LDS crossbar): DPP within 16-lane rows, v_permlane{16,32}_swap
//      across rows.  Values wider than 32 bits go through them one dword at a time. ----
template <int CTRL>
__device__ __forceinline__ int dpp32(int x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp(float v) { return __int_as_float(dpp32<CTRL>(__float_as_int(v))); }
template <int CTRL>
__device__ __forceinline__ unsigned dpp(unsigned v) { return (unsigned)dpp32<CTRL>((int)v); }
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
    return __hiloint2double(dpp32<CTRL>(__double2hiint(v)), dpp32<CTRL>(__double2loint(v)));
}
// same, writing only the banks (groups of 4 lanes of a 16-lane row) selected by BANKS; the other lanes keep `old`
template <int CTRL, int BANKS>
__device__ __forceinline__ int dpp32_banks(int old, int x) {
    return __builtin_amdgcn_update_dpp(old, x, CTRL, 0xF, BANKS, false);
}
template <int CTRL, int BANKS>
__device__ __forceinline__ float dpp_banks(float old, float v) {
    return __int_as_float(dpp32_banks<CTRL, BANKS>(__float_as_int(old), __float_as_int(v)));
}
template <int CTRL, int BANKS>
__device__ __forceinline__ double dpp_banks(double old, double v) {
    return __hiloint2double(dpp32_banks<CTRL, BANKS>(__double2hiint(old), __double2hiint(v)),
                            dpp32_banks<CTRL, BANKS>(__double2loint(old), __double2loint(v)));
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): loop indices that must be compile-time constants
// (DPP controls, lane numbers of v_readlane)
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}
constexpr int kDppXor1 = 0xB1;   // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;   // quad_perm [2,3,0,1]
constexpr int kDppRor4 = 0x124;  // row_ror:4
constexpr int kDppRor8 = 0x128;  // row_ror:8
constexpr int kDppHalfMirror = 0x141;  // row_half_mirror: lane l <-> 7 - l inside each group of 8
// a + b where b is the value of the lane 16 (32) lanes away: the swap of a register with itself
// leaves {own row pair member, other row pair member} in the two results
__device__ __forceinline__ float swapsum16(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swapsum32(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ double swapsum16(double v) {
    auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ double swapsum32(double v) {
    auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ unsigned swapmax16(unsigned v) {
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return r[0] > r[1] ? r[0] : r[1];
}
__device__ __forceinline__ unsigned swapmax32(unsigned v) {
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return r[0] > r[1] ? r[0] : r[1];
}


}  // namespace oiva
