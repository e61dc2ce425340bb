// Per-lane arithmetic of the demixing passes (reference overiva.py:140, :153/:155), shared by the streaming kernels
// (kernels_demix.hip) and the X-resident iteration kernel (kernels_resident.hip).
#pragma once
#include "oiva_device.h"

namespace oiva {

// conj(W[f][m][k0+kk]) for the lane's bin; W_hat is (F, M, M) row-major, column k = demixing vector k
template <int M, int KP>
__device__ __forceinline__ void load_wconj(const float2* __restrict__ What, int f, int k0, int K, float (&wr)[KP][M],
                                           float (&wi)[KP][M]) {
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float2 v = make_float2(0.f, 0.f);
            if (k0 + kk < K) v = What[((size_t)f * M + m) * M + k0 + kk];
            wr[kk][m] = v.x;
            wi[kk][m] = -v.y;
        }
    }
}

// y = sum_m conj(w_m) x_m
template <int M>
__device__ __forceinline__ void demix_one(const float (&wr)[M], const float (&wi)[M], const float (&xr)[M],
                                          const float (&xi)[M], float& yr, float& yi) {
    float ar = 0.f, ai = 0.f;
#pragma unroll
    for (int m = 0; m < M; ++m) {
        ar = fmaf(wr[m], xr[m], ar);
        ar = fmaf(-wi[m], xi[m], ar);
        ai = fmaf(wr[m], xi[m], ai);
        ai = fmaf(wi[m], xr[m], ai);
    }
    yr = ar;
    yi = ai;
}

// sum over the 16 lanes of a DPP row (= the 16 bins of one frame phase); every lane gets the total
__device__ __forceinline__ float row16_sum(float v) {
    int x;
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false);
    v += __int_as_float(x);
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, false);
    v += __int_as_float(x);
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141 /* row_half_mirror */, 0xF, 0xF, false);
    v += __int_as_float(x);
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140 /* row_mirror */, 0xF, 0xF, false);
    v += __int_as_float(x);
    return v;
}

}  // namespace oiva
