// Weighted spatial covariance on the matrix cores, for 9..16 channels             reference overiva.py:179 / :87
//
// With M = 16 and K = 16 the pass is 268 GFLOP per iteration against 1 GB of X: bound by the fp32 MFMA
// rate (157 TF -> >= 1.7 ms), not by HBM.  Real Gram form: x~ = (re_0, im_0, re_1, im_1, ...) is the M-vector
// as stored (2M <= 32 floats), G_k = sum_t w_k[t] x~ x~^T is one 32x32 tile of v_mfma_f32_32x32x2_f32 per
// (bin, source) with the frame axis as the contraction:
//     A[i][kk] = w_k[t+kk] * x~_i[t+kk],  B[kk][j] = x~_j[t+kk],  kk = 0,1  (lane l holds i = j = l&31, kk = l>>5:
//     the SAME register feeds A (scaled) and B).
// V_re[c][d] = G[2c][2d] + G[2c+1][2d+1],  V_im[c][d] = G[2c+1][2d] - G[2c][2d+1].
// fp32 MFMA is an exact fp32 FMA chain, so numerics match the VALU kernel.
//
// A workgroup = one bin x one frame split; wave w owns sources [KW*w, KW*w + KW) for the whole split, so
// there is no cross-wave reduction and X is read from HBM once (the waves of a workgroup request the same
// lines).  Output: the packed Hermitian partial layout of the VALU kernel.
#include "oiva_device.h"

namespace oiva {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// gamma_k = mean_t R[t,k], one wave, fixed order
__device__ __forceinline__ float wave_gamma(const float* __restrict__ R, int T, int K, int k) {
    const int lane = threadIdx.x & 63;
    double s = 0.;
    for (int t0 = lane; t0 < T; t0 += 64 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = t0 + 64 * u;
            v[u] = R[(size_t)(t < T ? t : T - 1) * K + k];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (t0 + 64 * u < T) ? (double)v[u] : 0.;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return (float)(s / (double)T);
}

constexpr int kPairUnroll = 4;   // frame pairs whose loads are issued together

template <int KW, bool UNIT>
__global__ __launch_bounds__(256) void cov_mfma_kernel(const float* __restrict__ Xf, const float* __restrict__ R,
                                                       float* __restrict__ wscale, int model, int raw,
                                                       float* __restrict__ Vpart, int T, int F, int M, int K, int tc) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i32 = lane & 31;          // row of A / column of B: index into x~
    const int half = lane >> 5;         // which frame of the pair this lane feeds
    const int f = blockIdx.x;
    const int k0 = wave * KW;
    const int t_begin = blockIdx.y * tc;
    const int t_end = min(T, t_begin + tc);
    const int npairs = (t_end - t_begin + 1) >> 1;
    const int M2 = 2 * M;
    const bool ivalid = i32 < M2;
    const int NA = M * M;

    float ginv[KW];
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
        ginv[kk] = 1.f;
        if constexpr (!UNIT) {
            const int k = k0 + kk;
            const float gamma = wave_gamma(R, T, K, k < K ? k : K - 1);
            if (!(raw & 1)) ginv[kk] = 1.f / gamma;
            if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && k < K && wscale != nullptr)
                wscale[k] = model == OIVA_MODEL_LAPLACE ? gamma : sqrtf(gamma);   // overiva.py:163 / :167
        }
    }

    f32x16 acc[KW];
#pragma unroll
    for (int kk = 0; kk < KW; ++kk)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kk][r] = 0.f;

    const size_t frame_stride = (size_t)F * M2;                       // floats per frame
    const float* px = Xf + (size_t)f * M2 + (ivalid ? i32 : 0);
    for (int p0 = 0; p0 < npairs; p0 += kPairUnroll) {
        float x[kPairUnroll], rv[kPairUnroll][KW];
#pragma unroll
        for (int u = 0; u < kPairUnroll; ++u) {
            const int t = t_begin + 2 * (p0 + u) + half;
            const int tcl = t < t_end ? t : T - 1;
            x[u] = px[(size_t)tcl * frame_stride];
            if constexpr (!UNIT) {
#pragma unroll
                for (int kk = 0; kk < KW; ++kk) {
                    const int k = k0 + kk;
                    rv[u][kk] = R[(size_t)tcl * K + (k < K ? k : K - 1)];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kPairUnroll; ++u) {
            const int t = t_begin + 2 * (p0 + u) + half;
            const bool live = (p0 + u < npairs) && (t < t_end) && ivalid;
            const float xv = live ? x[u] : 0.f;
#pragma unroll
            for (int kk = 0; kk < KW; ++kk) {
                float w = 1.f;
                if constexpr (!UNIT) w = (k0 + kk < K) ? activation_weight(rv[u][kk], ginv[kk]) : 0.f;
                acc[kk] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv * w, xv, acc[kk], 0, 0, 0);
            }
        }
    }

    // G -> packed Hermitian V.  C/D layout of 32x32: lane l, register r holds
    //   G[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31].
    // A row pair (2c, 2c+1) sits in registers (r, r+1) of one lane; a column pair (2d, 2d+1) in lanes (l, l+1).
    const int d = i32 >> 1;
    const bool even = (i32 & 1) == 0;
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
        const int k = k0 + kk;
        float* out = Vpart + (((size_t)blockIdx.y * F + f) * K + (k < K ? k : 0)) * NA;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const int c = (((r & 3) + 8 * (r >> 2) + 4 * half)) >> 1;    // r even -> row 2c
            const float g00 = acc[kk][r];          // G[2c][col]
            const float g10 = acc[kk][r + 1];      // G[2c+1][col]
            const float g01 = __shfl_down(g00, 1, 64);   // G[2c][col+1]
            const float g11 = __shfl_down(g10, 1, 64);   // G[2c+1][col+1]
            if (even && k < K && c < M && d < M && c <= d) {
                const float vre = g00 + g11;
                if (c == d) {
                    out[c] = vre;
                } else {
                    const int o = herm_pair_index(M, c, d);
                    out[o] = vre;
                    out[o + 1] = g10 - g01;
                }
            }
        }
    }
}

}  // namespace

int cov_mfma_sources_per_wave(int K) { return K == 1 ? 1 : (K == 2 ? 2 : 4); }

hipError_t launch_cov_mfma(hipStream_t s, const float2* X, const float* R, float* wscale, int model, int raw,
                           float* Vpart, int T, int F, int M, int K, int nsplit, int tc) {
    const bool unit = R == nullptr;
    const int kw = unit ? 1 : cov_mfma_sources_per_wave(K);
    const int waves = (K + kw - 1) / kw;
    if (waves > 4 || M > 16) return hipErrorInvalidValue;
    dim3 grid(F, nsplit);
    dim3 block(64 * waves);
    const float* Xf = reinterpret_cast<const float*>(X);
    if (unit)
        cov_mfma_kernel<1, true><<<grid, block, 0, s>>>(Xf, R, wscale, model, raw, Vpart, T, F, M, K, tc);
    else if (kw == 1)
        cov_mfma_kernel<1, false><<<grid, block, 0, s>>>(Xf, R, wscale, model, raw, Vpart, T, F, M, K, tc);
    else if (kw == 2)
        cov_mfma_kernel<2, false><<<grid, block, 0, s>>>(Xf, R, wscale, model, raw, Vpart, T, F, M, K, tc);
    else
        cov_mfma_kernel<4, false><<<grid, block, 0, s>>>(Xf, R, wscale, model, raw, Vpart, T, F, M, K, tc);
    return hipGetLastError();
}

}  // namespace oiva
