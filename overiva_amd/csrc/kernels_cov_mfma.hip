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
#include <cstdlib>

#include "oiva_device.h"

namespace oiva {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kPairUnroll = 4;   // frame pairs whose loads are issued together

// Final weights w[t,k] = 1 / max(r[t,k] / gamma_k, eps) (overiva.py:158-173) for the matrix-core kernel, which
// has no VALU slots to spare for the divide (4 MFMAs of 64 cycles per frame pair and wave).  One workgroup
// per 256 frames; every workgroup derives gamma itself (block_gamma, fixed order).
__global__ __launch_bounds__(kBlock) void weights_kernel(const float* __restrict__ R, float* __restrict__ Wt,
                                                         float* __restrict__ wscale, int model, int raw, int T, int K,
                                                         int Kp) {
    __shared__ double scratch[kWaves];
    const int t = blockIdx.x * kBlock + threadIdx.x;
    for (int k = K; k < Kp; ++k)
        if (t < T) Wt[(size_t)t * Kp + k] = 0.f;        // padding columns: sources that do not exist weigh 0
    for (int k = 0; k < K; ++k) {
        const float gamma = block_gamma(R, T, K, k, scratch);
        const float ginv = (raw & 1) ? 1.f : 1.f / gamma;
        if (t < T) Wt[(size_t)t * Kp + k] = activation_weight(R[(size_t)t * K + k], ginv);
        if (blockIdx.x == 0 && threadIdx.x == 0 && wscale != nullptr)
            wscale[k] = model == OIVA_MODEL_LAPLACE ? gamma : sqrtf(gamma);   // overiva.py:163 / :167
    }
}

template <int KW, bool UNIT>
__global__ __launch_bounds__(256) void cov_mfma_kernel(const float* __restrict__ Xf, const float* __restrict__ Wt,
                                                       float* __restrict__ Vpart, int T, int F, int M, int K, int Kp,
                                                       int tc) {
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

    f32x16 acc[KW];
#pragma unroll
    for (int kk = 0; kk < KW; ++kk)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kk][r] = 0.f;

    // The matrix pipe needs 4 MFMAs x 64 cycles per frame pair and wave, and four waves share a SIMD, so
    // the VALU budget is ~60 issue cycles per pair and wave: x is one vector load through a running
    // pointer (no per-load 64-bit multiplies), the weights of the pair's two frames are wave-uniform and
    // come through the scalar cache (one select per weight), masks are multiplications.
    const size_t frame_stride = (size_t)F * M2;                       // floats per frame
    const size_t pair_stride = 2 * frame_stride;
    const float xmask = ivalid ? 1.f : 0.f;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int kcl[KW];
    float kmask[KW];
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
        const int k = wave_u * KW + kk;
        kcl[kk] = k < K ? k : K - 1;
        kmask[kk] = k < K ? 1.f : 0.f;
    }
    const float* px = Xf + (size_t)f * M2 + (ivalid ? i32 : 0) + (size_t)(t_begin + half) * frame_stride;
    const int nfull = (t_end - t_begin) >> 1;                          // pairs with both frames inside the split
    int p0 = 0;
    for (; p0 + kPairUnroll <= nfull; p0 += kPairUnroll) {
        float x[kPairUnroll];
#pragma unroll
        for (int u = 0; u < kPairUnroll; ++u) {
            x[u] = *px;
            px += pair_stride;
        }
#pragma unroll
        for (int u = 0; u < kPairUnroll; ++u) {
            const float xv = x[u] * xmask;
            const float* w0 = Wt + (size_t)(t_begin + 2 * (p0 + u)) * Kp;    // uniform: scalar loads
#pragma unroll
            for (int kk = 0; kk < KW; ++kk) {
                float a = xv;
                if constexpr (!UNIT) {
                    const float wa = w0[kcl[kk]], wb = w0[Kp + kcl[kk]];
                    a = xv * ((half ? wb : wa) * kmask[kk]);
                }
                acc[kk] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xv, acc[kk], 0, 0, 0);
            }
        }
    }
    // tail: remaining pairs one at a time; the last one may have only its first frame inside the split
    for (; p0 < npairs; ++p0) {
        const int t = t_begin + 2 * p0 + half;
        const float live = t < t_end ? xmask : 0.f;
        const int tcl = t < t_end ? t : T - 1;
        const float xv = Xf[(size_t)tcl * frame_stride + (size_t)f * M2 + (ivalid ? i32 : 0)] * live;
#pragma unroll
        for (int kk = 0; kk < KW; ++kk) {
            float a = xv;
            if constexpr (!UNIT) a = xv * (Wt[(size_t)tcl * Kp + kcl[kk]] * kmask[kk]);
            acc[kk] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xv, acc[kk], 0, 0, 0);
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


// ---------------------------------------------------------------------------------------------
// Planar 16x16x4 form (default).  With A = w * Re x, w * Im x and B = Re x, Im x as separate operands
// (channel = row/column, 4 frames = contraction):
//     V_re      += (w xr) xr^T + (w xi) xi^T        two MFMAs into ONE 16x16 accumulator tile
//     G_ir      += (w xi) xr^T                      one MFMA;  V_im[c][d] = G_ir[c][d] - G_ir[d][c]
// i.e. 3 x v_mfma_f32_16x16x4_f32 (32 cycles each) per 4 frames and source = 96 cycles, against 128 for the
// 32x32x2 real-Gram tile, and 8 accumulator registers per source instead of 16, so ONE wave carries all 16
// sources of a bin (128 accumulators) and X is loaded once per bin: lane l loads the complex sample of channel
// l & 15 at frame t + (l >> 4) -- one 8-byte load per 4 frames, 128 contiguous bytes per frame.
// ---------------------------------------------------------------------------------------------
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int KW, bool UNIT>
__global__ __launch_bounds__(64, 2) void cov_mfma16_kernel(const float2* __restrict__ X, const float* __restrict__ Wt,
                                                           float* __restrict__ Vpart, int T, int F, int M, int K, int Kp,
                                                           int tc) {
    const int lane = threadIdx.x;
    const int ch = lane & 15;           // channel: row of A, column of B
    const int kf = lane >> 4;           // frame within the group of 4
    const int f = blockIdx.x;
    const int k0 = blockIdx.z * KW;
    const int t_begin = blockIdx.y * tc;
    const int t_end = min(T, t_begin + tc);
    const int NA = M * M;
    const bool cvalid = ch < M;
    const float xmask = cvalid ? 1.f : 0.f;
    // arithmetic select of the lane's frame among the group's four (the weights are wave-uniform scalars)
    const float m0 = kf == 0 ? 1.f : 0.f, m1 = kf == 1 ? 1.f : 0.f, m2 = kf == 2 ? 1.f : 0.f, m3 = kf == 3 ? 1.f : 0.f;
    f32x4 are[KW], air[KW];
#pragma unroll
    for (int kk = 0; kk < KW; ++kk)
#pragma unroll
        for (int r = 0; r < 4; ++r) are[kk][r] = air[kk][r] = 0.f;

    const size_t frame_stride = (size_t)F * M;                         // complex samples per frame
    const float2* pcol = X + (size_t)f * M + (cvalid ? ch : 0);
    const int ngroups = (t_end - t_begin + 3) >> 2;
    // x of group g+1 is requested before the 3*KW MFMAs of group g are issued (one wave may be alone on its
    // SIMD: 128 accumulators + operands leave room for two waves at most)
    auto fetch = [&](int g) {
        const int t = t_begin + 4 * g + kf;
        return pcol[(size_t)(t < t_end ? t : T - 1) * frame_stride];
    };
    // kDepth groups of x are in flight per wave (8 bytes per lane and group): with few sources the MFMA work
    // per group is short and a single outstanding load leaves the wave waiting on HBM
    constexpr int kDepth = 4;
    float2 xq[kDepth];
#pragma unroll
    for (int u = 0; u < kDepth; ++u) xq[u] = fetch(u < ngroups ? u : ngroups - 1);
    for (int g0 = 0; g0 < ngroups; g0 += kDepth) {
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
            const int g = g0 + u;
            const float2 x = xq[u];
            xq[u] = fetch(g + kDepth < ngroups ? g + kDepth : ngroups - 1);
            if (g < ngroups) {                                         // uniform
                const int t0 = t_begin + 4 * g;
                const float live = (t0 + kf < t_end) ? xmask : 0.f;
                const float xr = x.x * live, xi = x.y * live;
                // Wt is (T, Kp) with zero padding columns; rows past T-1 in the last group are clamped (uniform)
                const float* w0 = Wt + (size_t)min(t0, T - 1) * Kp + k0;
                const float* w1 = Wt + (size_t)min(t0 + 1, T - 1) * Kp + k0;
                const float* w2 = Wt + (size_t)min(t0 + 2, T - 1) * Kp + k0;
                const float* w3 = Wt + (size_t)min(t0 + 3, T - 1) * Kp + k0;
#pragma unroll
                for (int kk = 0; kk < KW; ++kk) {
                    float w = 1.f;
                    if constexpr (!UNIT) w = m0 * w0[kk] + m1 * w1[kk] + m2 * w2[kk] + m3 * w3[kk];
                    const float ar = xr * w, ai = xi * w;
                    are[kk] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar, xr, are[kk], 0, 0, 0);
                    are[kk] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, xi, are[kk], 0, 0, 0);
                    air[kk] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, xr, air[kk], 0, 0, 0);
                }
            }
        }
    }

    // C/D layout of 16x16: lane l, register r holds [row = (l >> 4) * 4 + r][col = l & 15].
    // V_im[c][d] needs G_ir[d][c]: held by lane (d >> 2) * 16 + c in register d & 3 = l & 3.
    const int d = ch;
    const int sel = lane & 3;
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
        const int k = k0 + kk;
        float* out = Vpart + (((size_t)blockIdx.y * F + f) * K + (k < K ? k : 0)) * NA;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = kf * 4 + r;
            const int src = (d >> 2) * 16 + c;
            const float t0 = __shfl(air[kk][0], src, 64), t1 = __shfl(air[kk][1], src, 64);
            const float t2 = __shfl(air[kk][2], src, 64), t3 = __shfl(air[kk][3], src, 64);
            const float gt = sel == 0 ? t0 : sel == 1 ? t1 : sel == 2 ? t2 : t3;      // G_ir[d][c]
            if (k < K && c < M && d < M && c <= d) {
                if (c == d) {
                    out[c] = are[kk][r];
                } else {
                    const int o = herm_pair_index(M, c, d);
                    out[o] = are[kk][r];
                    out[o + 1] = air[kk][r] - gt;
                }
            }
        }
    }
}

}  // namespace

int cov_mfma_sources_per_wave(int K) { return K == 1 ? 1 : (K == 2 ? 2 : 4); }

hipError_t launch_cov_mfma(hipStream_t s, const float2* X, const float* R, float* Wt, float* wscale, int model, int raw,
                           float* Vpart, int T, int F, int M, int K, int nsplit, int tc) {
    const bool unit = R == nullptr;
    if (!unit) {
        if (Wt == nullptr) return hipErrorInvalidValue;
    }
    const int Kp = (K + 15) / 16 * 16;      // padded row stride of the weights (scratch holds T * 16 floats)
    if (!unit) {
        weights_kernel<<<dim3((T + kBlock - 1) / kBlock), dim3(kBlock), 0, s>>>(R, Wt, wscale, model, raw, T, K, Kp);
    }
    static const bool use32 = getenv("OIVA_MFMA32") != nullptr;   // A/B: the 32x32x2 real-Gram kernel
    if (!use32) {
        const int kw16 = unit ? 1 : (K <= 2 ? 2 : (K <= 4 ? 4 : (K <= 8 ? 8 : 16)));
        dim3 grid16(F, nsplit, unit ? 1 : (K + kw16 - 1) / kw16);
        if (unit)
            cov_mfma16_kernel<1, true><<<grid16, dim3(64), 0, s>>>(X, Wt, Vpart, T, F, M, K, Kp, tc);
        else if (kw16 == 2)
            cov_mfma16_kernel<2, false><<<grid16, dim3(64), 0, s>>>(X, Wt, Vpart, T, F, M, K, Kp, tc);
        else if (kw16 == 4)
            cov_mfma16_kernel<4, false><<<grid16, dim3(64), 0, s>>>(X, Wt, Vpart, T, F, M, K, Kp, tc);
        else if (kw16 == 8)
            cov_mfma16_kernel<8, false><<<grid16, dim3(64), 0, s>>>(X, Wt, Vpart, T, F, M, K, Kp, tc);
        else
            cov_mfma16_kernel<16, false><<<grid16, dim3(64), 0, s>>>(X, Wt, Vpart, T, F, M, K, Kp, tc);
        return hipGetLastError();
    }
    const int kw = unit ? 1 : cov_mfma_sources_per_wave(K);
    const int waves = (K + kw - 1) / kw;
    if (waves > 4 || M > 16) return hipErrorInvalidValue;
    dim3 grid(F, nsplit);
    dim3 block(64 * waves);
    const float* Xf = reinterpret_cast<const float*>(X);
    if (unit)
        cov_mfma_kernel<1, true><<<grid, block, 0, s>>>(Xf, Wt, Vpart, T, F, M, K, Kp, tc);
    else if (kw == 1)
        cov_mfma_kernel<1, false><<<grid, block, 0, s>>>(Xf, Wt, Vpart, T, F, M, K, Kp, tc);
    else if (kw == 2)
        cov_mfma_kernel<2, false><<<grid, block, 0, s>>>(Xf, Wt, Vpart, T, F, M, K, Kp, tc);
    else
        cov_mfma_kernel<4, false><<<grid, block, 0, s>>>(Xf, Wt, Vpart, T, F, M, K, Kp, tc);
    return hipGetLastError();
}

}  // namespace oiva
