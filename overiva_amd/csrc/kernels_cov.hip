// Weighted spatial covariance pass (the dominant kernel of the iteration).
//
//   V_k[f] = sum_t rinv[t,k] * x_{t,f} x_{t,f}^H          reference overiva.py:179 (all k in one pass)
//   Cx[f]  = sum_t x_{t,f} x_{t,f}^H                      reference overiva.py:87   (unit weights)
//
// Reads the native (T, F, M) complex64 tensor once for KC sources.  Lane = (bin, frame phase): every
// lane owns the whole M-vector of one bin at one frame, so the Hermitian outer product is formed in
// registers with no cross-lane traffic and only its upper triangle is computed (M^2 real products
// per frame instead of 4 M^2).  Per-lane partial sums cover ~T/(16*nsplit) frames; the 16 frame
// phases of a block are combined through LDS (fixed order) and written as one packed partial per
// (frame split, bin, source).  The per-bin update kernel adds the nsplit partials in fp64.
#include "oiva_internal.h"

namespace oiva {
namespace {

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

// acc[kk][*] += w[kk] * pack(x x^H)
template <int M, int KC, bool UNIT>
__device__ __forceinline__ void accumulate(float (&acc)[KC][M * M], const float (&xr)[M], const float (&xi)[M],
                                           const float (&w)[KC]) {
    if constexpr (KC == 1) {
        // one source: scale x once, then 4 FMAs per complex pair
        float sr[M], si[M];
#pragma unroll
        for (int c = 0; c < M; ++c) {
            sr[c] = xr[c] * w[0];   // UNIT: w is 1 (live frame) or 0 (clamped tail frame)
            si[c] = xi[c] * w[0];
        }
#pragma unroll
        for (int c = 0; c < M; ++c) acc[0][c] = fmaf(sr[c], xr[c], fmaf(si[c], xi[c], acc[0][c]));
        int a = M;
#pragma unroll
        for (int c = 0; c < M; ++c) {
#pragma unroll
            for (int d = c + 1; d < M; ++d) {
                acc[0][a] = fmaf(sr[c], xr[d], fmaf(si[c], xi[d], acc[0][a]));           // Re x_c conj(x_d)
                acc[0][a + 1] = fmaf(si[c], xr[d], fmaf(-sr[c], xi[d], acc[0][a + 1]));  // Im x_c conj(x_d)
                a += 2;
            }
        }
    } else {
        // several sources: form each product once, one FMA per source
#pragma unroll
        for (int c = 0; c < M; ++c) {
            const float p = fmaf(xr[c], xr[c], xi[c] * xi[c]);
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) acc[kk][c] = fmaf(w[kk], p, acc[kk][c]);
        }
        int a = M;
#pragma unroll
        for (int c = 0; c < M; ++c) {
#pragma unroll
            for (int d = c + 1; d < M; ++d) {
                const float pre = fmaf(xr[c], xr[d], xi[c] * xi[d]);
                const float pim = fmaf(xi[c], xr[d], -(xr[c] * xi[d]));
#pragma unroll
                for (int kk = 0; kk < KC; ++kk) {
                    acc[kk][a] = fmaf(w[kk], pre, acc[kk][a]);
                    acc[kk][a + 1] = fmaf(w[kk], pim, acc[kk][a + 1]);
                }
                a += 2;
            }
        }
    }
}

constexpr int kChunk = 16;               // accumulators combined per LDS round
constexpr int kLdsStride = kBlock + 1;   // +1: conflict-free transposed read

template <int M, int KC, bool UNIT>
__global__ __launch_bounds__(kBlock) void cov_kernel(const float2* __restrict__ X, const float* __restrict__ rinv,
                                                     float* __restrict__ Vpart, int T, int F, int K, int tc) {
    constexpr int NA = M * M;
    constexpr int NACC = NA * KC;
    __shared__ float lds[kChunk * kLdsStride];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int b = lane & (kBinsPerWave - 1);
    const int q = wave * kPhasesPerWave + (lane >> 4);  // 0..15, == tid >> 4
    const int f = blockIdx.x * kBinsPerWave + b;
    const int fc = f < F ? f : F - 1;
    const int k0 = blockIdx.z * KC;
    const int t_begin = blockIdx.y * tc;
    const int t_end = min(T, t_begin + tc);
    const int nsteps = (t_end - t_begin + 15) >> 4;

    float acc[KC][NA];
#pragma unroll
    for (int kk = 0; kk < KC; ++kk)
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[kk][a] = 0.f;

    const size_t frame_stride = (size_t)F * M;
    const float2* px = X + ((size_t)(t_begin + q) * F + fc) * M;

    for (int i = 0; i < nsteps; ++i) {
        const int t = t_begin + q + 16 * i;
        const bool live = t < t_end;
        float xr[M], xi[M], w[KC];
        // frames past the end are clamped (address stays legal) and weighted by 0
        load_x<M>(live ? px : X + ((size_t)(T - 1) * F + fc) * M, xr, xi);
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            if constexpr (UNIT) {
                w[kk] = live ? 1.f : 0.f;
            } else {
                const int k = k0 + kk;
                w[kk] = (live && k < K) ? rinv[(size_t)t * K + k] : 0.f;
            }
        }
        accumulate<M, KC, UNIT>(acc, xr, xi, w);
        px += 16 * frame_stride;
    }

    // Combine the 16 frame phases (tid = q*16 + b) in rounds of kChunk accumulators through LDS:
    //   write lds[a][tid]; thread (bb = tid/16, aa = tid%16) sums lds[aa][qq*16 + bb] over qq.
    const int bb = tid >> 4, aa = tid & 15;
    const int fo = blockIdx.x * kBinsPerWave + bb;
    float* out = Vpart + (((size_t)blockIdx.y * F + fo) * K + k0) * NA;
#pragma unroll
    for (int r0 = 0; r0 < NACC; r0 += kChunk) {
        __syncthreads();
#pragma unroll
        for (int a = 0; a < kChunk; ++a) {
            if (r0 + a < NACC) lds[a * kLdsStride + tid] = acc[(r0 + a) / NA][(r0 + a) % NA];
        }
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) s += lds[aa * kLdsStride + qq * 16 + bb];
        const int e = r0 + aa;          // accumulator index = kk*NA + a
        const int kk = e / NA;          // constant-folded per round when NA % 16 == 0
        if (e < NACC && fo < F && k0 + kk < K) out[e] = s;
    }
}

template <int M, int KC>
hipError_t launch_one(hipStream_t s, const float2* X, const float* rinv, float* Vpart, int T, int F, int K,
                      const CovGeom& g) {
    dim3 grid(g.nbg, g.nsplit, (K + KC - 1) / KC);
    if (rinv == nullptr) {
        if constexpr (KC == 1) {
            hipLaunchKernelGGL((cov_kernel<M, 1, true>), grid, dim3(kBlock), 0, s, X, rinv, Vpart, T, F, K, g.tc);
        } else {
            return hipErrorInvalidValue;
        }
    } else {
        hipLaunchKernelGGL((cov_kernel<M, KC, false>), grid, dim3(kBlock), 0, s, X, rinv, Vpart, T, F, K, g.tc);
    }
    return hipGetLastError();
}

template <int M>
hipError_t launch_m(hipStream_t s, const float2* X, const float* rinv, float* Vpart, int T, int F, int K,
                    const CovGeom& g) {
    switch (g.kc) {
        case 1:
            return launch_one<M, 1>(s, X, rinv, Vpart, T, F, K, g);
        case 2:
            if constexpr (M * M * 2 <= 144) return launch_one<M, 2>(s, X, rinv, Vpart, T, F, K, g);
            break;
        case 4:
            if constexpr (M * M * 4 <= 144) return launch_one<M, 4>(s, X, rinv, Vpart, T, F, K, g);
            break;
    }
    return hipErrorInvalidValue;
}

}  // namespace

bool cov_supported(int M) { return M >= 1 && M <= 8; }

// sources handled per pass over X: as many as fit the accumulator budget (KC * M^2 <= 144 registers)
int cov_sources_per_pass(int M, int K) {
    int kc = 1;
    if (K >= 2 && M * M * 2 <= 144) kc = 2;
    if (K >= 3 && M * M * 4 <= 144) kc = 4;
    return kc;
}

hipError_t launch_cov(hipStream_t s, const float2* X, const float* rinv, float* Vpart, int T, int F, int M, int K,
                      const CovGeom& g) {
    switch (M) {
        case 1: return launch_m<1>(s, X, rinv, Vpart, T, F, K, g);
        case 2: return launch_m<2>(s, X, rinv, Vpart, T, F, K, g);
        case 3: return launch_m<3>(s, X, rinv, Vpart, T, F, K, g);
        case 4: return launch_m<4>(s, X, rinv, Vpart, T, F, K, g);
        case 5: return launch_m<5>(s, X, rinv, Vpart, T, F, K, g);
        case 6: return launch_m<6>(s, X, rinv, Vpart, T, F, K, g);
        case 7: return launch_m<7>(s, X, rinv, Vpart, T, F, K, g);
        case 8: return launch_m<8>(s, X, rinv, Vpart, T, F, K, g);
    }
    return hipErrorInvalidValue;
}

}  // namespace oiva
