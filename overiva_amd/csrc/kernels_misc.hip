// Small kernels: source-activation finalisation (reference overiva.py:152-173), fixed-order sums of
// partial buffers, unpacking of packed Hermitian matrices.
#include "oiva_internal.h"

namespace oiva {
namespace {

constexpr float kEpsR = 1e-15f;  // overiva.py:170

// block-wide sum of one double per thread (fixed order: wave shuffle tree, then waves in order)
__device__ __forceinline__ double block_sum(double v, double* scratch /* [kWaves] */) {
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

// R[e] = 2 sqrt(p) | p / F_total with p = sum over parts in part order (bin batch / rank order), e = t*K + k.
// kRsumLanes lanes share one element: lane l adds parts l, l+8, ...; a fixed shuffle tree adds the lanes.
constexpr int kRsumLanes = 8;
__global__ __launch_bounds__(kBlock) void rsum_kernel(const float* __restrict__ parts, int nparts,
                                                      float* __restrict__ R, long long n, int model,
                                                      float inv_f_total) {
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long e = gid / kRsumLanes;
    const int l = (int)(gid % kRsumLanes);
    float p = 0.f;
    if (e < n) {
#pragma unroll 4
        for (int i = l; i < nparts; i += kRsumLanes) p += parts[(size_t)i * n + e];
    }
#pragma unroll
    for (int off = 1; off < kRsumLanes; off <<= 1) p += __shfl_xor(p, off, kRsumLanes);
    if (e < n && l == 0) R[e] = model == OIVA_MODEL_LAPLACE ? 2.f * sqrtf(p) : p * inv_f_total;
}

// gamma_k = mean_t R (every block re-reduces the whole (T,K) array in a fixed order: it is a few KB);
// Rinv = 1 / max(R / gamma, eps) ; wscale = gamma (laplace) | sqrt(gamma) (gauss)
__global__ __launch_bounds__(kBlock) void rfin_kernel(const float* __restrict__ R, float* __restrict__ Rinv,
                                                      float* __restrict__ wscale, int T, int K, int model) {
    __shared__ double scratch[kWaves];
    __shared__ float gam[OIVA_MAX_CHANNELS];
    for (int k = 0; k < K; ++k) {
        double s = 0.;
        for (int t = threadIdx.x; t < T; t += kBlock) s += (double)R[(size_t)t * K + k];
        s = block_sum(s, scratch);
        if (threadIdx.x == 0) gam[k] = (float)(s / (double)T);
    }
    __syncthreads();
    const int t = blockIdx.x * kBlock + threadIdx.x;
    for (int k = 0; k < K; ++k) {
        const float gamma = gam[k];
        if (t < T) {
            float rn = R[(size_t)t * K + k] / gamma;
            rn = rn < kEpsR ? kEpsR : rn;   // NaN stays NaN, as r[r < eps] = eps does in the reference
            Rinv[(size_t)t * K + k] = 1.f / rn;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) wscale[k] = model == OIVA_MODEL_LAPLACE ? gamma : sqrtf(gamma);
    }
}

__global__ __launch_bounds__(kBlock) void sum_parts_kernel(const float* __restrict__ parts, int nparts,
                                                           float* __restrict__ out, long long n, float scale) {
    const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (e >= n) return;
    double s = 0.;
    for (int i = 0; i < nparts; ++i) s += (double)parts[(size_t)i * n + e];
    out[e] = (float)(s * (double)scale);
}

// packed Hermitian (M*M floats) -> full complex M x M
__global__ __launch_bounds__(kBlock) void unpack_herm_kernel(const float* __restrict__ packed,
                                                             float2* __restrict__ full, long long nmat, int M,
                                                             float scale) {
    const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;
    const int NA = M * M;
    if (e >= nmat * NA) return;
    const long long mat = e / NA;
    const int ij = (int)(e - mat * NA);
    const int i = ij / M, j = ij - i * M;
    const float* p = packed + mat * NA;
    float2 v;
    if (i == j) {
        v = make_float2(p[i], 0.f);
    } else if (i < j) {
        const int o = herm_pair_index(M, i, j);
        v = make_float2(p[o], p[o + 1]);
    } else {
        const int o = herm_pair_index(M, j, i);
        v = make_float2(p[o], -p[o + 1]);
    }
    full[e] = make_float2(v.x * scale, v.y * scale);
}

}  // namespace

int rsum_blocks(int T) { return (T + kBlock - 1) / kBlock; }

hipError_t launch_rsum(hipStream_t s, const float* parts, int nparts, float* R, int T, int K, int model, int F_total) {
    const long long n = (long long)T * K;
    const long long threads = n * kRsumLanes;
    hipLaunchKernelGGL(rsum_kernel, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, parts, nparts,
                       R, n, model, 1.f / (float)F_total);
    return hipGetLastError();
}

hipError_t launch_rfin(hipStream_t s, const float* R, float* Rinv, float* wscale, int T, int K, int model) {
    hipLaunchKernelGGL(rfin_kernel, dim3(rsum_blocks(T)), dim3(kBlock), 0, s, R, Rinv, wscale, T, K, model);
    return hipGetLastError();
}

hipError_t launch_sum_parts(hipStream_t s, const float* parts, int nparts, float* out, long long n, float scale) {
    hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, parts, nparts,
                       out, n, scale);
    return hipGetLastError();
}

hipError_t launch_unpack_herm(hipStream_t s, const float* packed, float2* full, long long nmat, int M, float scale) {
    const long long n = nmat * M * M;
    hipLaunchKernelGGL(unpack_herm_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, packed, full,
                       nmat, M, scale);
    return hipGetLastError();
}

}  // namespace oiva
