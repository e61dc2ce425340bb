// Small kernels: source-activation finalisation (reference overiva.py:152-173), fixed-order sums of
// partial buffers, unpacking of packed Hermitian matrices.
#include "oiva_device.h"

namespace oiva {
namespace {

// Source activation, reference overiva.py:152-155:
//   R[e] = 2 sqrt(p) (laplace) | p / F_total (gauss), p = sum over parts in part order (bin batch or rank
//   order), e = t*K + k.  kActLanes lanes share one element: lane l adds parts l, l+8, ...; a fixed
//   shuffle tree adds the lanes.  The scale normalisation (gamma, overiva.py:158-173) is applied by the
//   consumers (covariance and update kernels) through block_gamma() / activation_weight().
constexpr int kActLanes = 8;
__global__ __launch_bounds__(kBlock) void activation_kernel(const float* __restrict__ parts, int nparts,
                                                           float* __restrict__ R, long long n, int model,
                                                           float inv_f_total) {
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long e = gid / kActLanes;
    const int l = (int)(gid % kActLanes);
    float p = 0.f;
    if (e < n) {
#pragma unroll 4
        for (int i = l; i < nparts; i += kActLanes) p += parts[(size_t)i * n + e];
    }
#pragma unroll
    for (int off = 1; off < kActLanes; off <<= 1) p += __shfl_xor(p, off, kActLanes);
    if (e < n && l == 0) R[e] = model == OIVA_MODEL_LAPLACE ? 2.f * sqrtf(p) : p * inv_f_total;
}

template <typename IN>
__global__ __launch_bounds__(kBlock) void sum_parts_kernel(const IN* __restrict__ parts, int nparts,
                                                           double* __restrict__ out, long long n, double scale) {
    const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (e >= n) return;
    double s = 0.;
    for (int i = 0; i < nparts; ++i) s += (double)parts[(size_t)i * n + e];
    out[e] = s * scale;
}

// packed Hermitian (M*M float64) -> full complex M x M (complex64 or complex128)
template <typename C2>
__global__ __launch_bounds__(kBlock) void unpack_herm_kernel(const double* __restrict__ packed, C2* __restrict__ full,
                                                             long long nmat, int M) {
    const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;
    const int NA = M * M;
    if (e >= nmat * NA) return;
    const long long mat = e / NA;
    const int ij = (int)(e - mat * NA);
    const int i = ij / M, j = ij - i * M;
    const double* p = packed + mat * NA;
    double re, im = 0.;
    if (i == j) {
        re = p[i];
    } else if (i < j) {
        const int o = herm_pair_index(M, i, j);
        re = p[o];
        im = p[o + 1];
    } else {
        const int o = herm_pair_index(M, j, i);
        re = p[o];
        im = -p[o + 1];
    }
    C2 v;
    v.x = re;
    v.y = im;
    full[e] = v;
}

}  // namespace

hipError_t launch_activation(hipStream_t s, const float* parts, int nparts, float* R, int T, int K, int model,
                             int F_total) {
    const long long n = (long long)T * K;
    const long long threads = n * kActLanes;
    hipLaunchKernelGGL(activation_kernel, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, parts,
                       nparts, R, n, model, 1.f / (float)F_total);
    return hipGetLastError();
}

hipError_t launch_sum_parts(hipStream_t s, const void* parts, bool f64, int nparts, double* out, long long n, double scale) {
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    if (f64)
        hipLaunchKernelGGL(sum_parts_kernel<double>, grid, dim3(kBlock), 0, s, static_cast<const double*>(parts), nparts, out, n, scale);
    else
        hipLaunchKernelGGL(sum_parts_kernel<float>, grid, dim3(kBlock), 0, s, static_cast<const float*>(parts), nparts, out, n, scale);
    return hipGetLastError();
}

hipError_t launch_unpack_herm(hipStream_t s, const double* packed, void* full, bool out_f64, long long nmat, int M) {
    const long long n = nmat * M * M;
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    if (out_f64)
        hipLaunchKernelGGL(unpack_herm_kernel<double2>, grid, dim3(kBlock), 0, s, packed, static_cast<double2*>(full), nmat, M);
    else
        hipLaunchKernelGGL(unpack_herm_kernel<float2>, grid, dim3(kBlock), 0, s, packed, static_cast<float2*>(full), nmat, M);
    return hipGetLastError();
}

}  // namespace oiva
