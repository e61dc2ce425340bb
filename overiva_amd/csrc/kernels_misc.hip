// Small kernels: source-activation finalisation (reference overiva.py:152-173), fixed-order sums of
// partial buffers, unpacking of packed Hermitian matrices.
#include "oiva_device.h"

#include <algorithm>

namespace oiva {
namespace {

// Source activation, reference overiva.py:152-155:
//   R[t,k] = 2 sqrt(p) (laplace) | p / F_total (gauss), p = sum over parts in part order (bin batch, or rank then
//   batch).  One thread per frame adds the parts strictly in order, so parts that are all zero (the padding that
//   equalises the ranks' messages in a bin-sharded run) change nothing: a sharded run whose shard boundaries fall
//   on 64-bin batches gets the same bits as the single-GPU run.  The loads of a group of 8 parts are issued
//   together; the adds are sequential.  Each block (kBlock frames of one source) also leaves the float64 sum of its r per
//   source behind R (rsum_offset_floats): the consumers derive gamma (overiva.py:158) from those few values
//   (gamma_of) instead of re-reducing the T activations in every workgroup.
__global__ __launch_bounds__(kBlock) void activation_kernel(const float* __restrict__ parts, int nparts,
                                                           float* __restrict__ R, int T, int K, int model,
                                                           float inv_f_total) {
    __shared__ double wsum[kWaves];
    const int t = blockIdx.x * kBlock + threadIdx.x;      // grid = (blocks of kBlock frames, sources)
    const int k = blockIdx.y;
    const size_t n = (size_t)T * K;
    float r = 0.f;
    if (t < T) {
        const size_t e = (size_t)t * K + k;
        float p = 0.f;
        // the adds are sequential in part order whatever the grouping of the loads; 32 loads in flight = the 2048-bin
        // single-GPU case in one memory round trip instead of four
        int i = 0;
        for (; i + 32 <= nparts; i += 32) {
            float v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = parts[(size_t)(i + u) * n + e];
#pragma unroll
            for (int u = 0; u < 32; ++u) p += v[u];
        }
        for (; i + 8 <= nparts; i += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = parts[(size_t)(i + u) * n + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) p += v[u];
        }
        for (; i < nparts; ++i) p += parts[(size_t)i * n + e];
        r = model == OIVA_MODEL_LAPLACE ? 2.f * sqrtf(p) : (model == kModelOgiveLaplace ? sqrtf(p * inv_f_total) : p * inv_f_total);
        R[e] = r;
    }
    double s = (double)r;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) tot += wsum[w];
        reinterpret_cast<double*>(R + rsum_offset_floats(T, K))[(size_t)blockIdx.x * K + k] = tot;
    }
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

// complex128 <-> complex64 of a dense array (the device holds X and Y as complex64 whatever the caller's dtype)
template <typename SRC, typename DST>
__global__ __launch_bounds__(kBlock) void cast_complex_kernel(const SRC* __restrict__ in, DST* __restrict__ out, long long n) {
    for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < n; e += (long long)gridDim.x * kBlock) {
        const SRC v = in[e];
        DST o;
        o.x = v.x;
        o.y = v.y;
        out[e] = o;
    }
}

}  // namespace

hipError_t launch_cast_c128_to_c64(hipStream_t s, const double2* in, float2* out, long long n) {
    const unsigned grid = (unsigned)std::min<long long>((n + kBlock - 1) / kBlock, 1 << 16);
    hipLaunchKernelGGL((cast_complex_kernel<double2, float2>), dim3(grid), dim3(kBlock), 0, s, in, out, n);
    return hipGetLastError();
}
hipError_t launch_cast_c64_to_c128(hipStream_t s, const float2* in, double2* out, long long n) {
    const unsigned grid = (unsigned)std::min<long long>((n + kBlock - 1) / kBlock, 1 << 16);
    hipLaunchKernelGGL((cast_complex_kernel<float2, double2>), dim3(grid), dim3(kBlock), 0, s, in, out, n);
    return hipGetLastError();
}

hipError_t launch_activation(hipStream_t s, const float* parts, int nparts, float* R, int T, int K, int model,
                             int F_total) {
    hipLaunchKernelGGL(activation_kernel, dim3((unsigned)rsum_blocks(T), (unsigned)K), dim3(kBlock), 0, s, parts, nparts, R, T, K,
                       model, 1.f / (float)F_total);
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
