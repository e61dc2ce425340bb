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

// The same with the bins sharded over the GPUs of a node and NO collective and NO host in the loop (the iteration stays a
// captured graph of four kernels): every thread adds its rank's parts in order (the rank's sum of |y|^2 for its frame and
// source), stores it -- one naturally aligned 8-byte word {value, epoch}, system scope -- into its slot of every OTHER
// rank's gather buffer (peer stores over xGMI: T K 8 bytes per peer and iteration, 64 KB at the headline shape), polls its
// own buffer until the other ranks' words carry this epoch, and adds the ranks' sums in rank order: the same bits of r on
// every rank.  (Against the collective path, which gathers every 64-bin part and adds them one by one, the sum is
// associated rank by rank: r differs in its last bits.)  The buffers alternate with the epoch's parity -- a rank can be one
// iteration ahead of the slowest reader of its stores, never two: it cannot finish epoch e + 1 before every rank has
// stored e + 1, which a rank does only after it has read epoch e.  The epoch is counted on the device, one word per
// workgroup (read at the start, advanced at the end by the workgroup itself), so a replayed graph needs no new arguments.
// A wait gives up after `timeout` ticks of the 100 MHz clock and records it in ctrl[0]; the host looks at its next
// synchronisation.  loopback: one GPU plays all `world` ranks (own buffer, the phantom ranks' sums are zeros).
struct ActXchgArgs {
    unsigned long long* gath[OIVA_XCHG_MAX_RANKS];      // every rank's [2][world][T * K] words, as mapped here
    int rank, world, loopback;                          // loopback 2: the phantom ranks never store (test hook: a rank that does not deliver)
    unsigned* epochs;                                   // [gridDim.y][gridDim.x]
    unsigned* ctrl;
    long long timeout;
};
__global__ __launch_bounds__(kBlock) void activation_xchg_kernel(const float* __restrict__ parts, int nparts, ActXchgArgs a,
                                                                float* __restrict__ R, int T, int K, int model, float inv_f_total) {
    __shared__ double wsum[kWaves];
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const int k = blockIdx.y;
    const size_t n = (size_t)T * K;
    unsigned* my_epoch = a.epochs + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned epoch = *my_epoch + 1u;
    const int par = (int)(epoch & 1u);
    float r = 0.f;
    if (t < T) {
        const size_t e = (size_t)t * K + k;
        float p = 0.f;
        for (int i = 0; i < nparts; i += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = i + u < nparts ? parts[(size_t)(i + u) * n + e] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) p += v[u];              // (a + 0.f is exact: the padding changes nothing)
        }
        const unsigned long long mine = (unsigned long long)__float_as_uint(p) | ((unsigned long long)epoch << 32);
        const size_t base = (size_t)par * a.world * n + e;
        for (int q = 0; q < a.world && a.loopback != 2; ++q) {
            if (q == a.rank) continue;
            unsigned long long* dst = a.loopback ? a.gath[a.rank] + base + (size_t)q * n : a.gath[q] + base + (size_t)a.rank * n;
            __hip_atomic_store(dst, a.loopback ? ((unsigned long long)epoch << 32) : mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const unsigned long long* own = a.gath[a.rank] + base;
        float tot = 0.f;
        const long long t0 = wall_clock64();
        for (int q0 = 0; q0 < a.world; q0 += 8) {
            float v[8];
            for (unsigned spins = 1;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int q = q0 + u < a.world ? q0 + u : a.world - 1;
                    if (q == a.rank) {
                        v[u] = p;
                        continue;
                    }
                    const unsigned long long x = __hip_atomic_load(own + (size_t)q * n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    ok = ok && (unsigned)(x >> 32) == epoch;
                    v[u] = __uint_as_float((unsigned)x);
                }
                if (ok) break;
                if ((spins & 15u) == 0u && (wall_clock64() - t0 > a.timeout || __hip_atomic_load(a.ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    atomicCAS(a.ctrl, 0u, 1u + (unsigned)blockIdx.x);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) tot += q0 + u < a.world ? v[u] : 0.f;          // rank order
        }
        r = model == OIVA_MODEL_LAPLACE ? 2.f * sqrtf(tot) : (model == kModelOgiveLaplace ? sqrtf(tot * inv_f_total) : tot * inv_f_total);
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
        *my_epoch = epoch;
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

hipError_t launch_activation_xchg(hipStream_t s, const float* parts, int nparts, char* const* gath, int rank, int world, int loopback,
                                  unsigned* epochs, unsigned* ctrl, long long timeout_ticks, float* R, int T, int K, int model, int F_total) {
    ActXchgArgs a;
    for (int r = 0; r < OIVA_XCHG_MAX_RANKS; ++r) a.gath[r] = r < world ? reinterpret_cast<unsigned long long*>(gath[r]) : nullptr;
    a.rank = rank;
    a.world = world;
    a.loopback = loopback;
    a.epochs = epochs;
    a.ctrl = ctrl;
    a.timeout = timeout_ticks;
    hipLaunchKernelGGL(activation_xchg_kernel, dim3((unsigned)rsum_blocks(T), (unsigned)K), dim3(kBlock), 0, s, parts, nparts, a, R, T, K, model,
                       1.f / (float)F_total);
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
