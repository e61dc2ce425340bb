// Small kernels: source-activation finalisation (reference overiva.py:152-173), fixed-order sums of
// partial buffers, unpacking of packed Hermitian matrices.
#include "oiva_device.h"

#include <algorithm>

namespace oiva {
namespace {

// Source activation, reference overiva.py:152-155:
//   R[t,k] = 2 sqrt(p) (laplace) | p / F_total (gauss), p = sum over parts in part order (bin batch, or rank then
//   batch), associated in the canonical blocks described in the kernel: a sharded run with equal shards on 64-bin
//   batches gets the same bits as the single-GPU run -- of r; of W too where the plans also split the frame axis of the covariance
//   pass alike.  (16 channels with more than 4 sources: the 64-bin parts themselves come from power_lds_kernel where
//   a plan's bin count is a multiple of 64 and from power_mfma_kernel otherwise, which add a part's 64 bins in different orders: a
//   shard of ragged size -- a world that does not divide the 64-bin batches -- agrees with one GPU to rounding only)
//   The loads of a group of 8 parts are issued together.  Each block (kBlock frames of one source) also leaves the float64 sum of its r per
//   source behind R (rsum_offset_floats): the consumers derive gamma (overiva.py:158) from those few values
//   (gamma_of) instead of re-reducing the T activations in every workgroup.
// Canonical sum over `count` parts starting at part `first` of one (frame, source) element -- the parts in blocks of `bs`
// consecutive ones counted from part 0, each block added sequentially, then the block sums sequentially (see
// activation_kernel) -- with the loads of up to NP parts IN FLIGHT TOGETHER: the kernel is a chain of round trips to
// data other XCDs wrote, and round 4's form made one trip per block (8 at the headline shape: 5.6 us against 4.5 for the
// plain sequential sum of round 3).  State (p, pb) carries over chunks of NP parts; a block boundary is a select.
template <int NP>
struct CanonSum {
    float p = 0.f, pb = 0.f;
    // parts [i0, i0 + NP) of which those < i1 exist; a block ends after part i when (i + 1) % bs == 0
    __device__ __forceinline__ void chunk(const float* __restrict__ parts, size_t n, size_t e, int i0, int i1, int bs) {
        float v[NP];
#pragma unroll
        for (int u = 0; u < NP; ++u) v[u] = i0 + u < i1 ? parts[(size_t)(i0 + u) * n + e] : 0.f;
        int left = bs - i0 % bs;                       // parts until the current block is complete
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            pb += v[u];                                // (a + 0.f is exact: parts past i1 change nothing)
            const bool end = --left == 0;
            p = end ? p + pb : p;
            pb = end ? 0.f : pb;
            left = end ? bs : left;
        }
    }
    __device__ __forceinline__ float total() const { return p + pb; }       // the last block may be short (pb = 0.f if not: exact)
};

template <int NP>
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
        // Canonical order of the sum over the parts (all paths of the library, whatever the number of GPUs): the parts in
        // blocks of ceil(nparts / 8) consecutive ones -- at most 8 blocks --, each block added sequentially, then the block
        // sums sequentially.  A rank of a sharded run that holds whole blocks can send their sums instead of its parts and
        // every rank still forms the SAME sum: the same bits of r at 1, 2, 4 and 8 GPUs (activation_xchg_kernel).  Up to 8
        // parts the order is the plain sequential one.
        const int bs = (nparts + kCanonBlocks - 1) / kCanonBlocks;
        CanonSum<NP> cs;
        for (int i0 = 0; i0 < nparts; i0 += NP) cs.chunk(parts, n, e, i0, nparts, bs);
        // (a short last block: p + pb; complete blocks leave pb = 0.f and p + 0.f is exact -- but p = 0.f + pb for a single
        //  block must not become (0.f + pb) + 0.f with a different rounding: it is not, x + 0.f = x)
        const float p = cs.total();
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
// captured graph of four kernels): every thread adds its rank's parts into the rank's BLOCK sums of the canonical order
// (activation_kernel; a rank of 2 / 4 / 8 equal shards holds 4 / 2 / 1 of the 8 blocks), stores them -- naturally aligned
// 8-byte words {value, epoch}, system scope -- into its slot of every OTHER rank's gather buffer (peer stores over xGMI:
// T K 8 bytes per block, peer and iteration; 64-256 KB at the headline shape), polls its own buffer until the other
// ranks' words carry this epoch, and adds all blocks in rank and block order: the same bits of r on every rank, and -- with
// equal shards on whole blocks -- the same bits as the collective path and as ONE GPU.  (Shards that do not hold whole
// blocks send one sum per rank: then r agrees with the other paths to rounding.)  The buffers alternate with the epoch's parity -- a rank can be one
// iteration ahead of the slowest reader of its stores, never two: it cannot finish epoch e + 1 before every rank has
// stored e + 1, which a rank does only after it has read epoch e.  The epoch is counted on the device, one word per
// workgroup (read at the start, advanced at the end by the workgroup itself), so a replayed graph needs no new arguments.
// A wait gives up after `timeout` ticks of the 100 MHz clock and records it in ctrl[0]; the host looks at its next
// synchronisation.  loopback: one GPU plays all `world` ranks (own buffer, the phantom ranks' sums are zeros).
struct ActXchgArgs {
    unsigned long long* gath[OIVA_XCHG_MAX_RANKS];      // every rank's [2][world][T * K] words, as mapped here
    int rank, world, loopback;                          // loopback 2: the phantom ranks never store (test hook: a rank that does not deliver)
    int nblk_own, nblk_peer, bs;                        // block sums this rank forms (of bs parts; the last may be short) / words per other rank's slot (<= 8 each)
    unsigned* epochs;                                   // [gridDim.y][gridDim.x]
    unsigned* ctrl;
    long long timeout;
};
template <int BS>      // parts per block at compile time (1, 2, 4, 8: every block's loads in flight together), 0: any
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
        // this rank's block sums (canonical order, see activation_kernel): nblk_own blocks of nparts / nblk_own parts.
        // The sums are the head of the chain sums -> stores to the peers -> their words arrive, so ALL loads are issued
        // before the first addition (round 4 made one round trip per block: 4 at two equal shards of the headline shape).
        float pb[kCanonBlocks];
        if constexpr (BS > 0) {
            float v[kCanonBlocks][BS];
#pragma unroll
            for (int b = 0; b < kCanonBlocks; ++b)
#pragma unroll
                for (int u = 0; u < BS; ++u) v[b][u] = (b < a.nblk_own && b * BS + u < nparts) ? parts[(size_t)(b * BS + u) * n + e] : 0.f;
#pragma unroll
            for (int b = 0; b < kCanonBlocks; ++b) {
                pb[b] = 0.f;
#pragma unroll
                for (int u = 0; u < BS; ++u) pb[b] += v[b][u];
            }
        } else {
            const int bs = a.bs;
#pragma unroll
            for (int b = 0; b < kCanonBlocks; ++b) {
                pb[b] = 0.f;
                if (b < a.nblk_own) {
                    const int b1 = min(nparts, (b + 1) * bs);
                    for (int i = b * bs; i < b1; i += 16) {
                        float v[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) v[u] = i + u < b1 ? parts[(size_t)(i + u) * n + e] : 0.f;
#pragma unroll
                        for (int u = 0; u < 16; ++u) pb[b] += v[u];
                    }
                }
            }
        }
        // slot of rank q in a buffer: [nblk_peer][T * K] words
        const size_t slot = (size_t)a.nblk_peer * n;
        const size_t base = (size_t)par * a.world * slot + e;
        for (int q = 0; q < a.world && a.loopback != 2; ++q) {
            if (q == a.rank) continue;
            unsigned long long* dst = a.loopback ? a.gath[a.rank] + base + (size_t)q * slot : a.gath[q] + base + (size_t)a.rank * slot;
#pragma unroll
            for (int b = 0; b < kCanonBlocks; ++b) {
                if (b < a.nblk_peer) {
                    const unsigned long long w = a.loopback ? 0ull : (unsigned long long)__float_as_uint(pb[b]);
                    __hip_atomic_store(dst + (size_t)b * n, w | ((unsigned long long)epoch << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        // Poll: the words of ALL other ranks requested together (<= 8: 7 x 1, 3 x 2 or 1 x 4 blocks at 8 / 4 / 2 equal shards),
        // as buffer loads with the system-scope cache bits -- an atomic load would be followed by s_waitcnt vmcnt(0), i.e. one
        // round trip to the (uncached, fine-grained) buffer per word instead of one per pass.
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(a.gath[a.rank], 0, (int)((size_t)2 * a.world * slot * 8), 0x00020000);
        const int npeer = a.world - 1;
        float vq[kCanonBlocks];
        const long long t0 = wall_clock64();
        for (unsigned spins = 1;; ++spins) {
            unsigned long long x[kCanonBlocks];
#pragma unroll
            for (int w = 0; w < kCanonBlocks; ++w) {
                // word w of the pass: block w % nblk_peer of the (w / nblk_peer)-th OTHER rank
                const int pi = w / a.nblk_peer, b = w - pi * a.nblk_peer;
                const int q = pi < a.rank ? pi : pi + 1;
                const bool live = pi < npeer;
                const size_t off = live ? (base + (size_t)q * slot + (size_t)b * n) * 8 : base * 8;
                const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off, 0, /*sc0 sc1: system scope*/ 17);
                x[w] = (unsigned long long)v[0] | ((unsigned long long)v[1] << 32);
                if (!live) x[w] = (unsigned long long)epoch << 32;
            }
            bool ok = true;
#pragma unroll
            for (int w = 0; w < kCanonBlocks; ++w) {
                ok = ok && (unsigned)(x[w] >> 32) == epoch;
                vq[w] = __uint_as_float((unsigned)x[w]);
            }
            if (ok) break;
            if ((spins & 15u) == 0u && (wall_clock64() - t0 > a.timeout || __hip_atomic_load(a.ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                atomicCAS(a.ctrl, 0u, 1u + (unsigned)blockIdx.x);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        // all blocks in rank order, block order (own blocks from registers)
        float tot = 0.f;
        for (int q = 0; q < a.world; ++q) {
            if (q == a.rank) {
#pragma unroll
                for (int b = 0; b < kCanonBlocks; ++b) tot += b < a.nblk_own ? pb[b] : 0.f;
            } else {
                const int pi = q < a.rank ? q : q - 1;
#pragma unroll
                for (int w = 0; w < kCanonBlocks; ++w) tot += (w >= pi * a.nblk_peer && w < (pi + 1) * a.nblk_peer) ? vq[w] : 0.f;
            }
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
    const dim3 grid((unsigned)rsum_blocks(T), (unsigned)K);
    const float inv = 1.f / (float)F_total;
    if (nparts <= 8)
        hipLaunchKernelGGL(activation_kernel<8>, grid, dim3(kBlock), 0, s, parts, nparts, R, T, K, model, inv);
    else if (nparts <= 16)
        hipLaunchKernelGGL(activation_kernel<16>, grid, dim3(kBlock), 0, s, parts, nparts, R, T, K, model, inv);
    else
        hipLaunchKernelGGL(activation_kernel<32>, grid, dim3(kBlock), 0, s, parts, nparts, R, T, K, model, inv);
    return hipGetLastError();
}

hipError_t launch_activation_xchg(hipStream_t s, const float* parts, int nparts, char* const* gath, int rank, int world, int loopback,
                                  int nblk_own, int nblk_peer, unsigned* epochs, unsigned* ctrl, long long timeout_ticks, float* R, int T, int K,
                                  int model, int F_total) {
    ActXchgArgs a;
    a.nblk_own = nblk_own;
    a.nblk_peer = nblk_peer;
    a.bs = (nparts + nblk_own - 1) / nblk_own;
    for (int r = 0; r < OIVA_XCHG_MAX_RANKS; ++r) a.gath[r] = r < world ? reinterpret_cast<unsigned long long*>(gath[r]) : nullptr;
    a.rank = rank;
    a.world = world;
    a.loopback = loopback;
    a.epochs = epochs;
    a.ctrl = ctrl;
    a.timeout = timeout_ticks;
    const dim3 grid((unsigned)rsum_blocks(T), (unsigned)K);
    const float inv = 1.f / (float)F_total;
    switch (a.bs) {
        case 1: hipLaunchKernelGGL(activation_xchg_kernel<1>, grid, dim3(kBlock), 0, s, parts, nparts, a, R, T, K, model, inv); break;
        case 2: hipLaunchKernelGGL(activation_xchg_kernel<2>, grid, dim3(kBlock), 0, s, parts, nparts, a, R, T, K, model, inv); break;
        case 4: hipLaunchKernelGGL(activation_xchg_kernel<4>, grid, dim3(kBlock), 0, s, parts, nparts, a, R, T, K, model, inv); break;
        case 8: hipLaunchKernelGGL(activation_xchg_kernel<8>, grid, dim3(kBlock), 0, s, parts, nparts, a, R, T, K, model, inv); break;
        default: hipLaunchKernelGGL(activation_xchg_kernel<0>, grid, dim3(kBlock), 0, s, parts, nparts, a, R, T, K, model, inv);
    }
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
