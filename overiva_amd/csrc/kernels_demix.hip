// Kernels that apply the demixing vectors to X: the per-iteration source-power pass and the epilogue.
//
//   power : p[t,k] = sum_f |w_{f,k}^H x_{t,f}|^2      reference overiva.py:140 + the norms of :153/:155
//           (Y itself is never stored inside the loop: the reference's Y /= gamma at :162/:166 is dead)
//   stats : per-bin sums for projection back          reference overiva.py:197-198 (pyroomacoustics formula)
//   write : Y[t,f,k] = w_{f,k}^H x_{t,f} (* conj z)   reference overiva.py:192-199
#include "oiva_device.h"
#include "demix_arith.h"

namespace oiva {
namespace {

// ---------------------------------------------------------------------------------------------
// power: block = 4 waves x 16 bins = 64 bins, 4 frame phases per wave, frames [t_begin, t_begin+tcp)
// ---------------------------------------------------------------------------------------------
constexpr int kPowUnroll = 2;

template <int M, int KP>
__global__ __launch_bounds__(kBlock) void power_kernel(const float2* __restrict__ X, const float2* __restrict__ What,
                                                       float* __restrict__ Ppart, int T, int F, int K, int tcp) {
    extern __shared__ __attribute__((aligned(16))) float sp[];  // [kWaves][tcp][KP]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int b = lane & 15;
    const int q = lane >> 4;
    const int f = (blockIdx.x * kWaves + wave) * kBinsPerWave + b;
    const bool fvalid = f < F;
    const int fc = fvalid ? f : F - 1;
    const int k0 = blockIdx.z * KP;
    const int t_begin = blockIdx.y * tcp;
    const int t_end = min(T, t_begin + tcp);
    const int len = t_end - t_begin;
    const int nsteps = (len + 3) >> 2;

    float wr[KP][M], wi[KP][M];
    load_wconj<M, KP>(What, fc, k0, K, wr, wi);

    // kPowUnroll steps are loaded before any of them is consumed: a wave keeps kPowUnroll * 64 * M * 8
    // bytes in flight, which is what hides HBM latency here (more resident waves only thrash the L1,
    // because a lane's M*8 bytes arrive through M/2 separate dwordx4 requests to the same lines).
    const size_t frame_stride = (size_t)F * M;
    const float2* pbase = X + (size_t)fc * M;
    for (int i = 0; i < nsteps; i += kPowUnroll) {
        float xr[kPowUnroll][M], xi[kPowUnroll][M];
#pragma unroll
        for (int u = 0; u < kPowUnroll; ++u) {
            const int tl = 4 * (i + u) + q;
            const int t = tl < len ? t_begin + tl : T - 1;      // clamped: legal address, result unused
            load_x<M>(pbase + (size_t)t * frame_stride, xr[u], xi[u]);
        }
#pragma unroll
        for (int u = 0; u < kPowUnroll; ++u) {
            const int tl = 4 * (i + u) + q;
            const bool live = tl < len;
#pragma unroll
            for (int kk = 0; kk < KP; ++kk) {
                float yr, yi;
                demix_one<M>(wr[kk], wi[kk], xr[u], xi[u], yr, yi);
                float pw = fmaf(yr, yr, yi * yi);
                pw = fvalid ? pw : 0.f;
                pw = row16_sum(pw);
                if (b == 0 && live) sp[(wave * tcp + tl) * KP + kk] = pw;
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < len * KP; e += kBlock) {
        const int tl = e / KP, kk = e - tl * KP;
        float s = sp[(0 * tcp + tl) * KP + kk];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) s += sp[(w * tcp + tl) * KP + kk];
        if (k0 + kk < K) Ppart[((size_t)blockIdx.x * T + t_begin + tl) * K + k0 + kk] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// stats: same lane geometry as the covariance pass (16 bins per block, 16 frame phases);
//        per (bin, source): num = sum_t conj(x_0) y,  den = sum_t |y|^2
// ---------------------------------------------------------------------------------------------
template <int M, int KP>
__global__ __launch_bounds__(kBlock) void stats_kernel(const float2* __restrict__ X, const float2* __restrict__ What,
                                                       float* __restrict__ Spart, int T, int F, int K, int tc) {
    __shared__ float lds[3 * KP * (kBlock + 1)];
    const int tid = threadIdx.x;
    const int b = tid & 15;
    const int q = tid >> 4;
    const int f = blockIdx.x * kBinsPerWave + b;
    const int fc = f < F ? f : F - 1;
    const int k0 = blockIdx.z * KP;
    const int t_begin = blockIdx.y * tc;
    const int t_end = min(T, t_begin + tc);
    const int nsteps = (t_end - t_begin + 15) >> 4;

    float wr[KP][M], wi[KP][M];
    load_wconj<M, KP>(What, fc, k0, K, wr, wi);
    float nr[KP], ni[KP], dn[KP];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) nr[kk] = ni[kk] = dn[kk] = 0.f;

    const size_t frame_stride = (size_t)F * M;
    const float2* px = X + ((size_t)(t_begin + q) * F + fc) * M;
    for (int i = 0; i < nsteps; ++i) {
        const int t = t_begin + q + 16 * i;
        if (t < t_end) {
            float xr[M], xi[M];
            load_x<M>(px, xr, xi);
#pragma unroll
            for (int kk = 0; kk < KP; ++kk) {
                float yr, yi;
                demix_one<M>(wr[kk], wi[kk], xr, xi, yr, yi);
                // conj(x0) * y
                nr[kk] = fmaf(xr[0], yr, fmaf(xi[0], yi, nr[kk]));
                ni[kk] = fmaf(xr[0], yi, fmaf(-xi[0], yr, ni[kk]));
                dn[kk] = fmaf(yr, yr, fmaf(yi, yi, dn[kk]));
            }
        }
        px += 16 * frame_stride;
    }
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
        lds[(3 * kk + 0) * (kBlock + 1) + tid] = nr[kk];
        lds[(3 * kk + 1) * (kBlock + 1) + tid] = ni[kk];
        lds[(3 * kk + 2) * (kBlock + 1) + tid] = dn[kk];
    }
    __syncthreads();
    // thread e -> (bin bb, value v): sum the 16 phases
    for (int e = tid; e < 16 * 3 * KP; e += kBlock) {
        const int bb = e / (3 * KP), v = e - bb * (3 * KP);
        float s = 0.f;
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) s += lds[v * (kBlock + 1) + qq * 16 + bb];
        const int fo = blockIdx.x * kBinsPerWave + bb;
        const int kk = v / 3, c = v - 3 * kk;
        if (fo < F && k0 + kk < K) Spart[(((size_t)blockIdx.y * F + fo) * K + k0 + kk) * 3 + c] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// write: Y (T, F, K) complex64; z from the stats partials when projecting back
// ---------------------------------------------------------------------------------------------
template <int M, int KP>
__global__ __launch_bounds__(kBlock) void write_kernel(const float2* __restrict__ X, const float2* __restrict__ What,
                                                       const float* __restrict__ Spart, int nsplit,
                                                       float2* __restrict__ Y, int T, int F, int K, int tc) {
    const int tid = threadIdx.x;
    const int b = tid & 15;
    const int q = tid >> 4;
    const int f = blockIdx.x * kBinsPerWave + b;
    if (f >= F) return;
    const int k0 = blockIdx.z * KP;
    const int t_begin = blockIdx.y * tc;
    const int t_end = min(T, t_begin + tc);

    float wr[KP][M], wi[KP][M];
    load_wconj<M, KP>(What, f, k0, K, wr, wi);
    // fold conj(z) into the demixing vector: y*conj(z) = (conj(z) w^H) x
    if (Spart != nullptr) {
#pragma unroll
        for (int kk = 0; kk < KP; ++kk) {
            if (k0 + kk < K) {
                double sr = 0., si = 0., sd = 0.;
                for (int s = 0; s < nsplit; ++s) {
                    const float* p = Spart + (((size_t)s * F + f) * K + k0 + kk) * 3;
                    sr += p[0];
                    si += p[1];
                    sd += p[2];
                }
                float zr = 1.f, zi = 0.f;
                if (sd > 0.) {
                    zr = (float)(sr / sd);
                    zi = (float)(si / sd);
                }
                // multiply conj(w) (held as wr + i wi) by conj(z) = zr - i zi
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const float a = wr[kk][m], c = wi[kk][m];
                    wr[kk][m] = a * zr + c * zi;
                    wi[kk][m] = c * zr - a * zi;
                }
            }
        }
    }
    const size_t frame_stride = (size_t)F * M;
    const float2* px = X + ((size_t)(t_begin + q) * F + f) * M;
    for (int t = t_begin + q; t < t_end; t += 16) {
        float xr[M], xi[M];
        load_x<M>(px, xr, xi);
#pragma unroll
        for (int kk = 0; kk < KP; ++kk) {
            float yr, yi;
            demix_one<M>(wr[kk], wi[kk], xr, xi, yr, yi);
            if (k0 + kk < K) Y[((size_t)t * F + f) * K + k0 + kk] = make_float2(yr, yi);
        }
        px += 16 * frame_stride;
    }
}

template <int M, int KP>
hipError_t launch_power_one(hipStream_t s, const float2* X, const float2* What, float* Ppart, int T, int F, int K,
                            const PowGeom& g) {
    dim3 grid(g.nb, g.nsplit, (K + KP - 1) / KP);
    const size_t shmem = (size_t)kWaves * g.tcp * KP * sizeof(float);
    hipLaunchKernelGGL((power_kernel<M, KP>), grid, dim3(kBlock), shmem, s, X, What, Ppart, T, F, K, g.tcp);
    return hipGetLastError();
}

template <int M, int KP>
hipError_t launch_stats_one(hipStream_t s, const float2* X, const float2* What, float* Spart, int T, int F, int K,
                            const CovGeom& g) {
    dim3 grid(g.nbg, g.nsplit, (K + KP - 1) / KP);
    hipLaunchKernelGGL((stats_kernel<M, KP>), grid, dim3(kBlock), 0, s, X, What, Spart, T, F, K, g.tc);
    return hipGetLastError();
}

template <int M, int KP>
hipError_t launch_write_one(hipStream_t s, const float2* X, const float2* What, const float* Spart, int nsplit,
                            float2* Y, int T, int F, int K) {
    const int tc = 256;
    dim3 grid((F + kBinsPerWave - 1) / kBinsPerWave, (T + tc - 1) / tc, (K + KP - 1) / KP);
    hipLaunchKernelGGL((write_kernel<M, KP>), grid, dim3(kBlock), 0, s, X, What, Spart, nsplit, Y, T, F, K, tc);
    return hipGetLastError();
}

// dispatch (M, KP) -> instantiation.  KP in {1, 2, 4}.
#define OIVA_DISPATCH_M(CALL)            \
    switch (M) {                         \
        case 1: CALL(1); break;          \
        case 2: CALL(2); break;          \
        case 3: CALL(3); break;          \
        case 4: CALL(4); break;          \
        case 5: CALL(5); break;          \
        case 6: CALL(6); break;          \
        case 7: CALL(7); break;          \
        case 8: CALL(8); break;          \
        case 9: CALL(9); break;          \
        case 10: CALL(10); break;        \
        case 11: CALL(11); break;        \
        case 12: CALL(12); break;        \
        case 13: CALL(13); break;        \
        case 14: CALL(14); break;        \
        case 15: CALL(15); break;        \
        case 16: CALL(16); break;        \
    }

}  // namespace

int pow_sources_per_pass(int M, int K) {
    // (round 5, measured and dropped: all of 5..8 sources in ONE pass, power_kernel<M, 8> -- 2049 x 235: 5 / 5 10.3 -> 11.7 us,
    //  6 / 6 11.9 -> 12.8, 7 / 7 13.5 -> 14.5, 8 / 8 17.8 -> 18.9; 2048 x 4000: 8 / 8 121 -> 129, 5 / 5 75 -> 76, only 8 / 5
    //  140 -> 111: eight demixing products per frame make the pass arithmetic-bound, two passes of four overlap)
    if (K >= 3) return 4;
    if (K >= 2) return 2;
    return 1;
}

hipError_t pow_blocks_per_cu(int M, int kp, int tcp, int* n) {
    const size_t shmem = (size_t)kWaves * tcp * kp * sizeof(float);
#define CALL(MM)                                                                                                   \
    if (kp == 1) return hipOccupancyMaxActiveBlocksPerMultiprocessor(n, power_kernel<MM, 1>, kBlock, shmem);       \
    if (kp == 2) return hipOccupancyMaxActiveBlocksPerMultiprocessor(n, power_kernel<MM, 2>, kBlock, shmem);       \
    if (kp == 4) return hipOccupancyMaxActiveBlocksPerMultiprocessor(n, power_kernel<MM, 4>, kBlock, shmem);
    OIVA_DISPATCH_M(CALL)
#undef CALL
    return hipErrorInvalidValue;
}

hipError_t launch_power(hipStream_t s, const float2* X, const float2* Xpad, const float2* What, float* Ppart, int T, int F, int M,
                        int K, const PowGeom& g) {
    // more than 4 sources would take several VALU passes over X (register budget): one MFMA pass instead
    // (measured at 16 channels: 16 sources 770 -> 325 us; 2 sources 191 us VALU vs 332 us MFMA); an odd channel count
    // reads the copy of X padded by one zero channel (16-byte loads instead of 8-byte ones)
    if (M > 8 && K > 4) return Xpad ? launch_power_mfma(s, Xpad, What, Ppart, T, F, M, M + 1, K) : launch_power_mfma(s, X, What, Ppart, T, F, M, M, K);
#define CALL(MM)                                                                                        \
    if (g.kp == 1) return launch_power_one<MM, 1>(s, X, What, Ppart, T, F, K, g);                       \
    if (g.kp == 2) return launch_power_one<MM, 2>(s, X, What, Ppart, T, F, K, g);                       \
    if (g.kp == 4) return launch_power_one<MM, 4>(s, X, What, Ppart, T, F, K, g);
    OIVA_DISPATCH_M(CALL)
#undef CALL
    return hipErrorInvalidValue;
}

hipError_t launch_demix_stats(hipStream_t s, const float2* X, const float2* What, float* Spart, int T, int F, int M,
                              int K, const CovGeom& g) {
#define CALL(MM) return launch_stats_one<MM, 2>(s, X, What, Spart, T, F, K, g);
    OIVA_DISPATCH_M(CALL)
#undef CALL
    return hipErrorInvalidValue;
}

hipError_t launch_demix_write(hipStream_t s, const float2* X, const float2* What, const float* Spart, int nsplit,
                              float2* Y, int T, int F, int M, int K) {
#define CALL(MM) return launch_write_one<MM, 2>(s, X, What, Spart, nsplit, Y, T, F, K);
    OIVA_DISPATCH_M(CALL)
#undef CALL
    return hipErrorInvalidValue;
}

}  // namespace oiva
