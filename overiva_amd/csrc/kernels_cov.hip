// Weighted spatial covariance pass, vector-ALU form: 1..8 channels in float32 (the dominant kernel of the
// iteration at the headline shape) and 1..7 channels in the float64 accumulation mode (8 channels in float64:
// kernels_cov_pair64.hip; 9..16 channels: kernels_cov_quad.hip, kernels_cov_mfma.hip).
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

#include <cstdint>
#include <cstdlib>

#include "oiva_device.h"
#include "cov_arith.h"

namespace oiva {
namespace {

constexpr int kCovUnroll = 2;            // frame steps in flight per wave
constexpr int kChunk = 16;               // accumulators combined per LDS round
constexpr int kLdsStride = kBlock + 1;   // +1: conflict-free transposed read

// Combine the 16 frame phases of a workgroup (tid = q*16 + b) in rounds of kChunk accumulators through
// LDS and store one packed partial per (frame split, bin, source):
//   write lds[a][tid]; thread (bb = tid/16, aa = tid%16) sums lds[aa][qq*16 + bb] over qq (fixed order).
// The sum over the phases and the stored partial are float64 whatever the accumulator type: a float32 lane chain
// covers T / (16 nsplit) frames, and adding 16 (then nsplit) of them in float32 would cost as much accuracy again as
// the chains themselves (measured on the reference's fixtures: 1.5-4 of its complex64 floors -> below one).
template <int M, int KC, typename ACC, typename At>
__device__ __forceinline__ void reduce_and_store_at(At&& at, ACC* lds, double* __restrict__ Vpart, int F, int K, int k0) {
    constexpr int NA = M * M;
    constexpr int NACC = NA * KC;
    const int tid = threadIdx.x;
    const int bb = tid >> 4, aa = tid & 15;
    const int fo = blockIdx.x * kBinsPerWave + bb;
    double* out = Vpart + (((size_t)blockIdx.y * F + fo) * K + k0) * NA;
#pragma unroll
    for (int r0 = 0; r0 < NACC; r0 += kChunk) {
        __syncthreads();
#pragma unroll
        for (int a = 0; a < kChunk; ++a) {
            if (r0 + a < NACC) lds[a * kLdsStride + tid] = at(r0 + a);      // accumulator kk * NA + a, compile-time index
        }
        __syncthreads();
        double s = 0.;
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) s += (double)lds[aa * kLdsStride + qq * 16 + bb];
        const int e = r0 + aa;          // accumulator index = kk*NA + a
        const int kk = e / NA;          // constant-folded per round when NA % 16 == 0
        if (e < NACC && fo < F && k0 + kk < K) out[e] = s;
    }
}

template <int M, int KC, typename ACC>
__device__ __forceinline__ void reduce_and_store(const ACC (&acc)[KC][M * M], ACC* lds, double* __restrict__ Vpart,
                                                 int F, int K, int k0) {
    reduce_and_store_at<M, KC, ACC>([&](int e) { return acc[e / (M * M)][e % (M * M)]; }, lds, Vpart, F, K, k0);
}

template <int M, int KC, bool UNIT, typename ACC>
__global__ __launch_bounds__(kBlock) void cov_kernel(const float2* __restrict__ X, const float* __restrict__ R,
                                                     float* __restrict__ wscale, int model, int raw,
                                                     double* __restrict__ Vpart, int T, int F, int K, int tc) {
    constexpr int NA = M * M;
    __shared__ ACC lds[kChunk * kLdsStride];

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

    // scale normalisation of the activations (overiva.py:158-159): 1/gamma for this pass's sources
    double ginv[KC];
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) ginv[kk] = 1.;
    if constexpr (!UNIT) {
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            const int k = k0 + kk;
            if (raw & 1) continue;          // test hook: R holds the final reciprocal weights
            const double gamma = gamma_of(R, T, K, k < K ? k : K - 1);
            ginv[kk] = 1. / gamma;
            if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && k < K && wscale != nullptr)
                wscale[k] = model == OIVA_MODEL_LAPLACE ? (float)gamma : (float)sqrt(gamma);   // overiva.py:163 / :167
        }
    }

    ACC acc[KC][NA];
#pragma unroll
    for (int kk = 0; kk < KC; ++kk)
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[kk][a] = 0;

    // kCovUnroll steps are loaded before any of them is consumed, so a wave keeps kCovUnroll * 64 * M * 8
    // bytes in flight.  Frames past the end of the split are clamped to a legal address and weighted by 0.
    const size_t frame_stride = (size_t)F * M;
    const float2* pbase = X + (size_t)fc * M;
    for (int i = 0; i < nsteps; i += kCovUnroll) {
        float xr[kCovUnroll][M], xi[kCovUnroll][M], rv[kCovUnroll][KC];
#pragma unroll
        for (int u = 0; u < kCovUnroll; ++u) {
            const int t = t_begin + q + 16 * (i + u);
            const int tcl = t < t_end ? t : T - 1;
            if constexpr (!UNIT) {
#pragma unroll
                for (int kk = 0; kk < KC; ++kk) {
                    const int k = k0 + kk;
                    rv[u][kk] = R[(size_t)tcl * K + (k < K ? k : K - 1)];
                }
            }
            load_x<M>(pbase + (size_t)tcl * frame_stride, xr[u], xi[u]);
        }
#pragma unroll
        for (int u = 0; u < kCovUnroll; ++u) {
            const int t = t_begin + q + 16 * (i + u);
            const bool live = t < t_end;
            ACC w[KC], ar[M], ai[M];
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) {
                const bool on = live && (UNIT || k0 + kk < K);
                if constexpr (UNIT) {
                    w[kk] = on ? ACC(1) : ACC(0);
                } else if constexpr (sizeof(ACC) == 8) {
                    double rn = (double)rv[u][kk] * ginv[kk];
                    rn = rn < (double)kEpsR ? (double)kEpsR : rn;
                    w[kk] = on ? 1. / rn : 0.;
                } else {
                    w[kk] = on ? activation_weight(rv[u][kk], (float)ginv[kk]) : 0.f;
                }
            }
#pragma unroll
            for (int c = 0; c < M; ++c) {
                ar[c] = (ACC)xr[u][c];
                ai[c] = (ACC)xi[u][c];
            }
            accumulate<M, KC, UNIT, ACC>(acc, ar, ai, w);
        }
    }

    reduce_and_store<M, KC, ACC>(acc, lds, Vpart, F, K, k0);
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant of the weighted pass (even M).  Same lane geometry and arithmetic as cov_kernel, but
// X goes HBM -> LDS with global_load_lds (no staging registers) into a private 4-stage ring per wave:
// three steps (3 * 64 lanes * M*8 bytes) stay in flight while the fourth is consumed, which is what a
// kernel limited to two waves per SIMD by its 128 accumulators needs to cover HBM latency.  Each lane
// reads back exactly the bytes it requested (piece j of its own M-vector lands at
// stage_base + j*1024 + lane*16), so the ring needs no swizzle and no workgroup barrier; the only
// ordering is the wave's own vmcnt.  The LDS reads and both counted waits sit in one asm block:
// hipcc otherwise drains the whole DMA queue (vmcnt(0)) in front of any LDS read it can see.
// The weights r[t,k] of the wave's four frame phases are wave-uniform and come through the scalar cache.
// ---------------------------------------------------------------------------------------------
constexpr int kDmaStages = 4;

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

template <int PIECES>
__device__ __forceinline__ void ring_read(unsigned addr, float4 (&v)[PIECES]);

template <>
__device__ __forceinline__ void ring_read<4>(unsigned addr, float4 (&v)[4]) {
    asm volatile(
        "s_waitcnt vmcnt(12)\n\t"
        "ds_read_b128 %0, %4\n\t"
        "ds_read_b128 %1, %4 offset:1024\n\t"
        "ds_read_b128 %2, %4 offset:2048\n\t"
        "ds_read_b128 %3, %4 offset:3072\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
        : "v"(addr)
        : "memory");
}
template <>
__device__ __forceinline__ void ring_read<2>(unsigned addr, float4 (&v)[2]) {
    asm volatile(
        "s_waitcnt vmcnt(6)\n\t"
        "ds_read_b128 %0, %2\n\t"
        "ds_read_b128 %1, %2 offset:1024\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1])
        : "v"(addr)
        : "memory");
}

// (tools/build_variant.py ... -DOIVA_COVDMA_TRACE: 100 MHz wall-clock stamps of every workgroup, read by tools/r6/covdma_trace.py)
#ifdef OIVA_COVDMA_TRACE
__device__ long long g_cd_trace[8 * 2048];
#define CD_STAMP(slot)                                                                                                            \
    do {                                                                                                                          \
        if (threadIdx.x == 0 && blockIdx.z == 0) g_cd_trace[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (slot)] = wall_clock64(); \
    } while (0)
#else
#define CD_STAMP(slot) \
    do {               \
    } while (0)
#endif

template <int M, int KC>
__global__ __launch_bounds__(kBlock, 2) void cov_dma_kernel(const float2* __restrict__ X, const float* __restrict__ R,
                                                            float* __restrict__ wscale, int model, int raw,
                                                            double* __restrict__ Vpart, int T, int F, int K, int tc) {
    constexpr int NA = M * M;
    constexpr int PIECES = M / 2;                       // 16-byte pieces of one M-vector
    constexpr int STAGE = PIECES * 64;                  // float4 per stage per wave
    static_assert(M % 2 == 0 && (PIECES == 2 || PIECES == 4), "LDS-DMA path: M in {4, 8}");
    __shared__ float4 ring[kWaves * kDmaStages * STAGE];
    static_assert(sizeof(float4) * kWaves * kDmaStages * STAGE >= sizeof(float) * kChunk * kLdsStride,
                  "reduction scratch aliases the ring");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = lane & (kBinsPerWave - 1);
    const int ql = lane >> 4;                           // phase inside the wave
    const int q = wave * kPhasesPerWave + ql;           // 0..15
    const int f = blockIdx.x * kBinsPerWave + b;
    const int fc = f < F ? f : F - 1;
    const int k0 = blockIdx.z * KC;
    const int t_begin = blockIdx.y * tc;
    const int t_end = min(T, t_begin + tc);
    const int nsteps = (t_end - t_begin + 15) >> 4;

    constexpr bool kPacked = KC == 2;            // two sources: hand-packed arithmetic on (re, im) pairs
    float acc[kPacked ? 1 : KC][kPacked ? 1 : NA];
    PkAcc2<M> pacc;
    if constexpr (kPacked) {
        pacc.clear();
    } else {
#pragma unroll
        for (int kk = 0; kk < KC; ++kk)
#pragma unroll
            for (int a = 0; a < NA; ++a) acc[kk][a] = 0.f;
    }

    float ginv[KC];
    float4* wring = ring + wave * kDmaStages * STAGE;                       // wave-uniform
    const unsigned rd_base = (unsigned)(uintptr_t)(wring) + lane * 16;      // LDS byte address of this lane's slot
    const size_t frame_stride = (size_t)F * M;
    const float2* pbase = X + (size_t)fc * M;

    // issue the M/2 DMA requests of step i into stage s; steps past the end re-request the last frame
    // (legal address, never consumed) so that every step adds the same count to vmcnt
    auto issue = [&](int i, int s) {
        const int t = t_begin + q + 16 * i;
        const int tcl = (i < nsteps && t < t_end) ? t : T - 1;
        const float2* src = pbase + (size_t)tcl * frame_stride;
#pragma unroll
        for (int j = 0; j < PIECES; ++j)
            __builtin_amdgcn_global_load_lds((gvoid_t*)(src + 2 * j), (lvoid_t*)(wring + s * STAGE + j * 64), 16, 0, 0);
    };
    // lane masks selecting the lane's phase among the wave's four frames (arithmetic select: a ?: chain on
    // a lane-varying condition gets compiled into branches with the scalar loads inside them)
    const float m0 = ql == 0 ? 1.f : 0.f, m1 = ql == 1 ? 1.f : 0.f, m2 = ql == 2 ? 1.f : 0.f, m3 = ql == 3 ? 1.f : 0.f;
    auto consume = [&](int i, int s) {
        // weights of the wave's four consecutive frames: uniform addresses -> scalar loads.  R is allocated
        // with kPhasesPerWave zeroed rows of padding after frame T-1; waves whose four frames lie wholly past
        // the end of the tensor (last step of the last split) read those rows, never anything beyond them.
        const int tw = min(t_begin + wave * kPhasesPerWave + 16 * i, T);
        const float* rp = R + (size_t)tw * K;
        float w[KC];
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            const int k = k0 + kk < K ? k0 + kk : K - 1;
            const float r = m0 * rp[k] + m1 * rp[K + k] + m2 * rp[2 * K + k] + m3 * rp[3 * K + k];
            const float live = (t_begin + wave * kPhasesPerWave + 16 * i + ql < t_end && k0 + kk < K) ? 1.f : 0.f;
            w[kk] = activation_weight(r, ginv[kk]) * live;
        }
        float4 v[PIECES];
        ring_read<PIECES>(rd_base + s * STAGE * 16, v);
#if defined(OIVA_COVDMA_ABLATE) && (OIVA_COVDMA_ABLATE & 1)      // variant build (tools/build_variant.py): no arithmetic
        if constexpr (kPacked) {
            if (v[0].x == 12345.f) pacc.add(reinterpret_cast<const v2f(&)[M]>(v), v2f{w[0], w[1]});
            return;
        }
#endif
        if constexpr (kPacked) {
            v2f x[M];
#pragma unroll
            for (int j = 0; j < PIECES; ++j) {
                x[2 * j] = v2f{v[j].x, v[j].y};
                x[2 * j + 1] = v2f{v[j].z, v[j].w};
            }
            pacc.add(x, v2f{w[0], w[1]});
        } else {
            float xr[M], xi[M];
#pragma unroll
            for (int j = 0; j < PIECES; ++j) {
                xr[2 * j] = v[j].x;
                xi[2 * j] = v[j].y;
                xr[2 * j + 1] = v[j].z;
                xi[2 * j + 1] = v[j].w;
            }
            accumulate<M, KC, false, float>(acc, xr, xi, w);
        }
    };

    CD_STAMP(0);
    issue(0, 0);
    issue(1, 1);
    issue(2, 2);
    // scale normalisation of the activations (overiva.py:158-159) while the first three steps are on their way
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
        const int k = k0 + kk;
        ginv[kk] = 1.f;
        if (raw & 1) continue;              // test hook: R holds the final reciprocal weights
#if defined(OIVA_COVDMA_ABLATE) && (OIVA_COVDMA_ABLATE & 4)      // variant build: no sum over the frames in the prologue
        const float gamma = R[k < K ? k : K - 1] + 1.f;
#else
        const float gamma = (float)gamma_of(R, T, K, k < K ? k : K - 1);
#endif
        ginv[kk] = 1.f / gamma;
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && k < K && wscale != nullptr)
            wscale[k] = model == OIVA_MODEL_LAPLACE ? gamma : sqrtf(gamma);   // overiva.py:163 / :167
    }

    CD_STAMP(1);
    int i = 0;
    for (; i + 4 <= nsteps; i += 4) {       // stage indices are compile-time constants in the unrolled body
        issue(i + 3, 3); consume(i, 0);
        issue(i + 4, 0); consume(i + 1, 1);
        issue(i + 5, 1); consume(i + 2, 2);
        issue(i + 6, 2); consume(i + 3, 3);
    }
    // 0..3 remaining steps; the stage sequence restarts at 0 because i is a multiple of 4
    if (i < nsteps) { issue(i + 3, 3); consume(i, 0); }
    if (i + 1 < nsteps) { issue(i + 4, 0); consume(i + 1, 1); }
    if (i + 2 < nsteps) { issue(i + 5, 1); consume(i + 2, 2); }
    CD_STAMP(2);
    // drain the DMA queue before the ring is reused as reduction scratch
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CD_STAMP(3);
#if defined(OIVA_COVDMA_ABLATE) && (OIVA_COVDMA_ABLATE & 2)      // variant build: no epilogue
    if constexpr (kPacked) {
        if (pacc.at(0) == 12345.f) Vpart[0] = 1.;
        return;
    }
#endif
    // (adding the 4 phases of a wave in registers first -- v_permlane16/32_swap, a quarter of the LDS traffic, 4
    // barriers instead of 16 -- was measured slower: 102-104 us against 96-98)
    if constexpr (kPacked)
        reduce_and_store_at<M, KC, float>([&](int e) { return pacc.at(e); }, reinterpret_cast<float*>(ring), Vpart, F, K, k0);
    else
        reduce_and_store<M, KC, float>(acc, reinterpret_cast<float*>(ring), Vpart, F, K, k0);
    CD_STAMP(4);
}

template <typename ACC>
using CovKernel = void (*)(const float2*, const float*, float*, int, int, double*, int, int, int, int);   // ACC = accumulator type

// (M, KC, unit weights) -> kernel instantiation, handed to fn together with its KC.  Register budget:
// KC * M^2 accumulators of ACC must stay below ~144 registers.
template <int M, typename ACC, typename Fn>
hipError_t dispatch_kc(int kc, bool unit, Fn&& fn) {
    constexpr int kRegs = M * M * (int)(sizeof(ACC) / 4);
    if (unit) return kc == 1 ? fn((CovKernel<ACC>)cov_kernel<M, 1, true, ACC>, 1) : hipErrorInvalidValue;
    if constexpr ((M == 4 || M == 8) && sizeof(ACC) == 4) {       // LDS-DMA ring where available
        switch (kc) {
            case 1:
                return fn((CovKernel<ACC>)cov_dma_kernel<M, 1>, 1);
            case 2:
                if constexpr (kRegs * 2 <= 144) return fn((CovKernel<ACC>)cov_dma_kernel<M, 2>, 2);
                break;
            case 4:
                if constexpr (kRegs * 4 <= 144) return fn((CovKernel<ACC>)cov_dma_kernel<M, 4>, 4);
                break;
        }
        return hipErrorInvalidValue;
    }
    switch (kc) {
        case 1:
            return fn((CovKernel<ACC>)cov_kernel<M, 1, false, ACC>, 1);
        case 2:
            if constexpr (kRegs * 2 <= 144) return fn((CovKernel<ACC>)cov_kernel<M, 2, false, ACC>, 2);
            break;
        case 4:
            if constexpr (kRegs * 4 <= 144) return fn((CovKernel<ACC>)cov_kernel<M, 4, false, ACC>, 4);
            break;
        // (round 5) one source more than the budget of 144 accumulator registers allows where that saves a pass over X on a
        // short frame axis: 7 channels x 3 sources (147), 5 x 5 (125) -- see cov_sources_per_pass
        case 3:
            if constexpr (M == 7 && sizeof(ACC) == 4) return fn((CovKernel<ACC>)cov_kernel<M, 3, false, ACC>, 3);
            break;
        case 5:
            if constexpr (M == 5 && sizeof(ACC) == 4) return fn((CovKernel<ACC>)cov_kernel<M, 5, false, ACC>, 5);
            break;
    }
    return hipErrorInvalidValue;
}

template <typename ACC, typename Fn>
hipError_t dispatch_cov(int M, int kc, bool unit, Fn&& fn) {
    switch (M) {
        case 1: return dispatch_kc<1, ACC>(kc, unit, fn);
        case 2: return dispatch_kc<2, ACC>(kc, unit, fn);
        case 3: return dispatch_kc<3, ACC>(kc, unit, fn);
        case 4: return dispatch_kc<4, ACC>(kc, unit, fn);
        case 5: return dispatch_kc<5, ACC>(kc, unit, fn);
        case 6: return dispatch_kc<6, ACC>(kc, unit, fn);
        case 7: return dispatch_kc<7, ACC>(kc, unit, fn);
        case 8: return dispatch_kc<8, ACC>(kc, unit, fn);
    }
    return hipErrorInvalidValue;
}

}  // namespace

bool cov_supported(int M) { return M >= 1 && M <= OIVA_MAX_CHANNELS; }

// sources handled per pass over X
int cov_sources_per_pass(int M, int K, bool f64, bool short_axis) {
    if (f64 && cov_pair64_supported(M)) return cov_pair64_sources_per_pass(K);
    if (!f64 && cov_pair32_supported(M, K)) return cov_pair32_sources_per_pass();
    const int regs = M * M * (f64 ? 2 : 1);   // as many as fit the accumulator budget (KC * M^2 <= 144 registers)
    int kc = 1;
    if (K >= 2 && regs * 2 <= 144) kc = 2;
    if (K >= 3 && regs * 4 <= 144) kc = 4;
    // 7 channels / 3+ sources and 5 / 5: a third / fifth source per pass (147 / 125 accumulators: two waves per SIMD instead of
    // three, a pass of 7 channels 100 us instead of 75 at 2048 x 4000) where that saves a whole pass over X.  Measured (covariance
    // pass, iteration): 2048 x 4000 x 7 / 3 162 -> 105 us (239 -> 198), 5 / 5 121 -> 93 (203 -> 181), 2049 x 235 x 7 / 3 22.0 ->
    // 16.5 (51.8 -> 47.6), 7 / 7 34.0 -> 27.2 (73.8 -> 66.6); not 7 / 4 (two passes either way: 21.8 -> 23.4) and 7 / 7 only on
    // a short frame axis (2048 x 4000: three passes of three 294 us, four of two 283).  $OIVA_COV_KC_WIDE=0: off.
    static const bool wide = [] { const char* v = std::getenv("OIVA_COV_KC_WIDE"); return !(v && v[0] == '0'); }();
    if (!f64 && wide) {
        if (M == 7 && (K == 3 || K == 5 || K == 6 || (K == 7 && short_axis))) kc = 3;
        if (M == 5 && K >= 5) kc = 5;
    }
    return kc;
}

// one zero channel behind every (frame, bin)'s M: the even channel pitch the vector-ALU kernels of 10..16 channels read
__global__ __launch_bounds__(kBlock) void pad_channels_kernel(const float2* __restrict__ X, float2* __restrict__ Xpad, long long n, int M) {
    const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;          // element of the padded tensor
    if (e >= n) return;
    const long long tf = e / (M + 1);
    const int c = (int)(e - tf * (M + 1));
    Xpad[e] = c < M ? X[tf * M + c] : make_float2(0.f, 0.f);
}

hipError_t launch_pad_channels(hipStream_t s, const float2* X, float2* Xpad, long long n_tf, int M) {
    const long long n = n_tf * (M + 1);
    pad_channels_kernel<<<dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s>>>(X, Xpad, n, M);
    return hipGetLastError();
}

hipError_t launch_cov(hipStream_t s, const float2* X, const float2* Xpad, const float* R, float* Wt, float* wscale, int model, int raw,
                      void* Vpart, bool f64, int T, int F, int M, int K, const CovGeom& g) {
    if (g.pad && Xpad == nullptr) return hipErrorInvalidValue;
    const float2* Xv = g.pad ? Xpad : X;        // what the vector-ALU kernels of 10..16 channels read, at a pitch of Mp channels
    const int Mp = g.pad ? M + 1 : M;
    if (M > 8 && g.half16 && !f64) return launch_cov_half16(s, Xv, R, Wt, wscale, model, raw, static_cast<double*>(Vpart), T, F, Mp, M, K, g);
    if (M > 8 && g.half16 && f64 && R != nullptr)
        return launch_cov_half16_f64(s, Xv, R, Wt, wscale, model, raw, static_cast<double*>(Vpart), T, F, Mp, M, K, g);
    if (M > 8 && g.quad && !f64) return launch_cov_quad(s, Xv, R, Wt, wscale, model, raw, static_cast<double*>(Vpart), T, F, Mp, M, K, g);
    if (M > 8) return launch_cov_mfma(s, X, R, Wt, wscale, model, raw, Vpart, f64, T, F, M, K, g.nsplit, g.tc);
    if (!f64 && g.pair32) return launch_cov_pair32(s, X, R, Wt, wscale, model, raw, static_cast<double*>(Vpart), T, F, M, K, g);
    if (f64 && cov_pair64_supported(M)) return launch_cov_pair64(s, X, R, Wt, wscale, model, raw, static_cast<double*>(Vpart), T, F, M, K, g);
    const int kc = R == nullptr ? 1 : g.kc;
    if (f64)
        return dispatch_cov<double>(M, kc, R == nullptr, [&](CovKernel<double> kern, int KC) {
            kern<<<dim3(g.nbg, g.nsplit, (K + KC - 1) / KC), dim3(kBlock), 0, s>>>(X, R, wscale, model, raw,
                                                                                 static_cast<double*>(Vpart), T, F, K, g.tc);
            return hipGetLastError();
        });
    return dispatch_cov<float>(M, kc, R == nullptr, [&](CovKernel<float> kern, int KC) {
        return launch_dominant(kern, dim3(g.nbg, g.nsplit, (K + KC - 1) / KC), dim3(kBlock), 0, s, X, R, wscale, model, raw,
                               static_cast<double*>(Vpart), T, F, K, g.tc);
    });
}

// workgroups of this instantiation that one CU holds at once (registers / LDS limited)
hipError_t cov_blocks_per_cu(int M, int kc, bool f64, int* n) {
    if ((f64 && cov_pair64_supported(M)) || (!f64 && M == 8 && kc == 4)) {
        *n = 2;
        return hipSuccess;
    }
    if (f64)
        return dispatch_cov<double>(M, kc, false, [&](CovKernel<double> kern, int) {
            return hipOccupancyMaxActiveBlocksPerMultiprocessor(n, kern, kBlock, 0);
        });
    return dispatch_cov<float>(M, kc, false, [&](CovKernel<float> kern, int) {
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(n, kern, kBlock, 0);
    });
}

#ifdef OIVA_COVDMA_TRACE
extern "C" int oiva_debug_covdma_trace(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cd_trace), sizeof(g_cd_trace)); }
#endif

}  // namespace oiva
