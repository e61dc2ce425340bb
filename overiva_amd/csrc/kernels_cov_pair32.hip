// Weighted spatial covariance pass for 8 channels and MORE THAN TWO sources, float32 packed arithmetic: FOUR sources per
// pass over X.
//
//   V_k[f] = sum_t rinv[t,k] * x_{t,f} x_{t,f}^H          reference overiva.py:179
//
// cov_dma_kernel (kernels_cov.hip) keeps the whole Hermitian half of a (bin, frame) in one lane: 64 accumulators per
// source, two sources per pass, so three or four sources (overiva_sim_config.json sweeps 1..4 targets) cost two passes over
// X and the determined 8 x 8 case four.  Here the half is split over the TWO lanes of a pair (pair_position, cov_arith.h;
// same split as the float64 kernel kernels_cov_pair64.hip): 32 sums per source and lane, four sources = 128 accumulators,
// one pass.  Per lane and frame 5 ds_read_b128 and 14 x 6 + 4 x 6 vector instructions (products formed once, one packed
// FMA per source).  Memory side, weights and epilogue as in kernels_cov_quad.hip: 32 bins x 2 frames per wave and step,
// 4-stage global_load_lds ring, weights of the wave's frames as scalar loads from the pre-pass table one step ahead,
// float64 sum of the four waves (frame phases), float64 packed partials.
//
// Measured at 2048 bins x 4000 frames x 8 channels: 3 / 4 sources 151 / 167 us (two passes of
// cov_dma_kernel) -> 108 / 117 us; 8 sources (two passes instead of four) 204 us.

#include <cstdint>

#include "oiva_device.h"
#include "cov_arith.h"

namespace oiva {
namespace {

constexpr int kP32Bins = 32;                            // bins per workgroup
constexpr int kP32Stages = 4;
constexpr int kP32Frames = 2;                           // frames per step of a wave
constexpr int kP32Slot = kP32Bins * 64;                 // bytes of 32 bins x 8 channels of one frame
constexpr int kP32Stage = kP32Frames * kP32Slot;        // bytes per stage per wave
constexpr int kP32Pairs = 14;                           // complex entries per lane: 6 + 8
constexpr int kP32Acc = 4 + 2 * kP32Pairs;              // 32 floats per lane and source
constexpr int kP32Sources = 4;                          // sources per pass
constexpr int kP32Chunk = 16;
constexpr int kP32LdsStride = kBlock + 1;
constexpr int kP32WeightStride = 16;                    // row stride of the weight table (launch_cov_weights)

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

// the five 16-byte operand reads of one frame (own group: 2, its two rows of A x B: 1, group B: 2) and, for the first
// frame of a stage, the counted wait for that stage's DMA.  ad = {own, half, far}.
template <int OFF, bool WAIT>
__device__ __forceinline__ void p32_read(const unsigned (&ad)[3], float4 (&v)[5]) {
    if constexpr (WAIT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kP32Stages - 1) * 2 * kP32Frames) : "memory");
    asm volatile(
        "ds_read_b128 %0, %5 offset:%8\n\t"
        "ds_read_b128 %1, %5 offset:%9\n\t"
        "ds_read_b128 %2, %6 offset:%8\n\t"
        "ds_read_b128 %3, %7 offset:%8\n\t"
        "ds_read_b128 %4, %7 offset:%9\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4])
        : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "n"(OFF), "n"(OFF + 16)
        : "memory");
}

struct P32Acc {
    v2f pair[kP32Sources][kP32Pairs];
    float diag[kP32Sources][4];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < kP32Sources; ++k) {
#pragma unroll
            for (int i = 0; i < kP32Pairs; ++i) pair[k][i] = v2f{0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) diag[k][c] = 0.f;
        }
    }
    // wa = (w_0, w_1), wb = (w_2, w_3).  Volatile asm keeps source order = issue order: groups of 7 independent
    // instructions, each dependent one 7 issues behind its producer.
    __device__ __forceinline__ void add(const v2f (&own)[4], const v2f (&half)[2], const v2f (&far)[4], v2f wa, v2f wb) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const v2f sq = own[c] * own[c];
            const float p = sq.x + sq.y;
            diag[0][c] = fmaf(wa.x, p, diag[0][c]);
            diag[1][c] = fmaf(wa.y, p, diag[1][c]);
            diag[2][c] = fmaf(wb.x, p, diag[2][c]);
            diag[3][c] = fmaf(wb.y, p, diag[3][c]);
        }
        v2f a[kP32Pairs], b[kP32Pairs];
        int n = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = r + 1; c < 4; ++c) {
                a[n] = own[r];
                b[n] = own[c];
                ++n;
            }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a[n] = half[r];
                b[n] = far[c];
                ++n;
            }
        constexpr int G = 7;
#pragma unroll
        for (int g0 = 0; g0 < kP32Pairs; g0 += G) {
            v2f p[G];
#pragma unroll
            for (int e = 0; e < G; ++e) p[e] = qk_mul_lo_negim(a[g0 + e], b[g0 + e]);
#pragma unroll
            for (int e = 0; e < G; ++e) qk_fma_hi_swap(a[g0 + e], b[g0 + e], p[e]);
#pragma unroll
            for (int e = 0; e < G; ++e) qk_fma_w0(wa, p[e], pair[0][g0 + e]);
#pragma unroll
            for (int e = 0; e < G; ++e) qk_fma_w1(wa, p[e], pair[1][g0 + e]);
#pragma unroll
            for (int e = 0; e < G; ++e) qk_fma_w0(wb, p[e], pair[2][g0 + e]);
#pragma unroll
            for (int e = 0; e < G; ++e) qk_fma_w1(wb, p[e], pair[3][g0 + e]);
        }
    }
    // accumulator e = k * 32 + a;  a < 4: diagonal, else (re, im) of entry (a - 4) / 2
    __device__ __forceinline__ float at(int e) const {
        const int k = e / kP32Acc, a = e % kP32Acc;
        if (a < 4) return diag[k][a];
        return ((a - 4) & 1) ? pair[k][(a - 4) >> 1].y : pair[k][(a - 4) >> 1].x;
    }
};

template <bool UNIT>
__global__ __launch_bounds__(kBlock, 2) void cov_pair32_kernel(const float2* __restrict__ X, const float* __restrict__ Wt,
                                                               double* __restrict__ Vpart, int T, int F, int K, int tc) {
    constexpr int M = 8;
    __shared__ float4 ring[kWaves * kP32Stages * kP32Stage / 16];      // 64 KB: two workgroups per CU
    static_assert(sizeof(float4) * (kWaves * kP32Stages * kP32Stage / 16) >= sizeof(float) * kP32Chunk * kP32LdsStride,
                  "reduction scratch aliases the ring");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = lane >> 1;                            // bin inside the group of 32
    const int j = lane & 1;                             // member of the pair
    const int f0 = blockIdx.x * kP32Bins;
    const int k0 = blockIdx.z * kP32Sources;
    const int t_begin = blockIdx.y * tc;
    const int t_end = min(T, t_begin + tc);
    const int nsteps = (t_end - t_begin + 4 * kP32Frames - 1) / (4 * kP32Frames);

    P32Acc acc;
    acc.clear();

    // ---- DMA side: the LDS image of a frame is the run as it lies in memory ([bin][channel]); lane l of instruction h moves
    //      16-byte piece h * 64 + l; pieces past the run (fewer than 32 bins left) re-request its last piece
    char* wring = reinterpret_cast<char*>(ring) + wave * (kP32Stages * kP32Stage);       // wave-uniform
    const int run_pieces = min(kP32Bins, F - f0) * (M / 2);
    unsigned piece_off[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) piece_off[h] = (unsigned)min(h * 64 + lane, run_pieces - 1) * 16u;
    const char* xbytes = reinterpret_cast<const char*>(X);
    const size_t row_bytes = (size_t)F * M * 8;
    const size_t run0 = (size_t)f0 * M * 8;
    auto issue = [&](int i, int s) {
#pragma unroll
        for (int u = 0; u < kP32Frames; ++u) {
            const int t = t_begin + 4 * kP32Frames * i + kP32Frames * wave + u;
            const int tcl = (i < nsteps && t < t_end) ? t : T - 1;      // steps past the end: a legal address, never consumed
            const char* src = xbytes + (size_t)tcl * row_bytes + run0;  // wave-uniform
#pragma unroll
            for (int h = 0; h < 2; ++h)
                __builtin_amdgcn_global_load_lds((gvoid_t*)(src + piece_off[h]),
                                                 (lvoid_t*)(wring + s * kP32Stage + u * kP32Slot + h * 1024), 16, 0, 0);
        }
    };

    const unsigned lbase = (unsigned)(uintptr_t)wring + (unsigned)(b * 64);
    const unsigned ad[3] = {lbase + 32u * j, lbase + 16u * j, lbase + 32u};

    // weights of the wave's two frames of a step: scalar loads from the table (T, 16), whose columns past K are 0 (the
    // pass's four sources are columns k0 .. k0 + 3 <= 11), requested one step ahead (see kernels_cov_quad.hip)
    float4 wraw[kP32Frames];
    auto request_weights = [&](int i, float4 (&raw)[kP32Frames]) {
#pragma unroll
        for (int u = 0; u < kP32Frames; ++u) {
            const int t = t_begin + 4 * kP32Frames * i + kP32Frames * wave + u;
            if constexpr (UNIT) {
                raw[u] = float4{1.f, 0.f, 0.f, 0.f};
            } else {
                const float* wp = Wt + (size_t)min(t, T - 1) * kP32WeightStride + k0;
                raw[u] = float4{wp[0], wp[1], wp[2], wp[3]};
            }
        }
    };
    auto consume = [&](int i, auto stage) {
        constexpr int S = decltype(stage)::value;
        float4 w[kP32Frames], wnext[kP32Frames];
#pragma unroll
        for (int u = 0; u < kP32Frames; ++u) {
            const bool live = t_begin + 4 * kP32Frames * i + kP32Frames * wave + u < t_end;
            w[u] = float4{live ? wraw[u].x : 0.f, live ? wraw[u].y : 0.f, live ? wraw[u].z : 0.f, live ? wraw[u].w : 0.f};
        }
#pragma unroll
        for (int u = 0; u < kP32Frames; ++u) {
            float4 v[5];
            if (u == 0)
                p32_read<S * kP32Stage, true>(ad, v);
            else
                p32_read<S * kP32Stage + kP32Slot, false>(ad, v);
            if (u == 0) {
                __builtin_amdgcn_sched_barrier(0);
                request_weights(i + 1, wnext);
                __builtin_amdgcn_sched_barrier(0);
            }
            const v2f own[4] = {v2f{v[0].x, v[0].y}, v2f{v[0].z, v[0].w}, v2f{v[1].x, v[1].y}, v2f{v[1].z, v[1].w}};
            const v2f half[2] = {v2f{v[2].x, v[2].y}, v2f{v[2].z, v[2].w}};
            const v2f far[4] = {v2f{v[3].x, v[3].y}, v2f{v[3].z, v[3].w}, v2f{v[4].x, v[4].y}, v2f{v[4].z, v[4].w}};
            acc.add(own, half, far, v2f{w[u].x, w[u].y}, v2f{w[u].z, w[u].w});
            __builtin_amdgcn_sched_barrier(0);      // keep the next frame's operand reads behind this frame's arithmetic (registers)
        }
#pragma unroll
        for (int u = 0; u < kP32Frames; ++u) wraw[u] = wnext[u];
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    static_assert(kP32Stages == 4, "the loop below is unrolled for a 4-stage ring");

    issue(0, 0);
    issue(1, 1);
    issue(2, 2);
    request_weights(0, wraw);
    int i = 0;
    for (; i + 4 <= nsteps; i += 4) {       // stage indices are compile-time constants in the unrolled body
        issue(i + 3, 3); consume(i, S0{});
        issue(i + 4, 0); consume(i + 1, S1{});
        issue(i + 5, 1); consume(i + 2, S2{});
        issue(i + 6, 2); consume(i + 3, S3{});
    }
    if (i < nsteps) { issue(i + 3, 3); consume(i, S0{}); }
    if (i + 1 < nsteps) { issue(i + 4, 0); consume(i + 1, S1{}); }
    if (i + 2 < nsteps) { issue(i + 5, 1); consume(i + 2, S2{}); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the DMA queue before the ring becomes reduction scratch

    // ---- the four waves (frame phases) of the workgroup added in float64, fixed order; one packed partial per
    //      (frame split, bin, source)
    float* lds = reinterpret_cast<float*>(ring);
    constexpr int NACC = kP32Acc * kP32Sources;
    constexpr int NA = M * M;
#pragma unroll
    for (int r0 = 0; r0 < NACC; r0 += kP32Chunk) {
        __syncthreads();
#pragma unroll
        for (int a = 0; a < kP32Chunk; ++a) lds[a * kP32LdsStride + tid] = acc.at(r0 + a);
        __syncthreads();
#pragma unroll
        for (int v = 0; v < kP32Chunk * 64 / kBlock; ++v) {
            const int idx = tid + kBlock * v;
            const int aa = idx >> 6, l = idx & 63;
            double s = 0.;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) s += (double)lds[aa * kP32LdsStride + w * 64 + l];
            const int e = r0 + aa;
            const int kk = e / kP32Acc;             // constant per round (32 % 16 == 0)
            const int fo = f0 + (l >> 1);
            if (fo < F && k0 + kk < K)
                Vpart[(((size_t)blockIdx.y * F + fo) * K + k0 + kk) * NA + pair_position(l & 1, e % kP32Acc)] = s;
        }
    }
}

}  // namespace

bool cov_pair32_supported(int M, int K) { return M == 8 && K >= 3; }
int cov_pair32_sources_per_pass() { return kP32Sources; }
int cov_pair32_bins_per_block() { return kP32Bins; }

hipError_t launch_cov_pair32(hipStream_t s, const float2* X, const float* R, float* Wt, float* wscale, int model, int raw,
                             double* Vpart, int T, int F, int M, int K, const CovGeom& g) {
    if (M != 8 || g.tc % (4 * kP32Frames) != 0) return hipErrorInvalidValue;
    if (R == nullptr) {       // unit weights (Cx of a plan whose weighted pass runs here): one "source"
        if (K != 1) return hipErrorInvalidValue;
        return launch_dominant(cov_pair32_kernel<true>, dim3(g.nbg, g.nsplit, 1), dim3(kBlock), 0, s, X, (const float*)nullptr, Vpart, T,
                               F, K, g.tc);
    }
    if (!cov_pair32_supported(M, K) || Wt == nullptr) return hipErrorInvalidValue;
    hipError_t e = launch_cov_weights(s, R, Wt, wscale, model, raw, T, K, kP32WeightStride);
    if (e != hipSuccess) return e;
    return launch_dominant(cov_pair32_kernel<false>, dim3(g.nbg, g.nsplit, (K + kP32Sources - 1) / kP32Sources), dim3(kBlock), 0, s, X,
                           (const float*)Wt, Vpart, T, F, K, g.tc);
}

}  // namespace oiva
