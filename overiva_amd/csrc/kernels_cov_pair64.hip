// Weighted spatial covariance accumulated in float64 for 8 channels: the covariance pass of the "precise" arithmetic
// (OIVA_PREC_COV_F64), vector-ALU form.
//
//   V_k[f] = sum_t w_k[t] * x_{t,f} x_{t,f}^H            reference overiva.py:179 (two k per pass over X), :87 (w = 1)
//
// The reference's r_inv is float64, which silently promotes this product to complex128 even for complex64 input
// (overiva.py:127-128,179).  Products of float32 data are exact in float64, so converting x and forming the Hermitian HALF
// with v_mul_f64 / v_fma_f64 is that arithmetic (up to the order of the sum).  Rounds 1-2 ran this pass on the fp64 matrix
// cores in the real Gram form (kernels_cov_gram.hip, removed): both triangles and the re/im cross terms twice on a pipe
// that sustains 44 TFLOP/s here, 197-222 us at the headline shape.  The vector ALU issues a v_fma_f64 every 4 cycles per
// SIMD (tools/pkbench.hip: 2.0-2.2 ns, the same slot as a packed fp32 FMA = 60 TFLOP/s) and the Hermitian half needs a
// quarter of the Gram form's multiply-adds.
//
// 64 float64 accumulators per source do not fit one lane twice over (two sources, two waves per SIMD), so the matrix is
// split over TWO lanes per (bin, frame), same instruction stream in both (cf. kernels_cov_quad.hip): channels in two
// groups A = 0..3, B = 4..7; lane j takes the diagonal block of its own group (4 real + 6 complex entries) and rows
// 2j, 2j+1 of the block A x B (8 complex) = 32 float64 sums per source.  Only the LDS addresses differ between the lanes.
//
// Memory: a workgroup = 32 bins x 4 frame phases (waves); a wave takes 32 bins x 2 consecutive frames per step, contiguous
// runs of 2 KB in the native (T, F, 8) tensor, four fully coalesced global_load_lds per step into a private 4-stage ring
// (12 KB in flight per wave, 96 KB per CU), ordered by the wave's own vmcnt.  The float64 weights 1 / max(r / gamma, eps)
// come from a (T, 2 per pass) table written by a pre-pass; a wave's frames are wave-uniform, so they are scalar loads,
// requested one step ahead.  (Three workgroups per CU -- 168 registers, a 3-stage ring of 48 KB -- measured no faster.)
// The four waves' sums are added in float64 through LDS (fixed order) and stored as one packed
// Hermitian partial per (frame split, bin, source).

#include <cstdint>

#include "oiva_device.h"
#include "cov_arith.h"

namespace oiva {
namespace {

constexpr int kPairBins = 32;                            // bins per workgroup
constexpr int kPairStages = 4;
constexpr int kPairFrames = 2;                           // frames per step of a wave
constexpr int kPairSlot = kPairBins * 64;                // bytes of 32 bins x 8 channels of one frame
constexpr int kPairStage = kPairFrames * kPairSlot;      // bytes per stage per wave
constexpr int kPairAcc = 32;                             // float64 sums per lane and source: 4 + 2 * (6 + 8)
constexpr int kPairChunk = 16;
constexpr int kPairLdsStride = kBlock + 1;
constexpr int kPairWeightStride = 8;                     // doubles per frame in the weight table (64 bytes, as the float tables)

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

// The five 16-byte operand reads of one frame (own group: 2, its two rows of A x B: 1, group B: 2) and, for the first
// frame of a stage, the counted wait for that stage's DMA -- asm, because hipcc drains the whole DMA queue (vmcnt(0)) in
// front of any LDS read it can see.  ad = {own, half, far}.
template <int OFF, bool WAIT>
__device__ __forceinline__ void pair_read(const unsigned (&ad)[3], float4 (&v)[5]) {
    if constexpr (WAIT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kPairStages - 1) * 2 * kPairFrames) : "memory");
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

// acc[k][*] += w[k] * (entries of this lane):  x_c conj(x_d) = (xr_c xr_d + xi_c xi_d,  xi_c xr_d - xr_c xi_d)
template <int KC>
__device__ __forceinline__ void pair_accumulate(double (&acc)[KC][kPairAcc], const double (&w)[KC], const float4 (&v)[5]) {
    double own_r[4], own_i[4], half_r[2], half_i[2], far_r[4], far_i[4];
    own_r[0] = v[0].x, own_i[0] = v[0].y, own_r[1] = v[0].z, own_i[1] = v[0].w;
    own_r[2] = v[1].x, own_i[2] = v[1].y, own_r[3] = v[1].z, own_i[3] = v[1].w;
    half_r[0] = v[2].x, half_i[0] = v[2].y, half_r[1] = v[2].z, half_i[1] = v[2].w;
    far_r[0] = v[3].x, far_i[0] = v[3].y, far_r[1] = v[3].z, far_i[1] = v[3].w;
    far_r[2] = v[4].x, far_i[2] = v[4].y, far_r[3] = v[4].z, far_i[3] = v[4].w;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double p = fma(own_r[c], own_r[c], own_i[c] * own_i[c]);
#pragma unroll
        for (int k = 0; k < KC; ++k) acc[k][c] = fma(w[k], p, acc[k][c]);
    }
    int a = 4;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = c + 1; d < 4; ++d) {
            const double pre = fma(own_r[c], own_r[d], own_i[c] * own_i[d]);
            const double pim = fma(own_i[c], own_r[d], -(own_r[c] * own_i[d]));
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                acc[k][a] = fma(w[k], pre, acc[k][a]);
                acc[k][a + 1] = fma(w[k], pim, acc[k][a + 1]);
            }
            a += 2;
        }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const double pre = fma(half_r[c], far_r[d], half_i[c] * far_i[d]);
            const double pim = fma(half_i[c], far_r[d], -(half_r[c] * far_i[d]));
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                acc[k][a] = fma(w[k], pre, acc[k][a]);
                acc[k][a + 1] = fma(w[k], pim, acc[k][a + 1]);
            }
            a += 2;
        }
}

// Final float64 weights of one pass: Wt[t][kk] = 1 / max(r[t, k0 + kk] / gamma, eps) (overiva.py:158-173), 0 for sources
// that do not exist; one thread per frame.  Also writes wscale (pass 0 of the call writes all K).
__global__ __launch_bounds__(kBlock) void pair_weights_kernel(const float* __restrict__ R, double* __restrict__ Wt,
                                                              float* __restrict__ wscale, int model, int raw, int T, int K) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= T) return;
#pragma unroll
    for (int k = 0; k < kPairWeightStride; ++k) {
        double w = 0.;
        if (k < K) {
            const double gamma = (raw & 1) ? 1. : gamma_of(R, T, K, k);
            double rn = (double)R[(size_t)t * K + k] / gamma;
            rn = rn < (double)kEpsR ? (double)kEpsR : rn;          // a NaN stays NaN, like r[r < eps] = eps in the reference
            w = 1. / rn;
            if (t == 0 && wscale != nullptr && !(raw & 1))
                wscale[k] = model == OIVA_MODEL_LAPLACE ? (float)gamma : (float)sqrt(gamma);   // overiva.py:163 / :167
        }
        Wt[(size_t)t * kPairWeightStride + k] = w;
    }
}

template <int KC, bool UNIT>
__global__ __launch_bounds__(kBlock, 2) void cov_pair64_kernel(const float2* __restrict__ X, const double* __restrict__ Wt,
                                                               double* __restrict__ Vpart, int T, int F, int K, int tc) {
    constexpr int M = 8;
    __shared__ float4 ring[kWaves * kPairStages * kPairStage / 16];      // 64 KB: two workgroups per CU
    static_assert(sizeof(float4) * (kWaves * kPairStages * kPairStage / 16) >= sizeof(double) * kPairChunk * kPairLdsStride,
                  "reduction scratch aliases the ring");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = lane >> 1;                            // bin inside the group of 32
    const int j = lane & 1;                             // member of the pair
    const int f0 = blockIdx.x * kPairBins;
    const int k0 = blockIdx.z * KC;
    const int t_begin = blockIdx.y * tc;
    const int t_end = min(T, t_begin + tc);
    const int nsteps = (t_end - t_begin + 4 * kPairFrames - 1) / (4 * kPairFrames);

    double acc[KC][kPairAcc];
#pragma unroll
    for (int k = 0; k < KC; ++k)
#pragma unroll
        for (int a = 0; a < kPairAcc; ++a) acc[k][a] = 0.;

    // ---- DMA side: the LDS image of a frame is the run as it lies in memory ([bin][channel]); lane l of instruction h moves
    //      16-byte piece h * 64 + l; pieces past the run (fewer than 32 bins left) re-request its last piece
    char* wring = reinterpret_cast<char*>(ring) + wave * (kPairStages * kPairStage);       // wave-uniform
    const int run_pieces = min(kPairBins, F - f0) * (M / 2);
    unsigned piece_off[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) piece_off[h] = (unsigned)min(h * 64 + lane, run_pieces - 1) * 16u;
    const char* xbytes = reinterpret_cast<const char*>(X);
    const size_t row_bytes = (size_t)F * M * 8;
    const size_t run0 = (size_t)f0 * M * 8;
    auto issue = [&](int i, int s) {
#pragma unroll
        for (int u = 0; u < kPairFrames; ++u) {
            const int t = t_begin + 4 * kPairFrames * i + kPairFrames * wave + u;
            const int tcl = (i < nsteps && t < t_end) ? t : T - 1;      // steps past the end: a legal address, never consumed
            const char* src = xbytes + (size_t)tcl * row_bytes + run0;  // wave-uniform
#pragma unroll
            for (int h = 0; h < 2; ++h)
                __builtin_amdgcn_global_load_lds((gvoid_t*)(src + piece_off[h]),
                                                 (lvoid_t*)(wring + s * kPairStage + u * kPairSlot + h * 1024), 16, 0, 0);
        }
    };

    const unsigned lbase = (unsigned)(uintptr_t)wring + (unsigned)(b * 64);
    const unsigned ad[3] = {lbase + 32u * j, lbase + 16u * j, lbase + 32u};

    // weights of the wave's two frames of a step: scalar loads, requested one step ahead (see kernels_cov_quad.hip)
    double wraw[kPairFrames][KC];
    auto request_weights = [&](int i, double (&raw)[kPairFrames][KC]) {
#pragma unroll
        for (int u = 0; u < kPairFrames; ++u) {
            const int t = t_begin + 4 * kPairFrames * i + kPairFrames * wave + u;
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                if constexpr (UNIT)
                    raw[u][k] = k == 0 ? 1. : 0.;
                else
                    raw[u][k] = Wt[(size_t)min(t, T - 1) * kPairWeightStride + (k0 + k < kPairWeightStride ? k0 + k : 0)];
            }
        }
    };
    auto consume = [&](int i, auto stage) {
        constexpr int S = decltype(stage)::value;
        double w[kPairFrames][KC], wnext[kPairFrames][KC];
#pragma unroll
        for (int u = 0; u < kPairFrames; ++u) {
            const bool live = t_begin + 4 * kPairFrames * i + kPairFrames * wave + u < t_end;
#pragma unroll
            for (int k = 0; k < KC; ++k) w[u][k] = (live && (UNIT || k0 + k < K)) ? wraw[u][k] : 0.;
        }
#pragma unroll
        for (int u = 0; u < kPairFrames; ++u) {
            float4 v[5];
            if (u == 0)
                pair_read<S * kPairStage, true>(ad, v);
            else
                pair_read<S * kPairStage + kPairSlot, false>(ad, v);
            if (u == 0) {
                __builtin_amdgcn_sched_barrier(0);
                request_weights(i + 1, wnext);
                __builtin_amdgcn_sched_barrier(0);
            }
            pair_accumulate<KC>(acc, w[u], v);
            __builtin_amdgcn_sched_barrier(0);      // keep the next frame's operand reads behind this frame's arithmetic (registers)
        }
#pragma unroll
        for (int u = 0; u < kPairFrames; ++u)
#pragma unroll
            for (int k = 0; k < KC; ++k) wraw[u][k] = wnext[u][k];
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    static_assert(kPairStages == 4, "the loop below is unrolled for a 4-stage ring");

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

    // ---- the four waves (frame phases) of the workgroup added in fixed order; one packed partial per (frame split, bin, source)
    double* lds = reinterpret_cast<double*>(ring);
    constexpr int NACC = kPairAcc * KC;
    constexpr int NA = M * M;
#pragma unroll
    for (int r0 = 0; r0 < NACC; r0 += kPairChunk) {
        __syncthreads();
#pragma unroll
        for (int a = 0; a < kPairChunk; ++a) lds[a * kPairLdsStride + tid] = acc[(r0 + a) / kPairAcc][(r0 + a) % kPairAcc];
        __syncthreads();
#pragma unroll
        for (int v = 0; v < kPairChunk * 64 / kBlock; ++v) {
            const int idx = tid + kBlock * v;
            const int aa = idx >> 6, l = idx & 63;
            double s = 0.;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) s += lds[aa * kPairLdsStride + w * 64 + l];
            const int e = r0 + aa;
            const int kk = e / kPairAcc;            // constant per round (32 % 16 == 0)
            const int fo = f0 + (l >> 1);
            if (fo < F && k0 + kk < K)
                Vpart[(((size_t)blockIdx.y * F + fo) * K + k0 + kk) * NA + pair_position(l & 1, e % kPairAcc)] = s;
        }
    }
}

}  // namespace

bool cov_pair64_supported(int M) { return M == 8; }
int cov_pair64_sources_per_pass(int K) { return K >= 2 ? 2 : 1; }
int cov_pair64_bins_per_block() { return kPairBins; }

hipError_t launch_cov_pair64(hipStream_t s, const float2* X, const float* R, float* Wt, float* wscale, int model, int raw,
                             double* Vpart, int T, int F, int M, int K, const CovGeom& g) {
    if (!cov_pair64_supported(M) || g.tc % (4 * kPairFrames) != 0 || K > kPairWeightStride) return hipErrorInvalidValue;
    const dim3 block(kBlock);
    if (R == nullptr) {
        if (K != 1) return hipErrorInvalidValue;
        return launch_dominant(cov_pair64_kernel<1, true>, dim3(g.nbg, g.nsplit, 1), block, 0, s, X, (const double*)nullptr, Vpart, T,
                               F, K, g.tc);
    }
    if (Wt == nullptr) return hipErrorInvalidValue;
    double* wt = reinterpret_cast<double*>(Wt);       // the (T, 16) float scratch holds (T, 8) doubles
    pair_weights_kernel<<<dim3((T + kBlock - 1) / kBlock), block, 0, s>>>(R, wt, wscale, model, raw, T, K);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (g.kc == 2)
        return launch_dominant(cov_pair64_kernel<2, false>, dim3(g.nbg, g.nsplit, (K + 1) / 2), block, 0, s, X, (const double*)wt, Vpart,
                               T, F, K, g.tc);
    return launch_dominant(cov_pair64_kernel<1, false>, dim3(g.nbg, g.nsplit, K), block, 0, s, X, (const double*)wt, Vpart, T, F, K,
                           g.tc);
}

}  // namespace oiva
