// Weighted spatial covariance pass for 10, 12, 14 and 16 channels and FEW sources (K <= 4), vector-ALU form.
//
//   V_k[f] = sum_t rinv[t,k] * x_{t,f} x_{t,f}^H          reference overiva.py:179 (two k per pass over X)
//   Cx[f]  = sum_t x_{t,f} x_{t,f}^H                      reference overiva.py:87   (unit weights)
//
// The planar matrix-core kernel of kernels_cov_mfma.hip pays for a full 16 x 16 real tile per product term whatever
// K is; with one or two sources that is 2-3 times the time the memory system needs to deliver X.  Here the Hermitian
// HALF of the 16 x 16 matrix is formed on the vector ALU, split over FOUR lanes per (bin, frame) so that a lane carries
// 64 accumulators per source like the 8-channel kernel (kernels_cov.hip): channels in four groups A B C D of four,
// lane j of the quad takes
//     the diagonal block of its own group j                  (4 real + 6 complex entries)
//     the block  group j  x  group j+1 (mod 4)                (16 complex entries; j = 3 yields the conjugate of D x A's
//                                                             transpose, undone when the partial is stored)
//     one half of the block  group g  x  group g+2, g = j mod 2: rows 2(j/2), 2(j/2)+1 of group g  (8 complex entries)
// = 34 of the 136 entries each, with the same instruction stream in every lane: only the LDS addresses the operands are
// read from differ.  Channels past M (10, 12, 14 channels) are read from a clamped address and the entries they produce
// are dropped at the store, so no padding of X is needed.
//
// Memory: a wave takes 16 bins x 2 consecutive frames per step; their rows are contiguous runs of 16*M*8 bytes in the
// native (T, F, M) tensor, moved HBM -> LDS by four fully coalesced global_load_lds instructions into a private 4-stage
// ring (3 steps = 12 KB in flight per wave, 96 KB per CU at two workgroups), ordered by the wave's own vmcnt only.
// The final weights w[t,k] = 1 / max(r / gamma, eps) come from the table the pre-pass of the matrix-core path already
// produces (launch_cov_weights): a wave's frames are wave-uniform, so the pair (w_0, w_1) of a frame is one scalar load
// and stays in scalar registers as the broadcast operand of the packed FMAs -- computing the two divides in the loop cost
// 22 of 177 vector instructions per frame in a kernel whose vector ALU is the busiest unit.  The four waves of a
// workgroup take frames t = 8 i + 2 w + u; their per-lane sums (fp32 chains of T / (4 nsplit) frames) are added in
// float64 through LDS and stored as float64 packed Hermitian partials, the layout the update kernels read.
//
// Measured at 2048 bins x 4000 frames x 16 channels (1.05 GB of X), two sources: 253-293 us (matrix cores 288-349); what
// bounds it: DESIGN.md 3.4.1.

#include <cstdint>

#include "oiva_device.h"
#include "cov_arith.h"

namespace oiva {
namespace {

constexpr int kQuadStages = 4;
constexpr int kQuadFrames = 2;                          // frames per step of a wave
constexpr int kQuadSlot = 2048;                         // bytes of 16 bins x (<= 16) channels of one frame
constexpr int kQuadStage = kQuadFrames * kQuadSlot;     // bytes per stage per wave
constexpr int kQuadPairs = 30;                          // complex entries per lane
constexpr int kQuadAcc = 4 + 2 * kQuadPairs;            // 64 floats per lane and source
constexpr int kQuadChunk = 16;
constexpr int kQuadLdsStride = kBlock + 1;
constexpr int kQuadWeightStride = 16;                   // row stride of the weight table (launch_cov_weights)

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

// The seven 16-byte operand reads of one frame (own group: 2, next group: 2, far group: 2, the two rows of the half
// block: 1) and, for the first frame of a stage, the counted wait for that stage's DMA -- asm, because hipcc drains the
// whole DMA queue (vmcnt(0)) in front of any LDS read it can see.  ad = {own, next, far, half}.
template <int OFF, bool WAIT>
__device__ __forceinline__ void quad_read(const unsigned (&ad)[4], float4 (&v)[7]) {
    if constexpr (WAIT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kQuadStages - 1) * 2 * kQuadFrames) : "memory");
    asm volatile(
        "ds_read_b128 %0, %7 offset:%11\n\t"
        "ds_read_b128 %1, %7 offset:%12\n\t"
        "ds_read_b128 %2, %8 offset:%11\n\t"
        "ds_read_b128 %3, %8 offset:%12\n\t"
        "ds_read_b128 %4, %9 offset:%11\n\t"
        "ds_read_b128 %5, %9 offset:%12\n\t"
        "ds_read_b128 %6, %10 offset:%11\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6])
        : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "n"(OFF), "n"(OFF + 16)
        : "memory");
}

// entry p (0..29) of a lane = a[p] conj(b[p]):  p < 6 own x own (rows < columns), p < 22 own x next, else half x far
struct QuadOperands {
    v2f a[kQuadPairs], b[kQuadPairs];
    __device__ __forceinline__ QuadOperands(const v2f (&own)[4], const v2f (&next)[4], const v2f (&half)[2], const v2f (&far)[4]) {
        int p = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = r + 1; c < 4; ++c) {
                a[p] = own[r];
                b[p] = own[c];
                ++p;
            }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a[p] = own[r];
                b[p] = next[c];
                ++p;
            }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a[p] = half[r];
                b[p] = far[c];
                ++p;
            }
    }
};

template <int KC>
struct QuadAcc {
    v2f pair[KC][kQuadPairs];
    float diag[KC][4];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < KC; ++k) {
#pragma unroll
            for (int i = 0; i < kQuadPairs; ++i) pair[k][i] = v2f{0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) diag[k][c] = 0.f;
        }
    }
    // w = (w_0, w_1) (KC == 1: w_1 unused).  Inline asm is kept in source order, so the order written here is the issue
    // order: groups of 8 independent instructions, each dependent one 8 issues behind its producer.
    __device__ __forceinline__ void add(const v2f (&own)[4], const v2f (&next)[4], const v2f (&half)[2], const v2f (&far)[4], v2f w) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const v2f sq = own[c] * own[c];
            const float p = sq.x + sq.y;
            diag[0][c] = fmaf(w.x, p, diag[0][c]);
            if constexpr (KC == 2) diag[1][c] = fmaf(w.y, p, diag[1][c]);
        }
        const QuadOperands o(own, next, half, far);
        constexpr int G = 8;
#pragma unroll
        for (int g0 = 0; g0 < kQuadPairs; g0 += G) {
            v2f p[G];
#pragma unroll
            for (int e = 0; e < G; ++e)
                if (g0 + e < kQuadPairs) p[e] = qk_mul_lo_negim(o.a[g0 + e], o.b[g0 + e]);
#pragma unroll
            for (int e = 0; e < G; ++e)
                if (g0 + e < kQuadPairs) qk_fma_hi_swap(o.a[g0 + e], o.b[g0 + e], p[e]);
#pragma unroll
            for (int e = 0; e < G; ++e)
                if (g0 + e < kQuadPairs) qk_fma_w0(w, p[e], pair[0][g0 + e]);
            if constexpr (KC == 2) {
#pragma unroll
                for (int e = 0; e < G; ++e)
                    if (g0 + e < kQuadPairs) qk_fma_w1(w, p[e], pair[1][g0 + e]);
            }
        }
    }
    // accumulator e = k * 64 + a;  a < 4: diagonal, else (re, im) of entry (a - 4) / 2
    __device__ __forceinline__ float at(int e) const {
        const int k = e / kQuadAcc, a = e % kQuadAcc;
        if (a < 4) return diag[k][a];
        return ((a - 4) & 1) ? pair[k][(a - 4) >> 1].y : pair[k][(a - 4) >> 1].x;
    }
};

// accumulator a (0..63) of quad lane j -> position in the packed Hermitian layout of an M x M matrix (herm_pair_index:
// M real diagonals, then (re, im) of the entries c < d row by row), or -1 when the entry involves a channel >= M;
// neg: the lane holds the conjugate of the stored entry (only its imaginary part differs)
__device__ __forceinline__ int quad_position(int j, int a, int M, bool* neg) {
    *neg = false;
    if (a < 4) {
        const int c = 4 * j + a;
        return c < M ? c : -1;
    }
    const int p = (a - 4) >> 1, im = (a - 4) & 1;
    int c, d;
    if (p < 6) {
        const int r = p < 3 ? 0 : (p < 5 ? 1 : 2);
        const int cc = p < 3 ? p + 1 : (p < 5 ? p - 1 : 3);
        c = 4 * j + r;
        d = 4 * j + cc;
    } else if (p < 22) {
        c = 4 * j + ((p - 6) >> 2);
        d = 4 * ((j + 1) & 3) + ((p - 6) & 3);
    } else {
        const int g = j & 1;
        c = 4 * g + 2 * (j >> 1) + ((p - 22) >> 2);
        d = 4 * (g + 2) + ((p - 22) & 3);
    }
    if (c > d) {
        const int t = c;
        c = d;
        d = t;
        *neg = im != 0;
    }
    if (d >= M) return -1;
    return herm_pair_index(M, c, d) + im;
}

template <int KC, bool UNIT>
__global__ __launch_bounds__(kBlock, 2) void cov_quad_kernel(const float2* __restrict__ X, const float* __restrict__ Wt,
                                                             double* __restrict__ Vpart, int T, int F, int M, int Mv, int K, int tc) {
    __shared__ float4 ring[kWaves * kQuadStages * kQuadStage / 16];      // 64 KB: two workgroups per CU
    static_assert(sizeof(float4) * (kWaves * kQuadStages * kQuadStage / 16) >= sizeof(float) * kQuadChunk * kQuadLdsStride,
                  "reduction scratch aliases the ring");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = lane >> 2;                            // bin inside the group of 16
    const int j = lane & 3;                             // member of the quad
    const int f0 = blockIdx.x * kBinsPerWave;
    const int k0 = blockIdx.z * KC;
    const int t_begin = blockIdx.y * tc;
    const int t_end = min(T, t_begin + tc);
    const int nsteps = (t_end - t_begin + 4 * kQuadFrames - 1) / (4 * kQuadFrames);

    QuadAcc<KC> acc;
    acc.clear();
    // ---- DMA side: the LDS image of a frame is the run as it lies in memory ([bin][channel], bin stride M*8 bytes): lane l
    //      of instruction h moves 16-byte piece h * 64 + l; pieces past the run (fewer than 16 channels, fewer than 16 bins
    //      left) re-request its last piece and land in unused LDS.  (ds_read_b128 on this image is 2-way bank-conflicted
    //      for 16 channels; an XOR-swizzled image -- free with global_load_lds, a lane may fetch any piece -- measured the
    //      same kernel time, and so did 8-byte reads with 4-way conflicts: the LDS is not what the kernel waits for.)
    char* wring = reinterpret_cast<char*>(ring) + wave * (kQuadStages * kQuadStage);       // wave-uniform
    const int run_pieces = min(kBinsPerWave, F - f0) * M / 2;                            // 16-byte pieces of the run
    unsigned piece_off[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) piece_off[h] = (unsigned)min(h * 64 + lane, run_pieces - 1) * 16u;
    const char* xbytes = reinterpret_cast<const char*>(X);
    const size_t row_bytes = (size_t)F * M * 8;
    const size_t run0 = (size_t)f0 * M * 8;
    auto issue = [&](int i, int s) {
#pragma unroll
        for (int u = 0; u < kQuadFrames; ++u) {
            const int t = t_begin + 4 * kQuadFrames * i + kQuadFrames * wave + u;
            const int tcl = (i < nsteps && t < t_end) ? t : T - 1;      // steps past the end: a legal address, never consumed
            const char* src = xbytes + (size_t)tcl * row_bytes + run0;  // wave-uniform
#pragma unroll
            for (int h = 0; h < 2; ++h)
                __builtin_amdgcn_global_load_lds((gvoid_t*)(src + piece_off[h]),
                                                 (lvoid_t*)(wring + s * kQuadStage + u * kQuadSlot + h * 1024), 16, 0, 0);
        }
    };

    // ---- operand addresses of this lane inside a frame slot.  Channels past M are not clamped: bin b's row starts at
    //      b*M*8, so channel c <= 15 of any bin still lies inside the 2 KB slot (15*M*8 + 128 <= 2048); what such a read
    //      returns (the next bin's data, stale LDS) only reaches entries that are dropped at the store
    const unsigned lbase = (unsigned)(uintptr_t)wring + (unsigned)(b * M * 8);
    const int g = j & 1;
    const unsigned ad[4] = {lbase + 32u * j, lbase + 32u * ((j + 1) & 3), lbase + 32u * (g + 2), lbase + 32u * g + 16u * (j >> 1)};

    // the weights of the wave's two frames of a step: wave-uniform scalar loads from the table (T, 16), whose columns past K
    // are 0.  They are requested ONE STEP AHEAD (between the first frame's operand reads and its arithmetic): fetched at
    // the head of their own step, every step began with a wait for a scalar load that misses the scalar cache (the table
    // is walked with a stride of 8 rows) -- 27 % of the wave cycles of the arithmetic-only form of this kernel (PMC).
    v2f wraw[kQuadFrames];
    auto request_weights = [&](int i, v2f (&raw)[kQuadFrames]) {
#pragma unroll
        for (int u = 0; u < kQuadFrames; ++u) {
            if constexpr (UNIT) {
                raw[u] = v2f{1.f, 0.f};
            } else {
                const int t = t_begin + 4 * kQuadFrames * i + kQuadFrames * wave + u;
                const float* wp = Wt + (size_t)min(t, T - 1) * kQuadWeightStride + k0;
                raw[u] = v2f{wp[0], KC == 2 ? wp[1] : 0.f};
            }
        }
    };
    auto consume = [&](int i, auto stage) {
        constexpr int S = decltype(stage)::value;
        v2f w[kQuadFrames], wnext[kQuadFrames];
#pragma unroll
        for (int u = 0; u < kQuadFrames; ++u) {
            const bool live = t_begin + 4 * kQuadFrames * i + kQuadFrames * wave + u < t_end;
            w[u] = v2f{live ? wraw[u].x : 0.f, live ? wraw[u].y : 0.f};
        }
#pragma unroll
        for (int u = 0; u < kQuadFrames; ++u) {
            float4 v[7];
            if (u == 0)
                quad_read<S * kQuadStage, true>(ad, v);
            else
                quad_read<S * kQuadStage + kQuadSlot, false>(ad, v);
            const v2f own[4] = {v2f{v[0].x, v[0].y}, v2f{v[0].z, v[0].w}, v2f{v[1].x, v[1].y}, v2f{v[1].z, v[1].w}};
            const v2f next[4] = {v2f{v[2].x, v[2].y}, v2f{v[2].z, v[2].w}, v2f{v[3].x, v[3].y}, v2f{v[3].z, v[3].w}};
            const v2f far[4] = {v2f{v[4].x, v[4].y}, v2f{v[4].z, v[4].w}, v2f{v[5].x, v[5].y}, v2f{v[5].z, v[5].w}};
            const v2f half[2] = {v2f{v[6].x, v[6].y}, v2f{v[6].z, v[6].w}};
            if (u == 0) {
                __builtin_amdgcn_sched_barrier(0);
                request_weights(i + 1, wnext);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc.add(own, next, half, far, w[u]);
            __builtin_amdgcn_sched_barrier(0);      // keep the next frame's operand reads behind this frame's arithmetic (registers)
        }
#pragma unroll
        for (int u = 0; u < kQuadFrames; ++u) wraw[u] = wnext[u];
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    static_assert(kQuadStages == 4, "the loop below is unrolled for a 4-stage ring");

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
    constexpr int NACC = kQuadAcc * KC;
    const int NA = Mv * Mv;            // Mv <= M: the matrix that is stored (M: channel pitch of X)
#pragma unroll
    for (int r0 = 0; r0 < NACC; r0 += kQuadChunk) {
        __syncthreads();
#pragma unroll
        for (int a = 0; a < kQuadChunk; ++a) lds[a * kQuadLdsStride + tid] = acc.at(r0 + a);
        __syncthreads();
#pragma unroll
        for (int v = 0; v < kQuadChunk * 64 / kBlock; ++v) {
            const int idx = tid + kBlock * v;
            const int aa = idx >> 6, l = idx & 63;
            double s = 0.;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) s += (double)lds[aa * kQuadLdsStride + w * 64 + l];
            const int e = r0 + aa;
            const int kk = e / kQuadAcc;            // constant per round (64 % 16 == 0)
            const int fo = f0 + (l >> 2);
            bool neg;
            const int pos = quad_position(l & 3, e % kQuadAcc, Mv, &neg);
            if (pos >= 0 && fo < F && k0 + kk < K)
                Vpart[(((size_t)blockIdx.y * F + fo) * K + k0 + kk) * NA + pos] = neg ? -s : s;
        }
    }
}

}  // namespace

// M: channel pitch of X (even); Mv <= M: channels of the matrices (an odd channel count runs on a copy of X padded by one zero channel)
bool cov_quad_supported(int M, int K) { return M >= 10 && M <= 16 && M % 2 == 0 && K >= 1 && K <= 4; }

int cov_quad_sources_per_pass(int K) { return K >= 2 ? 2 : 1; }

hipError_t launch_cov_quad(hipStream_t s, const float2* X, const float* R, float* Wt, float* wscale, int model, int raw,
                           double* Vpart, int T, int F, int M, int Mv, int K, const CovGeom& g) {
    if (!cov_quad_supported(M, K) || Mv > M || Mv < M - 1 || g.tc % (4 * kQuadFrames) != 0) return hipErrorInvalidValue;
    if (R == nullptr) {
        if (K != 1) return hipErrorInvalidValue;
        return launch_dominant(cov_quad_kernel<1, true>, dim3(g.nbg, g.nsplit, 1), dim3(kBlock), 0, s, X, (const float*)nullptr, Vpart,
                               T, F, M, Mv, K, g.tc);
    }
    if (Wt == nullptr) return hipErrorInvalidValue;
    hipError_t e = launch_cov_weights(s, R, Wt, wscale, model, raw, T, K, kQuadWeightStride);
    if (e != hipSuccess) return e;
    if (g.kc == 2)
        return launch_dominant(cov_quad_kernel<2, false>, dim3(g.nbg, g.nsplit, (K + 1) / 2), dim3(kBlock), 0, s, X, (const float*)Wt,
                               Vpart, T, F, M, Mv, K, g.tc);
    return launch_dominant(cov_quad_kernel<1, false>, dim3(g.nbg, g.nsplit, K), dim3(kBlock), 0, s, X, (const float*)Wt, Vpart, T, F,
                           M, Mv, K, g.tc);
}

}  // namespace oiva
