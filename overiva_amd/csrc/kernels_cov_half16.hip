// Weighted spatial covariance pass for 10, 12, 14 and 16 channels and MANY sources (5..16: BASELINE configs[4] is the
// determined 16 x 16 case), float32 packed arithmetic on the vector ALU, all sources in one pass over X.
//
//   V_k[f] = sum_t rinv[t,k] * x_{t,f} x_{t,f}^H          reference overiva.py:179
//   Cx[f]  = sum_t x_{t,f} x_{t,f}^H                      reference overiva.py:87   (unit weights)
//
// On CDNA4 the fp32 matrix cores and the packed fp32 vector ALU have the same peak (157 TFLOP/s); the planar matrix-core
// kernel (kernels_cov_mfma.hip) spends 768 multiply-adds per frame and source on full 16 x 16 real tiles, the Hermitian
// half costs 272.  What stood in the way is registers: 16 sources x 256 real sums per bin.  Here a (bin, frame) is spread
// over THIRTY-TWO lanes, 5 complex entries each (160 slots for the 136 entries of the half, block-wise):
//     the matrix in 4 x 4 blocks of 4 channels; the 10 blocks (I <= J) have 40 block rows of 4 entries;
//     lane e takes block row e (blocks (0,0) (0,1) (0,2) (0,3) (1,1) (1,2) (1,3) (2,2)) and ONE entry of the 8 rows of
//     blocks (2,3), (3,3): row (e / 4) % 4 of block 8 + e / 16, column e % 4.
// One instruction stream for all lanes (only LDS addresses differ); entries below the diagonal of a diagonal block and
// entries of channels >= M are computed and dropped at the store.  Per lane and frame: 2 packed instructions per entry
// for the product and one per entry and source = 10 + 5 K; 10 K accumulator registers (160 at 16 sources), the weights
// (w_0, w_1), (w_2, w_3) ... as scalar register pairs straight from the pre-pass table.
//
// A wave = 2 bins x the frames of its phase; the 4 waves of a workgroup are the 4 frame phases of the same 2 bins.  One
// global_load_lds moves 4 frames x 256 bytes (2 bins x 16 channels) of the wave into a 4-stage ring; the wave's float32
// chains (T / (4 nsplit) frames) are added in float64 across the four waves through LDS and stored as float64 packed
// partials.  Frames past the end of a split take their weights from the zeroed row T of the table.

#include <cstdint>
#include <cstdlib>

#include "oiva_device.h"
#include "cov_arith.h"

namespace oiva {
namespace {

constexpr int kH16Stages = 4;
constexpr int kH16Frames = 4;                           // frames per stage of a wave (one DMA instruction)
constexpr int kH16Slot = 256;                           // bytes of 2 bins x (<= 16) channels of one frame
constexpr int kH16Stage = kH16Frames * kH16Slot;        // 1 KB per stage per wave
constexpr int kH16Entries = 5;                          // complex entries per lane
constexpr int kH16Chunk = 16;
constexpr int kH16LdsStride = kBlock + 1;
constexpr int kH16WeightStride = 16;                    // row stride of the weight table (launch_cov_weights)

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

__device__ __forceinline__ void h16_block(int b, int* I, int* J) {      // the 10 blocks I <= J, row-major
    const int bi = b < 4 ? 0 : (b < 7 ? 1 : (b < 9 ? 2 : 3));
    *I = bi;
    *J = b - (bi == 0 ? 0 : (bi == 1 ? 3 : (bi == 2 ? 5 : 6)));
}
// entry ent (0..4) of lane e (0..31): (row channel, column channel)
__device__ __forceinline__ void h16_entry(int e, int ent, int* ci, int* di) {
    int b, r, c;
    if (ent < 4) {
        b = e >> 2, r = e & 3, c = ent;
    } else {
        b = 8 + (e >> 4), r = (e >> 2) & 3, c = e & 3;
    }
    int I, J;
    h16_block(b, &I, &J);
    *ci = 4 * I + r;
    *di = 4 * J + c;
}

// The operand reads of one frame: the lane's row channel (8 bytes), the 4 column channels of its block (2 x 16 bytes), row
// and column channel of its fifth entry (8 bytes each); for the first frame of a stage the counted wait for that stage's
// DMA -- asm, because hipcc drains the whole DMA queue in front of any LDS read it can see.  ad = {row, cols, xrow, xcol}.
template <int OFF, bool WAIT>
__device__ __forceinline__ void h16_read(const unsigned (&ad)[4], v2f& row, float4& c01, float4& c23, v2f& xrow, v2f& xcol) {
    if constexpr (WAIT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kH16Stages - 1) : "memory");
    asm volatile(
        "ds_read_b64 %0, %5 offset:%9\n\t"
        "ds_read_b128 %1, %6 offset:%9\n\t"
        "ds_read_b128 %2, %6 offset:%10\n\t"
        "ds_read_b64 %3, %7 offset:%9\n\t"
        "ds_read_b64 %4, %8 offset:%9\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(row), "=&v"(c01), "=&v"(c23), "=&v"(xrow), "=&v"(xcol)
        : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "n"(OFF), "n"(OFF + 16)
        : "memory");
}

// One frame of a wave: operands from LDS, products of the lane's five entries, one packed FMA per entry and source.
// w: the frame's weights, requested during the previous frame; wn: the next frame's, requested here between the operand
// reads and the arithmetic from its row wpn of the weight table (wave-uniform scalar loads; UNIT: wn = (w1n, 0)).
template <int NP, bool UNIT, int OFF, bool WAIT>
__device__ __forceinline__ void h16_frame(v2f (&acc)[2 * NP][kH16Entries], const unsigned (&ad)[4], const v2f (&w)[NP], v2f (&wn)[NP],
                                          const float* wpn, float w1n) {
    v2f row, xrow, xcol;
    float4 c01, c23;
    h16_read<OFF, WAIT>(ad, row, c01, c23, xrow, xcol);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (UNIT) {
        wn[0] = v2f{w1n, 0.f};
    } else {
#pragma unroll
        for (int q = 0; q < NP; ++q) wn[q] = v2f{wpn[2 * q], wpn[2 * q + 1]};
    }
    __builtin_amdgcn_sched_barrier(0);
    const v2f a[kH16Entries] = {row, row, row, row, xrow};
    const v2f b[kH16Entries] = {v2f{c01.x, c01.y}, v2f{c01.z, c01.w}, v2f{c23.x, c23.y}, v2f{c23.z, c23.w}, xcol};
    v2f p[kH16Entries];
#pragma unroll
    for (int i = 0; i < kH16Entries; ++i) p[i] = qk_mul_lo_negim(a[i], b[i]);
#pragma unroll
    for (int i = 0; i < kH16Entries; ++i) qk_fma_hi_swap(a[i], b[i], p[i]);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
#pragma unroll
        for (int i = 0; i < kH16Entries; ++i) qk_fma_w0(w[q], p[i], acc[2 * q][i]);
        if (!UNIT) {
#pragma unroll
            for (int i = 0; i < kH16Entries; ++i) qk_fma_w1(w[q], p[i], acc[2 * q + 1][i]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);      // keep the next frame's operand reads behind this frame's arithmetic (registers)
}

// NP: source PAIRS per pass (8: up to 16 sources, 6: up to 12, 4: up to 8, 1: the unit-weight pass)
template <int NP, bool UNIT>
__global__ __launch_bounds__(kBlock, 2) void cov_half16_kernel(const float2* __restrict__ X, const float* __restrict__ Wt,
                                                               double* __restrict__ Vpart, int T, int F, int M, int Mv, int K, int tc) {
    constexpr int kRingBytes = kWaves * kH16Stages * kH16Stage;
    constexpr int kScratchBytes = (int)sizeof(float) * kH16Chunk * kH16LdsStride;
    __shared__ float4 ring[(kRingBytes > kScratchBytes ? kRingBytes : kScratchBytes) / 16 + 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;                            // bin of the pair
    const int e = lane & 31;
    const int f0 = blockIdx.x * 2;
    const int k0 = blockIdx.z * 2 * NP;
    const int t_begin = blockIdx.y * tc;
    const int t_end = min(T, t_begin + tc);
    const int nstages = (t_end - t_begin + 4 * kH16Frames - 1) / (4 * kH16Frames);

    v2f acc[2 * NP][kH16Entries];
#pragma unroll
    for (int s = 0; s < 2 * NP; ++s)
#pragma unroll
        for (int i = 0; i < kH16Entries; ++i) acc[s][i] = v2f{0.f, 0.f};

    // ---- DMA side: lane l of a stage's instruction moves 16-byte piece l & 15 of frame (l >> 4) of the stage: the wave's
    //      frames are t_begin + wave + 4 n, a stage holds n = 4 i .. 4 i + 3.  Pieces past the run (fewer than 16 channels,
    //      one bin left) re-request its last piece; frames past the tensor re-request frame T - 1 (their weights are 0).
    char* wring = reinterpret_cast<char*>(ring) + wave * (kH16Stages * kH16Stage);       // wave-uniform
    const int run_pieces = min(2, F - f0) * M / 2;
    const unsigned piece_off = (unsigned)min(lane & 15, run_pieces - 1) * 16u;
    const int lane_frame = 4 * (lane >> 4);
    const char* xbytes = reinterpret_cast<const char*>(X);
    const size_t row_bytes = (size_t)F * M * 8;
    const char* run0 = xbytes + (size_t)f0 * M * 8 + piece_off;
    auto issue = [&](int i, int s) {
        const int t = t_begin + wave + 4 * kH16Frames * i + lane_frame;
        const int tcl = min(i < nstages ? t : T - 1, T - 1);
        __builtin_amdgcn_global_load_lds((gvoid_t*)(run0 + (size_t)tcl * row_bytes), (lvoid_t*)(wring + s * kH16Stage), 16, 0, 0);
    };

    // ---- operand addresses of this lane inside a frame slot (bin h at h * M * 8; reads of channels >= M stay inside the
    //      256-byte slot and only reach entries that are dropped)
    int ci0, di0, cix, dix;
    h16_entry(e, 0, &ci0, &di0);
    h16_entry(e, 4, &cix, &dix);
    const unsigned lbase = (unsigned)(uintptr_t)wring + (unsigned)(h * M * 8);
    const unsigned ad[4] = {lbase + 8u * ci0, lbase + 8u * di0, lbase + 8u * cix, lbase + 8u * dix};

    // frame u of stage i of this wave: its row of the weight table (frames past the split: the zeroed row T)
    auto wrow = [&](int i, int u) -> const float* {
        const int t = t_begin + wave + 4 * (kH16Frames * i + u);
        return Wt + (size_t)(t < t_end ? t : T) * kH16WeightStride + k0;
    };
    auto wunit = [&](int i, int u) { return t_begin + wave + 4 * (kH16Frames * i + u) < t_end ? 1.f : 0.f; };
    // frame (I, U) uses the weights requested during the frame before it and requests those of the frame after it
    v2f wa[NP], wb[NP];
#define OIVA_H16_FRAME(I, S, U, W, WN, IN, UN)                                                                        \
    h16_frame<NP, UNIT, (S) * kH16Stage + (U) * kH16Slot, (U) == 0>(acc, ad, W, WN, UNIT ? nullptr : wrow(IN, UN), wunit(IN, UN));
#define OIVA_H16_STAGE(I, S)                  \
    OIVA_H16_FRAME(I, S, 0, wa, wb, I, 1)     \
    OIVA_H16_FRAME(I, S, 1, wb, wa, I, 2)     \
    OIVA_H16_FRAME(I, S, 2, wa, wb, I, 3)     \
    OIVA_H16_FRAME(I, S, 3, wb, wa, (I) + 1, 0)
    static_assert(kH16Stages == 4 && kH16Frames == 4, "the loop below is unrolled for a 4-stage ring of 4 frames");

    issue(0, 0);
    issue(1, 1);
    issue(2, 2);
    if constexpr (UNIT) {
        wa[0] = v2f{wunit(0, 0), 0.f};
    } else {
        const float* wp0 = wrow(0, 0);
#pragma unroll
        for (int q = 0; q < NP; ++q) wa[q] = v2f{wp0[2 * q], wp0[2 * q + 1]};
    }
    int i = 0;
    for (; i + 4 <= nstages; i += 4) {      // stage indices are compile-time constants in the unrolled body
        issue(i + 3, 3); OIVA_H16_STAGE(i, 0)
        issue(i + 4, 0); OIVA_H16_STAGE(i + 1, 1)
        issue(i + 5, 1); OIVA_H16_STAGE(i + 2, 2)
        issue(i + 6, 2); OIVA_H16_STAGE(i + 3, 3)
    }
    if (i < nstages) { issue(i + 3, 3); OIVA_H16_STAGE(i, 0) }
    if (i + 1 < nstages) { issue(i + 4, 0); OIVA_H16_STAGE(i + 1, 1) }
    if (i + 2 < nstages) { issue(i + 5, 1); OIVA_H16_STAGE(i + 2, 2) }
#undef OIVA_H16_STAGE
#undef OIVA_H16_FRAME
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the DMA queue before the ring becomes reduction scratch

    // ---- the four waves (frame phases) of the workgroup added in float64, fixed order; one packed partial per
    //      (frame split, bin, source).  accumulator n = source * 10 + entry * 2 + (re | im)
    float* lds = reinterpret_cast<float*>(ring);
    constexpr int NACC = 2 * NP * kH16Entries * 2;
    const int NA = Mv * Mv;            // Mv <= M: the matrix that is stored (M: channel pitch of X)
    // thread tid sums accumulator wave + 4 v (+ r0) of LANE `lane` in every round: the packed positions of that lane's five
    // entries are computed once (-1: dropped -- below the diagonal of a diagonal block, or a channel >= M)
    int pos[kH16Entries];
    bool offdiag[kH16Entries];
#pragma unroll
    for (int ent = 0; ent < kH16Entries; ++ent) {
        int ci, di;
        h16_entry(e, ent, &ci, &di);
        offdiag[ent] = ci < di;
        pos[ent] = (di < Mv && ci <= di) ? (ci == di ? ci : herm_pair_index(Mv, ci, di)) : -1;
    }
    const int fo = f0 + h;
    double* const vb = Vpart + (((size_t)blockIdx.y * F + (fo < F ? fo : F - 1)) * K + k0) * NA;
#pragma unroll
    for (int r0 = 0; r0 < NACC; r0 += kH16Chunk) {
        __syncthreads();
#pragma unroll
        for (int a = 0; a < kH16Chunk; ++a) {
            const int n = r0 + a;       // compile-time
            const v2f v = acc[n / (2 * kH16Entries)][(n % (2 * kH16Entries)) / 2];
            lds[a * kH16LdsStride + tid] = (n & 1) ? v.y : v.x;
        }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < kH16Chunk * 64 / kBlock; ++v) {
            const int aa = wave + 4 * v;            // == (tid + kBlock * v) >> 6; the lane is this thread's own
            double s = 0.;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) s += (double)lds[aa * kH16LdsStride + w * 64 + lane];
            const int n = r0 + aa;
            const int src = n / (2 * kH16Entries), rem = n % (2 * kH16Entries), ent = rem >> 1, im = rem & 1;
            int pe = pos[0];
            bool od = offdiag[0];
#pragma unroll
            for (int x = 1; x < kH16Entries; ++x) {
                pe = ent == x ? pos[x] : pe;
                od = ent == x ? offdiag[x] : od;
            }
            const bool keep = pe >= 0 && (im == 0 || od) && fo < F && k0 + src < K && (!UNIT || src == 0);
            if (keep) vb[(size_t)src * NA + pe + im] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// float64 form (the `precise` arithmetic, OIVA_PREC_COV_F64) of the same decomposition for 3..16 sources: float64 sums of
// exact float64 products, FOUR or EIGHT sources per pass (40 / 80 float64 accumulators per lane).  The fp64 matrix-core
// kernel it replaces there (kernels_cov_mfma.hip) spends 768 multiply-adds per frame and source on a pipe that sustains
// 44 TFLOP/s; this one 14 conversions + 20 + 10 NS float64 vector instructions per lane, frame and pass at 60 TFLOP/s peak.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kH64WeightStride = 16;                    // doubles per row of the float64 weight table
constexpr int kH64Chunk = kH16Chunk / 2;                // doubles per LDS round of the epilogue (same scratch bytes)

// Final float64 weights: Wt[t][k] = 1 / max(r[t,k] / gamma_k, eps) (overiva.py:158-173), columns >= K and row T zero
__global__ __launch_bounds__(kBlock) void h64_weights_kernel(const float* __restrict__ R, double* __restrict__ Wt, float* __restrict__ wscale,
                                                             int model, int raw, int T, int K) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= (T + 1) * kH64WeightStride) return;
    const int t = e / kH64WeightStride, k = e % kH64WeightStride;
    double w = 0.;
    if (t < T && k < K) {
        const double gamma = (raw & 1) ? 1. : gamma_of(R, T, K, k);
        double rn = (double)R[(size_t)t * K + k] / gamma;
        rn = rn < (double)kEpsR ? (double)kEpsR : rn;          // a NaN stays NaN, like r[r < eps] = eps in the reference
        w = 1. / rn;
        if (t == 0 && wscale != nullptr && !(raw & 1))
            wscale[k] = model == OIVA_MODEL_LAPLACE ? (float)gamma : (float)sqrt(gamma);   // overiva.py:163 / :167
    }
    Wt[e] = w;
}

template <int NS, int OFF, bool WAIT>
__device__ __forceinline__ void h64_frame(double (&accr)[NS][kH16Entries], double (&acci)[NS][kH16Entries],
                                          const unsigned (&ad)[4], const double (&w)[NS], double (&wn)[NS],
                                          const double* wpn) {
    v2f row, xrow, xcol;
    float4 c01, c23;
    h16_read<OFF, WAIT>(ad, row, c01, c23, xrow, xcol);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NS; ++q) wn[q] = wpn[q];          // the next frame's weights: wave-uniform scalar loads
    __builtin_amdgcn_sched_barrier(0);
    const double ar[kH16Entries] = {row.x, row.x, row.x, row.x, xrow.x}, ai[kH16Entries] = {row.y, row.y, row.y, row.y, xrow.y};
    const double br[kH16Entries] = {c01.x, c01.z, c23.x, c23.z, xcol.x}, bi[kH16Entries] = {c01.y, c01.w, c23.y, c23.w, xcol.y};
#pragma unroll
    for (int i = 0; i < kH16Entries; ++i) {
        const double pre = fma(ar[i], br[i], ai[i] * bi[i]);       // x_c conj(x_d)
        const double pim = fma(ai[i], br[i], -(ar[i] * bi[i]));
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            accr[q][i] = fma(w[q], pre, accr[q][i]);
            acci[q][i] = fma(w[q], pim, acci[q][i]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int NS>
__global__ __launch_bounds__(kBlock, NS == 8 ? 2 : 4) void cov_half16f64_kernel(const float2* __restrict__ X, const double* __restrict__ Wt,
                                                                  double* __restrict__ Vpart, int T, int F, int M, int Mv, int K, int tc) {
    constexpr int kRingBytes = kWaves * kH16Stages * kH16Stage;
    constexpr int kScratchBytes = (int)sizeof(double) * kH64Chunk * kH16LdsStride;
    __shared__ float4 ring[(kRingBytes > kScratchBytes ? kRingBytes : kScratchBytes) / 16 + 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int e = lane & 31;
    const int f0 = blockIdx.x * 2;
    const int k0 = blockIdx.z * NS;
    const int t_begin = blockIdx.y * tc;
    const int t_end = min(T, t_begin + tc);
    const int nstages = (t_end - t_begin + 4 * kH16Frames - 1) / (4 * kH16Frames);

    double accr[NS][kH16Entries], acci[NS][kH16Entries];
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int i = 0; i < kH16Entries; ++i) accr[q][i] = acci[q][i] = 0.;

    char* wring = reinterpret_cast<char*>(ring) + wave * (kH16Stages * kH16Stage);       // wave-uniform
    const int run_pieces = min(2, F - f0) * M / 2;
    const unsigned piece_off = (unsigned)min(lane & 15, run_pieces - 1) * 16u;
    const int lane_frame = 4 * (lane >> 4);
    const char* xbytes = reinterpret_cast<const char*>(X);
    const size_t row_bytes = (size_t)F * M * 8;
    const char* run0 = xbytes + (size_t)f0 * M * 8 + piece_off;
    auto issue = [&](int i, int s) {
        const int t = t_begin + wave + 4 * kH16Frames * i + lane_frame;
        const int tcl = min(i < nstages ? t : T - 1, T - 1);
        __builtin_amdgcn_global_load_lds((gvoid_t*)(run0 + (size_t)tcl * row_bytes), (lvoid_t*)(wring + s * kH16Stage), 16, 0, 0);
    };
    int ci0, di0, cix, dix;
    h16_entry(e, 0, &ci0, &di0);
    h16_entry(e, 4, &cix, &dix);
    const unsigned lbase = (unsigned)(uintptr_t)wring + (unsigned)(h * M * 8);
    const unsigned ad[4] = {lbase + 8u * ci0, lbase + 8u * di0, lbase + 8u * cix, lbase + 8u * dix};

    auto wrow = [&](int i, int u) -> const double* {      // frames past the split: the zeroed row T
        const int t = t_begin + wave + 4 * (kH16Frames * i + u);
        return Wt + (size_t)(t < t_end ? t : T) * kH64WeightStride + k0;
    };
    double wa[NS], wb[NS];
#define OIVA_H64_FRAME(I, S, U, W, WN, IN, UN) \
    h64_frame<NS, (S) * kH16Stage + (U) * kH16Slot, (U) == 0>(accr, acci, ad, W, WN, wrow(IN, UN));
#define OIVA_H64_STAGE(I, S)                  \
    OIVA_H64_FRAME(I, S, 0, wa, wb, I, 1)     \
    OIVA_H64_FRAME(I, S, 1, wb, wa, I, 2)     \
    OIVA_H64_FRAME(I, S, 2, wa, wb, I, 3)     \
    OIVA_H64_FRAME(I, S, 3, wb, wa, (I) + 1, 0)
    issue(0, 0);
    issue(1, 1);
    issue(2, 2);
    {
        const double* wp0 = wrow(0, 0);
#pragma unroll
        for (int q = 0; q < NS; ++q) wa[q] = wp0[q];
    }
    int i = 0;
    for (; i + 4 <= nstages; i += 4) {
        issue(i + 3, 3); OIVA_H64_STAGE(i, 0)
        issue(i + 4, 0); OIVA_H64_STAGE(i + 1, 1)
        issue(i + 5, 1); OIVA_H64_STAGE(i + 2, 2)
        issue(i + 6, 2); OIVA_H64_STAGE(i + 3, 3)
    }
    if (i < nstages) { issue(i + 3, 3); OIVA_H64_STAGE(i, 0) }
    if (i + 1 < nstages) { issue(i + 4, 0); OIVA_H64_STAGE(i + 1, 1) }
    if (i + 2 < nstages) { issue(i + 5, 1); OIVA_H64_STAGE(i + 2, 2) }
#undef OIVA_H64_STAGE
#undef OIVA_H64_FRAME
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // the four waves added in fixed order, rounds of 8 doubles; accumulator n = source * 10 + entry * 2 + (re | im)
    double* lds = reinterpret_cast<double*>(ring);
    constexpr int NACC = NS * kH16Entries * 2;
    const int NA = Mv * Mv;            // Mv <= M: the matrix that is stored (M: channel pitch of X)
    int pos[kH16Entries];
    bool offdiag[kH16Entries];
#pragma unroll
    for (int ent = 0; ent < kH16Entries; ++ent) {
        int ci, di;
        h16_entry(e, ent, &ci, &di);
        offdiag[ent] = ci < di;
        pos[ent] = (di < Mv && ci <= di) ? (ci == di ? ci : herm_pair_index(Mv, ci, di)) : -1;
    }
    const int fo = f0 + h;
    double* const vb = Vpart + (((size_t)blockIdx.y * F + (fo < F ? fo : F - 1)) * K + k0) * NA;
#pragma unroll
    for (int r0 = 0; r0 < NACC; r0 += kH64Chunk) {
        __syncthreads();
#pragma unroll
        for (int a = 0; a < kH64Chunk; ++a) {
            const int n = r0 + a < NACC ? r0 + a : 0;       // compile-time
            const int q = n / (2 * kH16Entries), ent = (n % (2 * kH16Entries)) / 2;
            lds[a * kH16LdsStride + tid] = (n & 1) ? acci[q][ent] : accr[q][ent];
        }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < kH64Chunk * 64 / kBlock; ++v) {
            const int aa = wave + 4 * v;
            double s = 0.;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) s += lds[aa * kH16LdsStride + w * 64 + lane];
            const int n = r0 + aa;
            const int src = n / (2 * kH16Entries), rem = n % (2 * kH16Entries), ent = rem >> 1, im = rem & 1;
            int pe = pos[0];
            bool od = offdiag[0];
#pragma unroll
            for (int x = 1; x < kH16Entries; ++x) {
                pe = ent == x ? pos[x] : pe;
                od = ent == x ? offdiag[x] : od;
            }
            const bool keep = pe >= 0 && (im == 0 || od) && fo < F && k0 + src < K && n < NACC;
            if (keep) vb[(size_t)src * NA + pe + im] = s;
        }
    }
}

}  // namespace

bool cov_half16_supported(int M, int K) { return M >= 10 && M <= 16 && M % 2 == 0 && K >= 1 && K <= 16; }
int cov_half16_sources_per_pass(int K) { return K <= 8 ? 8 : (K <= 12 ? 12 : 16); }

// Wt: (T + 1, 16) scratch for the final weights (row T zeroed here); R == nullptr: unit weights (K = 1)
hipError_t launch_cov_half16(hipStream_t s, const float2* X, const float* R, float* Wt, float* wscale, int model, int raw,
                             double* Vpart, int T, int F, int M, int Mv, int K, const CovGeom& g) {
    if (!cov_half16_supported(M, K) || Mv > M || Mv < M - 1 || g.tc % (4 * kH16Frames) != 0) return hipErrorInvalidValue;
    const dim3 grid((F + 1) / 2, g.nsplit, 1), block(kBlock);
    if (R == nullptr) {
        if (K != 1) return hipErrorInvalidValue;
        return launch_dominant(cov_half16_kernel<1, true>, grid, block, 0, s, X, (const float*)nullptr, Vpart, T, F, M, Mv, K, g.tc);
    }
    if (Wt == nullptr) return hipErrorInvalidValue;
    hipError_t e = launch_cov_weights(s, R, Wt, wscale, model, raw, T, K, kH16WeightStride);
    if (e == hipSuccess) e = hipMemsetAsync(Wt + (size_t)T * kH16WeightStride, 0, kH16WeightStride * sizeof(float), s);
    if (e != hipSuccess) return e;
    // nine and more sources: the weighted sums of all sources as one small GEMM per bin on the fp32 matrix cores, the
    // Hermitian products on the vector ALU beside it (kernels_cov_hmfma.hip; CovGeom::hmfma, on by default)
    if (g.hmfma && cov_hmfma_supported(M, K)) return launch_cov_hmfma(s, X, Wt, Vpart, T, F, M, Mv, K, g);
    if (K <= 8)
        return launch_dominant(cov_half16_kernel<4, false>, grid, block, 0, s, X, (const float*)Wt, Vpart, T, F, M, Mv, K, g.tc);
    if (K <= 12)
        return launch_dominant(cov_half16_kernel<6, false>, grid, block, 0, s, X, (const float*)Wt, Vpart, T, F, M, Mv, K, g.tc);
    return launch_dominant(cov_half16_kernel<8, false>, grid, block, 0, s, X, (const float*)Wt, Vpart, T, F, M, Mv, K, g.tc);
}

// float64 sums (the `precise` arithmetic): 4 or 8 sources per pass.  Wt: scratch of (T + 1) x 16 doubles.
// (one or two sources stay on the fp64 matrix-core kernel: a two-source instantiation of this one measured 523-590 us against
//  474 at 2048 bins x 4000 frames x 16 channels -- the 34 conversion and product instructions per lane and frame are then
//  two thirds of the work; three or four sources: 692 against 836 us, five to eight 1.06 against 1.57 ms, sixteen 2.08 against 3.10)
bool cov_half16_f64_supported(int M, int K) { return M >= 10 && M <= 16 && M % 2 == 0 && K >= 3 && K <= 16; }
int cov_half16_f64_sources_per_pass(int K) { return K <= 4 ? 4 : 8; }

hipError_t launch_cov_half16_f64(hipStream_t s, const float2* X, const float* R, float* Wt, float* wscale, int model, int raw,
                                 double* Vpart, int T, int F, int M, int Mv, int K, const CovGeom& g) {
    if (!cov_half16_f64_supported(M, K) || Mv > M || Mv < M - 1 || R == nullptr || Wt == nullptr || (!g.hmfma && g.tc % (4 * kH16Frames) != 0)) return hipErrorInvalidValue;
    double* wt = reinterpret_cast<double*>(Wt);
    h64_weights_kernel<<<dim3(((T + 1) * kH64WeightStride + kBlock - 1) / kBlock), dim3(kBlock), 0, s>>>(R, wt, wscale, model, raw, T, K);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (g.hmfma && cov_hmfma64_supported(M, K)) return launch_cov_hmfma64(s, X, wt, Vpart, T, F, M, Mv, K, g);
    const int ns = cov_half16_f64_sources_per_pass(K);
    const dim3 grid((F + 1) / 2, g.nsplit, (K + ns - 1) / ns);
    if (ns == 4) return launch_dominant(cov_half16f64_kernel<4>, grid, dim3(kBlock), 0, s, X, (const double*)wt, Vpart, T, F, M, Mv, K, g.tc);
    return launch_dominant(cov_half16f64_kernel<8>, grid, dim3(kBlock), 0, s, X, (const double*)wt, Vpart, T, F, M, Mv, K, g.tc);
}

}  // namespace oiva
