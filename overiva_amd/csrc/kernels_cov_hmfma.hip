// Weighted spatial covariance pass for 10..16 channels and MANY sources (9..16: BASELINE configs[4] is the determined
// 16 x 16 case) with the SOURCES on the matrix cores.
//
//   V_k[f] = sum_t rinv[t,k] * x_{t,f} x_{t,f}^H          reference overiva.py:179, all K sources in one pass over X
//
// Per (bin, frame) the Hermitian half of x x^H is 256 real numbers (16 x 16: the upper triangle holds the real parts, the
// strict lower triangle the imaginary parts), and the weighted sums of ALL sources are one small GEMM per bin:
//       V[k][e] = sum_t w[k][t] * H[t][e]        k: 16 sources, e: 256 packed entries, t: frames
// i.e. per 4 frames 16 instructions v_mfma_f32_16x16x4_f32 -- A = the weights (16 sources x 4 frames), B = one column j of
// H for those 4 frames (4 frames x 16 rows i), D = 16 sources x 16 rows -- for all 16 sources together, where the planar
// matrix-core kernel (kernels_cov_mfma.hip: rank-1 updates of full real 16 x 16 tiles, 3 instructions per 4 frames and
// SOURCE) issues 48 and the vector-ALU kernel (kernels_cov_half16.hip) spends one packed FMA per entry and source.  The
// products H are formed on the vector ALU (4 instructions per lane and column: two selects by "row <= column", a multiply,
// an FMA) while the matrix pipe runs: 128 vector instructions against 32 matrix instructions (1 024 matrix-pipe cycles)
// per wave and stage.  The fp32 matrix instruction is an exact fmaf chain, so the arithmetic class is that of the
// vector-ALU kernel: float32 products, float32 chains of T / (4 nsplit) frames, float64 sums across waves and splits.
//
// Geometry and memory exactly as kernels_cov_half16.hip: a workgroup = 2 bins x 4 frame phases (waves), one
// global_load_lds per wave and stage moves 4 frames x 256 bytes into a 4-stage ring; the 64 weights of a stage (4 frames x
// 16 sources) ride the same ring by a second, 4-byte DMA whose lane l lands exactly where lane l reads its A operand.
#include <cstdint>

#include "oiva_device.h"

namespace oiva {
namespace {

constexpr int kHmStages = 4;
constexpr int kHmFrames = 4;                            // frames per stage of a wave = the contraction of one MFMA
constexpr int kHmSlot = 256;                            // bytes of 2 bins x (<= 16) channels of one frame
constexpr int kHmX = kHmFrames * kHmSlot;               // 1 KB of X per stage per wave
constexpr int kHmStage = kHmX + 256;                    // + 64 weights
constexpr int kHmChunk = 16;
constexpr int kHmLdsStride = kBlock + 1;
constexpr int kHmWeightStride = 16;                     // row stride of the weight table (launch_cov_weights)

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

// The lane's operands of one bin of one stage: its own row channel, the 16 column channels (8 x 16 bytes, the same addresses
// in the 16 lanes of a frame: broadcast reads) and -- with the first bin -- its weight, with the counted wait for the stage's
// two DMAs in front.  asm: hipcc drains the whole DMA queue in front of any LDS read it can see.  The reads are ISSUED by
// one statement and WAITED for by another (which names every destination), so that the second bin's reads fly while the
// first bin's matrix instructions issue.
struct HmOps {
    float2 row;
    float4 c[8];
};
template <bool FIRST>
__device__ __forceinline__ void hm_read_issue(unsigned a_row, unsigned a_cols, unsigned a_w, HmOps& o, float& w) {
    if constexpr (FIRST) {
        asm volatile(
            "s_waitcnt vmcnt(%11)\n\t"
            "ds_read_b32 %9, %12\n\t"
            "ds_read_b64 %0, %10\n\t"
            "ds_read_b128 %1, %13\n\t"
            "ds_read_b128 %2, %13 offset:16\n\t"
            "ds_read_b128 %3, %13 offset:32\n\t"
            "ds_read_b128 %4, %13 offset:48\n\t"
            "ds_read_b128 %5, %13 offset:64\n\t"
            "ds_read_b128 %6, %13 offset:80\n\t"
            "ds_read_b128 %7, %13 offset:96\n\t"
            "ds_read_b128 %8, %13 offset:112"
            : "=&v"(o.row), "=&v"(o.c[0]), "=&v"(o.c[1]), "=&v"(o.c[2]), "=&v"(o.c[3]), "=&v"(o.c[4]), "=&v"(o.c[5]), "=&v"(o.c[6]), "=&v"(o.c[7]),
              "=&v"(w)
            : "v"(a_row), "n"(2 * (kHmStages - 1)), "v"(a_w), "v"(a_cols)
            : "memory");
    } else {
        asm volatile(
            "ds_read_b64 %0, %9\n\t"
            "ds_read_b128 %1, %10\n\t"
            "ds_read_b128 %2, %10 offset:16\n\t"
            "ds_read_b128 %3, %10 offset:32\n\t"
            "ds_read_b128 %4, %10 offset:48\n\t"
            "ds_read_b128 %5, %10 offset:64\n\t"
            "ds_read_b128 %6, %10 offset:80\n\t"
            "ds_read_b128 %7, %10 offset:96\n\t"
            "ds_read_b128 %8, %10 offset:112"
            : "=&v"(o.row), "=&v"(o.c[0]), "=&v"(o.c[1]), "=&v"(o.c[2]), "=&v"(o.c[3]), "=&v"(o.c[4]), "=&v"(o.c[5]), "=&v"(o.c[6]), "=&v"(o.c[7])
            : "v"(a_row), "v"(a_cols)
            : "memory");
    }
}
// (the wait names no register -- tied operands of struct members are not supported -- so a scheduling barrier behind it keeps
//  every consumer of the destinations below it)
__device__ __forceinline__ void hm_read_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(kBlock, 2) void cov_hmfma_kernel(const float2* __restrict__ X, const float* __restrict__ Wt,
                                                              double* __restrict__ Vpart, int T, int F, int M, int Mv, int K, int tc) {
    constexpr int kRingBytes = kWaves * kHmStages * kHmStage;
    constexpr int kScratchBytes = (int)sizeof(float) * kHmChunk * kHmLdsStride;
    __shared__ float4 ring[(kRingBytes > kScratchBytes ? kRingBytes : kScratchBytes) / 16 + 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4;                            // frame of the stage (contraction index)
    const int n = lane & 15;                            // A: source | B: row channel i
    const int f0 = blockIdx.x * 2;
    const int t_begin = blockIdx.y * tc;
    const int t_end = min(T, t_begin + tc);
    const int nstages = (t_end - t_begin + 4 * kHmFrames - 1) / (4 * kHmFrames);

    f32x4 acc[2][16];                                   // [bin of the pair][column channel j]: sources 4 q + r, row channel n
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[h][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- DMA side (as kernels_cov_half16.hip): lane l moves 16-byte piece l & 15 of frame l >> 4 of the stage; the wave's
    //      frames are t_begin + wave + 4 n, a stage holds n = 4 i .. 4 i + 3.  The weights of the same 4 frames: lane l moves
    //      Wt[frame l >> 4][source l & 15] (frames past the split: the zeroed row T of the table).
    char* wring = reinterpret_cast<char*>(ring) + wave * (kHmStages * kHmStage);       // wave-uniform
    const int run_pieces = min(2, F - f0) * M / 2;
    const unsigned piece_off = (unsigned)min(lane & 15, run_pieces - 1) * 16u;
    const char* xbytes = reinterpret_cast<const char*>(X);
    const size_t row_bytes = (size_t)F * M * 8;
    const char* run0 = xbytes + (size_t)f0 * M * 8 + piece_off;
    auto issue = [&](int i, int s) {
        const int t = t_begin + wave + 4 * (kHmFrames * i + q);
        const int tcl = min(i < nstages ? t : T - 1, T - 1);
        __builtin_amdgcn_global_load_lds((gvoid_t*)(run0 + (size_t)tcl * row_bytes), (lvoid_t*)(wring + s * kHmStage), 16, 0, 0);
        const int tw = (i < nstages && t < t_end) ? t : T;
        __builtin_amdgcn_global_load_lds((gvoid_t*)(Wt + (size_t)tw * kHmWeightStride + n), (lvoid_t*)(wring + s * kHmStage + kHmX), 4, 0, 0);
    };

    // ---- operand addresses of this lane inside stage 0 (bin h at + h * M * 8; reads past M channels stay inside the slot
    //      and only reach entries that are dropped)
    const unsigned lbase = (unsigned)(uintptr_t)wring + (unsigned)(q * kHmSlot);
    const unsigned a_row0 = lbase + 8u * (unsigned)(n < M ? n : 0);
    const unsigned a_w0 = (unsigned)(uintptr_t)wring + (unsigned)kHmX + 4u * (unsigned)lane;
    const unsigned binoff = (unsigned)(M * 8);

    // the 16 column channels of one bin: per column j the product of the lane's row n with it -- Re(x_n conj x_j) for n <= j,
    // Im(x_j conj x_n) below the diagonal, each rounded exactly as kernels_cov_half16.hip rounds it (a product, then an FMA
    // onto it), so that the two kernels give the same bits -- and one MFMA with the stage's weights
    auto columns = [&](const HmOps& o, float w, f32x4 (&a)[16]) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < M) {                                      // (wave-uniform)
                const float xr = (j & 1) ? o.c[j >> 1].z : o.c[j >> 1].x, xi = (j & 1) ? o.c[j >> 1].w : o.c[j >> 1].y;
                const bool up = n <= j;
                // up:    re = fma(row.y, xi, row.x * xr)                      (entry (n, j), a = x_n, b = x_j)
                // below: im = fma(xi, row.x, -(xr * row.y))                   (entry (j, n), a = x_j, b = x_n)
                const float u = up ? o.row.y : o.row.x, s1 = up ? o.row.x : o.row.y, t1 = up ? xr : -xr;
                const float hv = fmaf(u, xi, s1 * t1);
                a[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, hv, a[j], 0, 0, 0);
            }
        }
    };
    // one stage: both bins; the second bin's operands are read while the first bin's matrix instructions issue
    auto stage = [&](int s) {
        const unsigned so = (unsigned)(s * kHmStage);
        float w, wdummy = 0.f;
        HmOps o0, o1;
        hm_read_issue<true>(a_row0 + so, lbase + so, a_w0 + so, o0, w);
        hm_read_wait();
        hm_read_issue<false>(a_row0 + so + binoff, lbase + so + binoff, 0u, o1, wdummy);
        columns(o0, w, acc[0]);
        hm_read_wait();
        columns(o1, w, acc[1]);
    };

    issue(0, 0);
    issue(1, 1);
    issue(2, 2);
    for (int i = 0; i < nstages; ++i) {
        issue(i + 3, (i + 3) & 3);
        stage(i & 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the DMA queue before the ring becomes reduction scratch

    // ---- the four waves (frame phases) added in float64, fixed order; accumulator m = (h * 16 + j) * 4 + r of lane (q, n)
    //      is source 4 q + r, entry (row n, column j) of bin h: packed position n <= j ? re(n, j) : im(j, n)
    float* lds = reinterpret_cast<float*>(ring);
    const int NA = Mv * Mv;
    constexpr int NACC = 2 * 16 * 4;
#pragma unroll
    for (int r0 = 0; r0 < NACC; r0 += kHmChunk) {
        __syncthreads();
#pragma unroll
        for (int a = 0; a < kHmChunk; ++a) {
            const int m = r0 + a;       // compile-time
            lds[a * kHmLdsStride + tid] = acc[m / 64][(m / 4) % 16][m % 4];
        }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < kHmChunk * 64 / kBlock; ++v) {
            const int aa = wave + 4 * v;            // the lane is this thread's own
            double s = 0.;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) s += (double)lds[aa * kHmLdsStride + w * 64 + lane];
            const int m = r0 + aa;
            const int h = m / 64, j = (m / 4) % 16, r = m % 4;
            const int src = 4 * q + r, fo = f0 + h;
            if (fo < F && src < K && n < Mv && j < Mv) {
                const int pos = n == j ? n : (n < j ? herm_pair_index(Mv, n, j) : herm_pair_index(Mv, j, n) + 1);
                Vpart[(((size_t)blockIdx.y * F + fo) * K + src) * NA + pos] = s;
            }
        }
    }
}

}  // namespace

// 9..16 sources on 10/12/14/16 channels (odd counts: the padded copy of X).  Wt: the (T + 1, 16) table of final weights of
// launch_cov_weights, row T zeroed (kernels_cov_half16.hip fills it and calls this).
bool cov_hmfma_supported(int M, int K) { return M >= 10 && M <= 16 && M % 2 == 0 && K >= 9 && K <= 16; }

hipError_t launch_cov_hmfma(hipStream_t s, const float2* X, const float* Wt, double* Vpart, int T, int F, int M, int Mv, int K, const CovGeom& g) {
    if (!cov_hmfma_supported(M, K) || Mv > M || Mv < M - 1 || Wt == nullptr || g.tc % (4 * kHmFrames) != 0) return hipErrorInvalidValue;
    const dim3 grid((F + 1) / 2, g.nsplit, 1), block(kBlock);
    return launch_dominant(cov_hmfma_kernel, grid, block, 0, s, X, Wt, Vpart, T, F, M, Mv, K, g.tc);
}

}  // namespace oiva
