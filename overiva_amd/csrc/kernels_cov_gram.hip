// Weighted spatial covariance accumulated in float64 on the fp64 matrix cores, for 4 and 8 channels: the
// covariance pass of the "precise" arithmetic (OIVA_PREC_COV_F64).
//
//   V_k[f] = sum_t w_k[t] * x_{t,f} x_{t,f}^H            reference overiva.py:179 (all k in one pass), :87 (w = 1)
//
// The reference's r_inv is float64, which silently promotes this product to complex128 even for complex64 input
// (overiva.py:127-128,179): products of float32 data are exact in float64 and the sum is a float64 chain.  That
// is what v_mfma_f64_16x16x4_f64 computes, in the real Gram form: one frame of the native (T, F, M) complex64
// tensor is a row of F*M*2 floats; a TILE is 16 consecutive floats of that row = the interleaved (re, im) vectors
// of 16/(2M) bins (one bin at M = 8, two at M = 4).  With x~ the 16-vector of a tile,
//       G_k = sum_t w_k[t] x~ x~^T          is one 16x16 accumulator tile with 4 frames as the contraction:
// lane l supplies A[i][kf] = w_k[t+kf] * x~_i[t+kf] and B[kf][j] = x~_j[t+kf] with i = j = l & 15, kf = l >> 4 --
// ONE float per lane feeds both operands.  Per bin  V_re[c][d] = G[2c][2d] + G[2c+1][2d+1],
// V_im[c][d] = G[2c+1][2d] - G[2c][2d+1]  (the off-diagonal blocks of a two-bin tile are not used).
//
// Bound: the fp64 matrix pipe.  8.4 GFLOP (16*16*2 per bin, frame, source) at the headline shape; measured with
// tools/mfmabench.hip the instruction sustains 47 ns per SIMD (44 TFLOP/s chip-wide under its own power limit),
// i.e. >= 190 us, twice the time the 524 MB of X need from HBM.  (The same Gram form on the fp32 matrix cores
// was built and measured too: 116-130 us against 100 us for the vector-ALU kernel of kernels_cov.hip, which
// computes only the Hermitian half; it was dropped.  float64 VALU operations share the matrix pipe -- 8 of them
// cost as much as one MFMA -- so folding fp32 chains into float64 registers is no cheap alternative either.)
//
// Memory: X goes HBM -> LDS with global_load_lds (16 B per lane, no staging registers) into a private ring of
// kStages x 1 KB per wave; a wave owns 4 tiles = 256 contiguous bytes of every frame, one DMA instruction moves
// 4 frames of them (lane -> frame l >> 4, piece l & 15).  No workgroup barrier in the loop, no cross-lane
// reduction: the MFMA contracts over frames and a wave keeps its tiles for the whole frame split.  The only
// ordering is the wave's own vmcnt; the counted wait and the LDS reads sit in asm blocks because hipcc otherwise
// drains the DMA queue (vmcnt(0)) in front of every LDS read it can see.  The weights of the split's frames are
// computed once per workgroup into an LDS table (a frame's weights are one LDS read per stage).
#include <cstdint>
#include <type_traits>

#include "oiva_device.h"

namespace oiva {
namespace {

constexpr int kStages = 8;                 // ring depth per wave: 6 KB in flight while two stages are consumed
constexpr int kTilesPerWave = 4;
constexpr int kTilesPerBlock = kTilesPerWave * kWaves;   // 16 tiles = 1 KB of every frame
constexpr int kStageFloats = 256;          // 4 frames x 4 tiles x 16 floats

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

constexpr int kMaxFrames = 1024;           // frames per workgroup (LDS table of their weights); choose_cov_geom obeys it
constexpr int kTabFrames = kMaxFrames + 8; // + the two stages the software pipeline reads past the end (weight 0)

// LDS -> register traffic of one stage: this lane's float of the stage's 4 tiles and the KC float64 weights of
// its frame (NW dwords).  The reads are ISSUED one stage ahead of their use (stage_fetch) and waited for just
// before it (stage_wait), so the LDS round trip hides behind the previous stage's MFMAs; two explicit register
// sets alternate, nothing is copied between them.
template <int NW>
struct WBits;
template <>
struct WBits<1> { using type = float; };
template <>
struct WBits<2> { using type = double; };
template <>
struct WBits<4> { using type = float4; };
template <int NW>
struct StageRegs {
    float x0, x1, x2, x3;
    typename WBits<NW>::type w;
};

// wait until at most kStages - 2 DMA instructions are outstanding (= the stage to read has landed), issue the reads
template <int NW>
__device__ __forceinline__ void stage_fetch(unsigned xaddr, unsigned waddr, StageRegs<NW>& r) {
#define OIVA_FETCH(WREAD)                                                                                 \
    asm volatile("s_waitcnt vmcnt(%7)\n\t"                                                                \
                 "ds_read_b32 %0, %5\n\t"                                                                 \
                 "ds_read_b32 %1, %5 offset:64\n\t"                                                       \
                 "ds_read_b32 %2, %5 offset:128\n\t"                                                      \
                 "ds_read_b32 %3, %5 offset:192\n\t" WREAD " %4, %6"                                      \
                 : "=&v"(r.x0), "=&v"(r.x1), "=&v"(r.x2), "=&v"(r.x3), "=&v"(r.w)                         \
                 : "v"(xaddr), "v"(waddr), "n"(kStages - 2)                                               \
                 : "memory")
    if constexpr (NW == 1)
        OIVA_FETCH("ds_read_b32");
    else if constexpr (NW == 2)
        OIVA_FETCH("ds_read_b64");
    else
        OIVA_FETCH("ds_read_b128");
#undef OIVA_FETCH
}
// the registers of a fetched stage may be used after this (they are tied through it so that no use moves above)
template <int NW>
__device__ __forceinline__ void stage_wait(StageRegs<NW>& r) {
    if constexpr (NW == 4)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(r.x0), "+v"(r.x1), "+v"(r.x2), "+v"(r.x3), "+v"(r.w.x), "+v"(r.w.y), "+v"(r.w.z), "+v"(r.w.w)::"memory");
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.x0), "+v"(r.x1), "+v"(r.x2), "+v"(r.x3), "+v"(r.w)::"memory");
}

// KC: sources per pass over X (1 or 2: a frame's KC float64 weights are one LDS read)
template <int KC>
__global__ __launch_bounds__(kBlock, 3) void cov_gram_kernel(const float* __restrict__ Xf, const float* __restrict__ R,
                                                             float* __restrict__ wscale, int model, int raw,
                                                             double* __restrict__ Vpart, int T, int F, int M, int K, int tc) {
    using REAL = double;
    using acc_t = Mfma<double>::acc_t;
    constexpr int NW = KC * 2;
    static_assert(NW == 2 || NW == 4, "a frame's weights are one LDS read");
    __shared__ __attribute__((aligned(16))) float ring[kWaves][kStages][kStageFloats];
    __shared__ __attribute__((aligned(16))) double wtab[kTabFrames * KC];   // weights of this workgroup's frames

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kf = lane >> 4;            // frame within a stage = contraction index of the MFMA
    const int k0 = blockIdx.z * KC;
    const int t_begin = blockIdx.y * tc;
    const int t_end = min(T, t_begin + tc);
    const int nstages = (t_end - t_begin + 3) >> 2;
    const bool unit = R == nullptr;

    const long long row_floats = (long long)F * M * 2;                  // one frame
    const long long tile0 = ((long long)blockIdx.x * kWaves + wave) * kTilesPerWave;
    // this lane's 16-byte piece of the wave's 256-byte run; pieces past the end of the frame (last, partial
    // group of tiles) are redirected to the start of the frame: legal address, lands in tiles that are never stored
    long long piece = tile0 * 16 + (lane & 15) * 4;
    if (piece + 4 > row_floats) piece = 0;
    const float* src0 = Xf + piece;

    float* wring = &ring[wave][0][0];                                    // wave-uniform
    const unsigned rd_base = (unsigned)(uintptr_t)wring + (unsigned)(kf * 256 + (lane & 15) * 4);
    const unsigned wt_base = (unsigned)(uintptr_t)wtab + (unsigned)(kf * NW * 4);

    // stage s -> slot s % kStages; frames past the end of the split re-request its last frame (they weigh 0)
    auto dma = [&](int s) {
        const int t = min(t_begin + 4 * s + kf, t_end - 1);
        __builtin_amdgcn_global_load_lds((gvoid_t*)(src0 + (long long)t * row_floats),
                                         (lvoid_t*)(wring + (s & (kStages - 1)) * kStageFloats), 16, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < kStages - 1; ++s) dma(s);

    // while the first stages are on their way: scale normalisation of the activations (overiva.py:158-159) and
    // the table of this split's weights w[t,k] = 1 / max(r[t,k] / gamma_k, eps) (overiva.py:170-173)
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
        const int k = k0 + kk;
        double ginv = 1.0;
        if (!unit && !(raw & 1)) {
            const double gamma = gamma_of(R, T, K, k < K ? k : K - 1);
            ginv = 1.0 / gamma;
            if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && k < K && wscale != nullptr)
                wscale[k] = model == OIVA_MODEL_LAPLACE ? (float)gamma : (float)sqrt(gamma);   // overiva.py:163 / :167
        }
        for (int tl = threadIdx.x; tl < 4 * nstages + 8; tl += kBlock) {
            const int t = t_begin + tl;
            REAL w = 0;
            if (t < t_end && k < K) {
                if (unit) {
                    w = 1;
                } else {
                    double rn = (double)R[(size_t)t * K + k] * ginv;
                    rn = rn < (double)kEpsR ? (double)kEpsR : rn;
                    w = 1.0 / rn;
                }
            }
            wtab[tl * KC + kk] = w;
        }
    }
    __syncthreads();

    acc_t acc[kTilesPerWave][KC];
#pragma unroll
    for (int j = 0; j < kTilesPerWave; ++j)
#pragma unroll
        for (int kk = 0; kk < KC; ++kk)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][kk][r] = 0;

    auto fetch = [&](int s, StageRegs<NW>& r) {
        stage_fetch<NW>(rd_base + (unsigned)((s & (kStages - 1)) * kStageFloats * 4), wt_base + (unsigned)(s * 4 * NW * 4), r);
    };
    auto compute = [&](const StageRegs<NW>& r) {          // the MFMAs of one stage
        const float xs[kTilesPerWave] = {r.x0, r.x1, r.x2, r.x3};
        double w[KC];
        __builtin_memcpy(w, &r.w, sizeof(w));
#pragma unroll
        for (int j = 0; j < kTilesPerWave; ++j) {
            const double b = (double)xs[j];
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) acc[j][kk] = Mfma<double>::run(w[kk] * b, b, acc[j][kk]);
        }
    };

    // two stages per trip, register sets A and B; stages past the end of the split
    // carry weight 0 (wtab) and valid data (dma clamps), so the trip count needs no guards
    StageRegs<NW> ra, rb;
    fetch(0, ra);
    stage_wait<NW>(ra);
    for (int s = 0; s < nstages; s += 2) {
        dma(s + kStages - 1);
        fetch(s + 1, rb);
        compute(ra);
        stage_wait<NW>(rb);
        dma(s + kStages);
        fetch(s + 2, ra);
        compute(rb);
        stage_wait<NW>(ra);
    }
    // drain the DMA queue (the trailing re-requests) before the ring is reused as unpacking scratch
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // G -> packed Hermitian V, through this wave's own ring (2 KB of the 8): the tile is laid down as
    // [row][col] float64 whatever the accumulator layout of the instruction was, then lane e of the first
    // 8*M lanes gathers entry (bin b, c, d) of the tile's 16/(2M) bins.
    double* tile = reinterpret_cast<double*>(wring);                     // [16][16]
    const int M2 = 2 * M;
    const int NA = M * M;
    const int bins_per_tile = 16 / M2;
    const int ent = lane;                                                // entry of the tile: (b, c, d)
    const int eb = ent / NA, ec = (ent - eb * NA) / M, ed = ent % M;
    const bool ent_ok = ent < bins_per_tile * NA && ec <= ed;
#pragma unroll
    for (int j = 0; j < kTilesPerWave; ++j) {
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                tile[Mfma<double>::row(lane, r) * 16 + (lane & 15)] = acc[j][kk][r];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                          // lgkmcnt(0): the wave's own LDS writes
            __builtin_amdgcn_wave_barrier();
            const long long bin = (tile0 + j) * bins_per_tile + eb;
            const int k = k0 + kk;
            if (ent_ok && bin < F && k < K) {
                const int r0 = eb * M2 + 2 * ec, c0 = eb * M2 + 2 * ed;
                double* out = Vpart + (((size_t)blockIdx.y * F + bin) * K + k) * NA;
                const double re = tile[r0 * 16 + c0] + tile[(r0 + 1) * 16 + c0 + 1];
                if (ec == ed) {
                    out[ec] = re;
                } else {
                    const int o = herm_pair_index(M, ec, ed);
                    out[o] = re;
                    out[o + 1] = tile[(r0 + 1) * 16 + c0] - tile[r0 * 16 + c0 + 1];
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int KC>
hipError_t launch_one(hipStream_t s, const float2* X, const float* R, float* wscale, int model, int raw, void* Vpart,
                      int T, int F, int M, int K, const CovGeom& g) {
    const long long tiles = ((long long)F * M * 2 + 15) / 16;
    dim3 grid((unsigned)((tiles + kTilesPerBlock - 1) / kTilesPerBlock), g.nsplit, (K + KC - 1) / KC);
    return launch_dominant(cov_gram_kernel<KC>, grid, dim3(kBlock), 0, s, reinterpret_cast<const float*>(X), R, wscale, model,
                           raw, static_cast<double*>(Vpart), T, F, M, K, g.tc);
}

}  // namespace

bool cov_gram_supported(int M) { return M == 4 || M == 8; }

int cov_gram_sources_per_pass(int K) { return K >= 2 ? 2 : 1; }
int cov_gram_max_frames() { return kMaxFrames; }

hipError_t launch_cov_gram(hipStream_t s, const float2* X, const float* R, float* wscale, int model, int raw, void* Vpart,
                           int T, int F, int M, int K, const CovGeom& g) {
    if (!cov_gram_supported(M)) return hipErrorInvalidValue;
    if (R == nullptr || g.kc == 1) return launch_one<1>(s, X, R, wscale, model, raw, Vpart, T, F, M, K, g);
    if (g.kc == 2) return launch_one<2>(s, X, R, wscale, model, raw, Vpart, T, F, M, K, g);
    return hipErrorInvalidValue;
}

hipError_t cov_gram_blocks_per_cu(int kc, int* n) {
    if (kc == 1) return hipOccupancyMaxActiveBlocksPerMultiprocessor(n, cov_gram_kernel<1>, kBlock, 0);
    if (kc == 2) return hipOccupancyMaxActiveBlocksPerMultiprocessor(n, cov_gram_kernel<2>, kBlock, 0);
    return hipErrorInvalidValue;
}

}  // namespace oiva
