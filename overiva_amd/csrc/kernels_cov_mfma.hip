// Weighted spatial covariance on the matrix cores, for 9..16 channels             reference overiva.py:179 / :87
//
// With M = 16 and K = 16 the pass is 268 GFLOP per iteration (naive complex count) against 1 GB of X: bound by
// the fp32 matrix rate (157 TF), not by HBM.  Planar form on v_mfma_f32_16x16x4_f32 -- channel = row/column,
// 4 frames = contraction, A = w * Re x | w * Im x, B = Re x | Im x as separate operands:
//     V_re      += (w xr) xr^T + (w xi) xi^T        two MFMAs into ONE 16x16 accumulator tile
//     G_ir      += (w xi) xr^T                      one MFMA;  V_im[c][d] = G_ir[c][d] - G_ir[d][c]
// i.e. 3 MFMAs of 32 cycles per 4 frames and source, and 8 accumulator registers per source, so ONE wave carries
// all 16 sources of a bin (128 accumulators) and X is loaded once per bin: lane l loads the complex sample of
// channel l & 15 at frame t + (l >> 4) -- one 8-byte load per 4 frames, 128 contiguous bytes per frame.
// fp32 MFMA is an exact fp32 FMA chain over the frames of a split (<= 512 frames, see choose_cov_geom).
// REAL = double runs the same code on v_mfma_f64_16x16x4_f64 (float64 accumulation mode, 8 sources per wave).
#include "oiva_device.h"

namespace oiva {
namespace {

// Final weights w[t,k] = 1 / max(r[t,k] / gamma_k, eps) (overiva.py:158-173) for the matrix-core kernel, which
// has no VALU slots to spare for the divide: one thread per (frame, padded source column).
// (Round 5, measured and dropped: the table written by the activation kernel itself instead of this launch of 5-7 us -- by its
//  last workgroup: one workgroup for 64 000 divisions, activation stage 4 -> 67-135 us; by the last workgroup of every SOURCE,
//  column k of the table each: strided 4-byte stores and loads from 16 CUs into the same lines, 4 + 5 -> 17 us at 8 channels /
//  2 sources in float64, 7 + 7 -> 50 at 16 / 16, equal (9.7 against 10.0) at 235 frames.  gamma needs every block of a source, so a
//  row-major split would have to wait inside the kernel.)
constexpr int kMaxK = OIVA_MAX_CHANNELS;
__global__ __launch_bounds__(kBlock) void weights_kernel(const float* __restrict__ R, float* __restrict__ Wt,
                                                         float* __restrict__ wscale, int model, int raw, int T, int K,
                                                         int Kp) {
    const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (e >= (long long)T * Kp) return;
    const int t = (int)(e / Kp), k = (int)(e - (long long)t * Kp);
    float w = 0.f;                                      // padding columns: sources that do not exist weigh 0
    if (k < K) {
        const float gamma = (raw & 1) ? 1.f : (float)gamma_of(R, T, K, k);
        w = activation_weight(R[(size_t)t * K + k], 1.f / gamma);
        if (t == 0 && wscale != nullptr && !(raw & 1))
            wscale[k] = model == OIVA_MODEL_LAPLACE ? gamma : sqrtf(gamma);   // overiva.py:163 / :167
    }
    Wt[e] = w;
}

template <typename REAL, int KW, bool UNIT>
__global__ __launch_bounds__(64, 2) void cov_mfma16_kernel(const float2* __restrict__ X, const float* __restrict__ Wt,
                                                           REAL* __restrict__ Vpart, int T, int F, int M, int K, int Kp,
                                                           int tc) {
    using acc_t = typename Mfma<REAL>::acc_t;
    __shared__ REAL tile[2][16][17];    // epilogue: V_re tile and G_ir tile as [row][col]
    const int lane = threadIdx.x;
    const int ch = lane & 15;           // channel: row of A, column of B
    const int kf = lane >> 4;           // frame within the group of 4
    const int f = blockIdx.x;
    const int k0 = blockIdx.z * KW;
    const int t_begin = blockIdx.y * tc;
    const int t_end = min(T, t_begin + tc);
    const int NA = M * M;
    const bool cvalid = ch < M;
    const float xmask = cvalid ? 1.f : 0.f;
    acc_t are[KW], air[KW];
#pragma unroll
    for (int kk = 0; kk < KW; ++kk)
#pragma unroll
        for (int r = 0; r < 4; ++r) are[kk][r] = air[kk][r] = 0;

    const size_t frame_stride = (size_t)F * M;                         // complex samples per frame
    const float2* pcol = X + (size_t)f * M + (cvalid ? ch : 0);
    const int ngroups = (t_end - t_begin + 3) >> 2;
    // x of group g: this lane's channel at frame t_begin + 4g + kf (frames past the split: clamped, masked below)
    auto fetch = [&](int g) {
        const int t = t_begin + 4 * g + kf;
        return pcol[(size_t)(t < t_end ? t : T - 1) * frame_stride];
    };
    // the KW weights of the lane's own frame: one row of the (T, Kp) table, as KW/4 (or fewer) 16-byte loads that
    // the 16 lanes of a frame share.  (Selecting them from wave-uniform scalar loads of all four frames costs
    // 4 FMAs per source on the VALU, next to 3 MFMAs per source: measured 1.95 -> see DESIGN.md.)
    constexpr int WV = (KW + 3) / 4;
    auto fetch_w = [&](int g, float4 (&w)[WV]) {
        const int t = min(t_begin + 4 * g + kf, T - 1);
        const float4* row = reinterpret_cast<const float4*>(Wt + (size_t)t * Kp + k0);
#pragma unroll
        for (int v = 0; v < WV; ++v) w[v] = row[v];
    };
    // kDepth groups of x and kWDepth groups of weights are in flight per wave.  (The compiler's wait at the loop head is
    // vmcnt(0), i.e. the effective distance is about one group; a variant with inline-assembly loads into kDepth slots
    // and counted waits -- true distance 3 groups = 2 us -- measured the same 1.81-1.83 ms at 2048 x 4000 x 16 / 16, so
    // latency is not what keeps the matrix pipe at 75 %, and the compiler-visible loads stay.)
    constexpr int kDepth = 4, kWDepth = 2;
    float2 xq[kDepth];
    float4 wq[kWDepth][WV];
#pragma unroll
    for (int u = 0; u < kDepth; ++u) xq[u] = fetch(u < ngroups ? u : ngroups - 1);
    if constexpr (!UNIT) {
#pragma unroll
        for (int u = 0; u < kWDepth; ++u) fetch_w(u < ngroups ? u : ngroups - 1, wq[u]);
    }
    for (int g0 = 0; g0 < ngroups; g0 += kDepth) {
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
            const int g = g0 + u;
            const float2 x = xq[u];
            xq[u] = fetch(g + kDepth < ngroups ? g + kDepth : ngroups - 1);
            float wl[4 * WV];
            if constexpr (!UNIT) {
#pragma unroll
                for (int v = 0; v < WV; ++v) {
                    wl[4 * v] = wq[u % kWDepth][v].x;
                    wl[4 * v + 1] = wq[u % kWDepth][v].y;
                    wl[4 * v + 2] = wq[u % kWDepth][v].z;
                    wl[4 * v + 3] = wq[u % kWDepth][v].w;
                }
                fetch_w(g + kWDepth < ngroups ? g + kWDepth : ngroups - 1, wq[u % kWDepth]);
            }
            if (g < ngroups) {                                         // uniform
                const int t0 = t_begin + 4 * g;
                const float live = (t0 + kf < t_end) ? xmask : 0.f;
                const REAL xr = (REAL)(x.x * live), xi = (REAL)(x.y * live);
                // With many sources: two rounds over them, so that the two MFMAs into are[kk] are never adjacent (a
                // dependent MFMA issued right behind its producer waits for the result while the pipe idles): 1.82 ->
                // 1.80 ms at 16 sources; with 2 sources the plain order measures 7 % faster (369 vs 395 us).
                REAL ai[KW];
#pragma unroll
                for (int kk = 0; kk < KW; ++kk) {
                    REAL ar = xr;
                    ai[kk] = xi;
                    if constexpr (!UNIT) {
                        ar = xr * (REAL)wl[kk];
                        ai[kk] = xi * (REAL)wl[kk];
                    }
                    are[kk] = Mfma<REAL>::run(ar, xr, are[kk]);
                    if constexpr (KW < 8) are[kk] = Mfma<REAL>::run(ai[kk], xi, are[kk]);
                    air[kk] = Mfma<REAL>::run(ai[kk], xr, air[kk]);
                }
                if constexpr (KW >= 8) {
#pragma unroll
                    for (int kk = 0; kk < KW; ++kk) are[kk] = Mfma<REAL>::run(ai[kk], xi, are[kk]);
                }
            }
        }
    }

    // accumulator tiles -> [row][col] in LDS (whatever the C/D layout of the instruction), then every lane
    // gathers 4 of the 256 (c, d) entries: V_re[c][d], V_im[c][d] = G_ir[c][d] - G_ir[d][c]
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
        const int k = k0 + kk;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            tile[0][Mfma<REAL>::row(lane, r)][ch] = are[kk][r];
            tile[1][Mfma<REAL>::row(lane, r)][ch] = air[kk][r];
        }
        __syncthreads();
        if (k < K) {
            REAL* out = Vpart + (((size_t)blockIdx.y * F + f) * K + k) * NA;
#pragma unroll
            for (int e = lane; e < 256; e += 64) {
                const int c = e >> 4, d = e & 15;
                if (c <= d && d < M) {
                    if (c == d) {
                        out[c] = tile[0][c][c];
                    } else {
                        const int o = herm_pair_index(M, c, d);
                        out[o] = tile[0][c][d];
                        out[o + 1] = tile[1][c][d] - tile[1][d][c];
                    }
                }
            }
        }
    }
}

template <typename REAL>
hipError_t launch_planar(hipStream_t s, const float2* X, const float* Wt, void* Vpart, bool unit, int T, int F, int M, int K,
                         int Kp, int nsplit, int tc) {
    constexpr int kMaxKw = sizeof(REAL) == 8 ? 8 : 16;          // 16 accumulator registers per source in float64
    int kw = unit ? 1 : (K <= 2 ? 2 : (K <= 4 ? 4 : (K <= 8 ? 8 : 16)));
    if (kw > kMaxKw) kw = kMaxKw;
    dim3 grid(F, nsplit, unit ? 1 : (K + kw - 1) / kw);
    REAL* V = static_cast<REAL*>(Vpart);
    if (unit)
        return launch_dominant(cov_mfma16_kernel<REAL, 1, true>, grid, dim3(64), 0, s, X, Wt, V, T, F, M, K, Kp, tc);
    else if (kw == 2)
        return launch_dominant(cov_mfma16_kernel<REAL, 2, false>, grid, dim3(64), 0, s, X, Wt, V, T, F, M, K, Kp, tc);
    else if (kw == 4)
        return launch_dominant(cov_mfma16_kernel<REAL, 4, false>, grid, dim3(64), 0, s, X, Wt, V, T, F, M, K, Kp, tc);
    else if (kw == 8)
        return launch_dominant(cov_mfma16_kernel<REAL, 8, false>, grid, dim3(64), 0, s, X, Wt, V, T, F, M, K, Kp, tc);
    else if constexpr (kMaxKw >= 16)
        return launch_dominant(cov_mfma16_kernel<REAL, 16, false>, grid, dim3(64), 0, s, X, Wt, V, T, F, M, K, Kp, tc);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_cov_weights(hipStream_t s, const float* R, float* Wt, float* wscale, int model, int raw, int T, int K, int Kp) {
    weights_kernel<<<dim3((unsigned)(((long long)T * Kp + kBlock - 1) / kBlock)), dim3(kBlock), 0, s>>>(R, Wt, wscale, model, raw, T, K, Kp);
    return hipGetLastError();
}

hipError_t launch_cov_mfma(hipStream_t s, const float2* X, const float* R, float* Wt, float* wscale, int model, int raw,
                           void* Vpart, bool f64, int T, int F, int M, int K, int nsplit, int tc) {
    const bool unit = R == nullptr;
    if (M > 16 || K > kMaxK) return hipErrorInvalidValue;
    const int Kp = (K + 15) / 16 * 16;      // padded row stride of the weights (scratch holds T * 16 floats)
    if (!unit) {
        if (Wt == nullptr) return hipErrorInvalidValue;
        hipError_t e = launch_cov_weights(s, R, Wt, wscale, model, raw, T, K, Kp);
        if (e != hipSuccess) return e;
    }
    if (f64) return launch_planar<double>(s, X, Wt, Vpart, unit, T, F, M, K, Kp, nsplit, tc);
    return launch_planar<float>(s, X, Wt, Vpart, unit, T, F, M, K, Kp, nsplit, tc);
}

}  // namespace oiva
