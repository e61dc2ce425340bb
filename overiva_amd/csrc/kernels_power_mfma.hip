// Demix + source power on the matrix cores, for 9..16 channels            reference overiva.py:140 + :153/:155
//
//   p[t,k] = sum_f |y_{t,f,k}|^2,   y = W^H x  is per bin a (K x M)(M x frames) product: planar form with
//   v_mfma_f32_16x16x4_f32, sources = rows, 16 frames = columns, channels = contraction in chunks of 4:
//       Y_re += Wr Xr + Wi Xi          Y_im += Wr Xi - Wi Xr          (y = sum_m conj(w_m) x_m)
//   i.e. 4 MFMAs per channel chunk, 16 per bin and 16 frames.  |y|^2 is accumulated over the bins of a batch
//   in the accumulator layout (lane = frame, 4 sources per lane), so nothing is exchanged between lanes and Y is
//   never stored.  The VALU kernel needs K/4 passes over X at K = 16 (register budget); this one reads X once.
#include "oiva_device.h"

namespace oiva {
namespace {

constexpr int kBinsPerBatch = kBinsPerWave * kWaves;   // 64: same batches (and Ppart layout) as power_kernel

constexpr int kFrameTiles = 4;       // 16-frame tiles per wave: the W operands of a bin are loaded once for 64 frames

__global__ __launch_bounds__(kBlock) void power_mfma_kernel(const float2* __restrict__ X, const float2* __restrict__ What,
                                                            float* __restrict__ Ppart, int T, int F, int M, int K) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 15;             // A: source (row) | B: frame (column)
    const int q = lane >> 4;             // channel within the chunk of 4 (contraction index)
    const int t0 = (blockIdx.y * kWaves + wave) * 16 * kFrameTiles + j;
    const int f0 = blockIdx.x * kBinsPerBatch;
    const int nbins = min(kBinsPerBatch, F - f0);
    const int nchunks = (M + 3) >> 2;

    float P[kFrameTiles][4];             // sources 4q..4q+3 at frame t0 + 16 * tile
    const float2* px[kFrameTiles];
    float tmask[kFrameTiles];
    const size_t frame_stride = (size_t)F * M;
#pragma unroll
    for (int tl = 0; tl < kFrameTiles; ++tl) {
        const int t = t0 + 16 * tl;
        tmask[tl] = t < T ? 1.f : 0.f;
        px[tl] = X + (size_t)(t < T ? t : T - 1) * frame_stride + (size_t)f0 * M;   // + b*M + m
#pragma unroll
        for (int r = 0; r < 4; ++r) P[tl][r] = 0.f;
    }
    const float2* pw = What + (size_t)f0 * M * M;                                   // + (b*M + m)*M + k

    // operands of one bin: chunk c holds channel m = 4c + q; zero outside M x K
    auto fetch_w = [&](int b, float2 (&w)[4]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int m = 4 * c + q;
            const bool ok = m < M && j < K;
            const float2 wv = pw[((size_t)b * M + (m < M ? m : 0)) * M + (j < K ? j : 0)];
            w[c] = make_float2(ok ? wv.x : 0.f, ok ? wv.y : 0.f);
        }
    };
    auto fetch_x = [&](int b, float2 (&x)[kFrameTiles][4]) {
#pragma unroll
        for (int tl = 0; tl < kFrameTiles; ++tl)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int m = 4 * c + q;
                x[tl][c] = px[tl][(size_t)b * M + (m < M ? m : 0)];
            }
    };
    float2 w[4], wn[4], x[kFrameTiles][4], xn[kFrameTiles][4];
    fetch_w(0, w);
    fetch_x(0, x);
    for (int b = 0; b < nbins; ++b) {
        const int bn = b + 1 < nbins ? b + 1 : b;
        fetch_w(bn, wn);                 // next bin's operands are in flight during the MFMAs
        fetch_x(bn, xn);
#pragma unroll
        for (int tl = 0; tl < kFrameTiles; ++tl) {
            f32x4 yr = {0.f, 0.f, 0.f, 0.f}, yi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < nchunks) {
                    const float xm = (4 * c + q < M) ? tmask[tl] : 0.f;
                    const float xr = x[tl][c].x * xm, xi = x[tl][c].y * xm;
                    yr = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].x, xr, yr, 0, 0, 0);
                    yi = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].x, xi, yi, 0, 0, 0);
                    yr = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].y, xi, yr, 0, 0, 0);
                    yi = __builtin_amdgcn_mfma_f32_16x16x4f32(-w[c].y, xr, yi, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) P[tl][r] = fmaf(yr[r], yr[r], fmaf(yi[r], yi[r], P[tl][r]));
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) w[c] = wn[c];
#pragma unroll
        for (int tl = 0; tl < kFrameTiles; ++tl)
#pragma unroll
            for (int c = 0; c < 4; ++c) x[tl][c] = xn[tl][c];
    }
    // D layout: lane l, register r = [source 4*(l>>4) + r][frame l&15]
#pragma unroll
    for (int tl = 0; tl < kFrameTiles; ++tl) {
        const int t = t0 + 16 * tl;
        if (t < T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * q + r;
                if (k < K) Ppart[((size_t)blockIdx.x * T + t) * K + k] = P[tl][r];
            }
        }
    }
}

}  // namespace

hipError_t launch_power_mfma(hipStream_t s, const float2* X, const float2* What, float* Ppart, int T, int F, int M, int K) {
    dim3 grid((F + kBinsPerBatch - 1) / kBinsPerBatch, (T + 16 * kWaves * kFrameTiles - 1) / (16 * kWaves * kFrameTiles));
    power_mfma_kernel<<<grid, dim3(kBlock), 0, s>>>(X, What, Ppart, T, F, M, K);
    return hipGetLastError();
}

}  // namespace oiva
