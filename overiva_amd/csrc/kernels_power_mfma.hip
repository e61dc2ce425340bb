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

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kBinsPerBatch = kBinsPerWave * kWaves;   // 64: same batches (and Ppart layout) as power_kernel

__global__ __launch_bounds__(kBlock) void power_mfma_kernel(const float2* __restrict__ X, const float2* __restrict__ What,
                                                            float* __restrict__ Ppart, int T, int F, int M, int K) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 15;             // A: source (row) | B: frame (column)
    const int q = lane >> 4;             // channel within the chunk of 4 (contraction index)
    const int t = (blockIdx.y * kWaves + wave) * 16 + j;
    const int tcl = t < T ? t : T - 1;
    const float tmask = t < T ? 1.f : 0.f;
    const int f0 = blockIdx.x * kBinsPerBatch;
    const int nbins = min(kBinsPerBatch, F - f0);
    const int nchunks = (M + 3) >> 2;

    float P[4] = {0.f, 0.f, 0.f, 0.f};   // sources 4q..4q+3 at frame j
    const size_t frame_stride = (size_t)F * M;
    const float2* px = X + (size_t)tcl * frame_stride + (size_t)f0 * M;   // + b*M + m
    const float2* pw = What + (size_t)f0 * M * M;                         // + (b*M + m)*M + k

    // operands of one bin: chunk c holds channel m = 4c + q; zero outside M x K
    auto fetch = [&](int b, float2 (&w)[4], float2 (&x)[4]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int m = 4 * c + q;
            const bool mv = m < M;
            const int mc = mv ? m : 0;
            const float2 wv = pw[((size_t)b * M + mc) * M + (j < K ? j : 0)];
            const float2 xv = px[(size_t)b * M + mc];
            const float wm = (mv && j < K) ? 1.f : 0.f, xm = mv ? tmask : 0.f;
            w[c] = make_float2(wv.x * wm, wv.y * wm);
            x[c] = make_float2(xv.x * xm, xv.y * xm);
        }
    };
    float2 w[4], x[4], wn[4], xn[4];
    fetch(0, w, x);
    for (int b = 0; b < nbins; ++b) {
        fetch(b + 1 < nbins ? b + 1 : b, wn, xn);     // next bin's operands are in flight during the MFMAs
        f32x4 yr = {0.f, 0.f, 0.f, 0.f}, yi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < nchunks) {
                yr = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].x, x[c].x, yr, 0, 0, 0);
                yi = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].x, x[c].y, yi, 0, 0, 0);
                yr = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].y, x[c].y, yr, 0, 0, 0);
                yi = __builtin_amdgcn_mfma_f32_16x16x4f32(-w[c].y, x[c].x, yi, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) P[r] = fmaf(yr[r], yr[r], fmaf(yi[r], yi[r], P[r]));
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            w[c] = wn[c];
            x[c] = xn[c];
        }
    }
    // D layout: lane l, register r = [source 4*(l>>4) + r][frame l&15]
    if (t < T) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 4 * q + r;
            if (k < K) Ppart[((size_t)blockIdx.x * T + t) * K + k] = P[r];
        }
    }
}

}  // namespace

hipError_t launch_power_mfma(hipStream_t s, const float2* X, const float2* What, float* Ppart, int T, int F, int M, int K) {
    dim3 grid((F + kBinsPerBatch - 1) / kBinsPerBatch, (T + 16 * kWaves - 1) / (16 * kWaves));
    power_mfma_kernel<<<grid, dim3(kBlock), 0, s>>>(X, What, Ppart, T, F, M, K);
    return hipGetLastError();
}

}  // namespace oiva
