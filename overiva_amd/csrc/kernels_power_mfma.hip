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

// One wave = 16 x TILES frames x the 64 bins of a batch, in groups of GROUP bins.  The operands of a whole group are
// requested at once -- per frame row one contiguous run of GROUP x M x 8 bytes -- and two further groups are in flight
// while the MFMAs of this one run.  The contraction order is free, so lane (j, q) takes the four CONSECUTIVE channels
// m = 4q + c, c = 0..3, as its share of the four chunks: its 32 bytes of a frame are two 16-byte loads (a chunk of every
// fourth channel would be four 8-byte loads 32 bytes apart).
// Mp: channel pitch of X (>= M; an odd channel count reads the plan's copy of X padded by one zero channel, Mp = M + 1)
template <bool VEC, int TILES, int GROUP>   // VEC: even pitch -- a lane's channel quarter starts on a 16-byte boundary
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2, 2))) void power_mfma_kernel(
    const float2* __restrict__ X, const float2* __restrict__ What, float* __restrict__ Ppart, int T, int F, int M, int Mp, int K) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 15;             // A: source (row) | B: frame (column)
    const int q = lane >> 4;             // owner of channels 4q .. 4q + 3 (contraction index)
    const int t0 = (blockIdx.y * kWaves + wave) * 16 * TILES + j;
    const int f0 = blockIdx.x * kBinsPerBatch;
    const int nbins = min(kBinsPerBatch, F - f0);
    // frames past T and channels past M read a valid (clamped) address: the first are never stored, the second meet a
    // zero in W
    const float2* px[TILES];
#pragma unroll
    for (int tl = 0; tl < TILES; ++tl) {
        const int t = t0 + 16 * tl;
        px[tl] = X + ((size_t)(t < T ? t : T - 1) * F + f0) * Mp + (4 * q < Mp ? 4 * q : 0);   // + b*Mp (+ c)
    }
    const float2* pw = What + (size_t)f0 * M * M;                                            // + (b*M + m)*M + k
    float P[TILES][4];                   // sources 4q..4q+3 at frame t0 + 16 tl
#pragma unroll
    for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[tl][r] = 0.f;

    struct Ops {
        float2 w[GROUP][4];
        float2 x[TILES][GROUP][4];
    };
    auto fetch = [&](int b0, Ops& o) {
#pragma unroll
        for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
            for (int g = 0; g < GROUP; ++g) {
                const int b = b0 + g < nbins ? b0 + g : nbins - 1;
                if constexpr (VEC) {
                    // (a pitch that is no multiple of 4: the last quarter holds two channels, its second load repeats the
                    //  first instead of running past the bin -- those channels meet a zero in W)
                    const float4* p = reinterpret_cast<const float4*>(px[tl] + (size_t)b * Mp);
                    const float4 lo = p[0], hi = p[4 * q + 2 < Mp ? 1 : 0];
                    o.x[tl][g][0] = make_float2(lo.x, lo.y);
                    o.x[tl][g][1] = make_float2(lo.z, lo.w);
                    o.x[tl][g][2] = make_float2(hi.x, hi.y);
                    o.x[tl][g][3] = make_float2(hi.z, hi.w);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) o.x[tl][g][c] = px[tl][(size_t)b * Mp + (4 * q + c < Mp ? c : 0)];
                }
            }
#pragma unroll
        for (int g = 0; g < GROUP; ++g) {
            const int b = b0 + g < nbins ? b0 + g : nbins - 1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int m = 4 * q + c;
                const bool ok = m < M && j < K && b0 + g < nbins;     // bins past the batch contribute nothing
                const float2 wv = pw[((size_t)b * M + (m < M ? m : 0)) * M + (j < K ? j : 0)];
                o.w[g][c] = make_float2(ok ? wv.x : 0.f, ok ? wv.y : 0.f);
            }
        }
    };
    // chunk-major over the tiles and bins of the group: 2 x TILES x GROUP independent accumulators, so an MFMA never
    // waits for the result of the one issued just before it
    auto compute = [&](const Ops& o) {
        f32x4 yr[TILES][GROUP], yi[TILES][GROUP];
#pragma unroll
        for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
            for (int g = 0; g < GROUP; ++g) yr[tl][g] = yi[tl][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                for (int g = 0; g < GROUP; ++g) {
                    yr[tl][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.w[g][c].x, o.x[tl][g][c].x, yr[tl][g], 0, 0, 0);
                    yi[tl][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.w[g][c].x, o.x[tl][g][c].y, yi[tl][g], 0, 0, 0);
                }
#pragma unroll
            for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                for (int g = 0; g < GROUP; ++g) {
                    yr[tl][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.w[g][c].y, o.x[tl][g][c].y, yr[tl][g], 0, 0, 0);
                    yi[tl][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(-o.w[g][c].y, o.x[tl][g][c].x, yi[tl][g], 0, 0, 0);
                }
        }
#pragma unroll
        for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
            for (int g = 0; g < GROUP; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) P[tl][r] = fmaf(yr[tl][g][r], yr[tl][g][r], fmaf(yi[tl][g][r], yi[tl][g][r], P[tl][r]));
    };
    // Two groups in flight behind the one being multiplied.  The scheduling barriers keep the machine scheduler from
    // sinking the loads next to their use to save registers, which is what it does otherwise -- and what removes the
    // prefetch.  (Three groups in flight need a fourth operand set: 256 registers + 36-52 bytes of scratch, and measured
    // 4-10 % slower -- 2048 x 4000 x 16 / 16: 255-273 against 244-266 us on the same box.)
    Ops o0, o1, o2;
    auto stage = [&](int b, Ops& in_flight, const Ops& ready) {
        fetch(b, in_flight);             // past the batch: clamped addresses, zero W
        __builtin_amdgcn_sched_barrier(0);
        compute(ready);
        __builtin_amdgcn_sched_barrier(0);
    };
    fetch(0, o0);
    fetch(GROUP, o1);
    __builtin_amdgcn_sched_barrier(0);
    for (int b0 = 0; b0 < nbins; b0 += 3 * GROUP) {
        stage(b0 + 2 * GROUP, o2, o0);
        if (b0 + GROUP >= nbins) break;
        stage(b0 + 3 * GROUP, o0, o1);
        if (b0 + 2 * GROUP >= nbins) break;
        stage(b0 + 4 * GROUP, o1, o2);
    }
    // D layout: lane l, register r = [source 4*(l>>4) + r][frame l&15]
#pragma unroll
    for (int tl = 0; tl < TILES; ++tl) {
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

template <int TILES, int GROUP>
static hipError_t launch_shape(hipStream_t s, const float2* X, const float2* What, float* Ppart, int T, int F, int M, int Mp, int K) {
    dim3 grid((F + kBinsPerBatch - 1) / kBinsPerBatch, (T + 16 * kWaves * TILES - 1) / (16 * kWaves * TILES));
    if ((Mp & 1) == 0)
        power_mfma_kernel<true, TILES, GROUP><<<grid, dim3(kBlock), 0, s>>>(X, What, Ppart, T, F, M, Mp, K);
    else
        power_mfma_kernel<false, TILES, GROUP><<<grid, dim3(kBlock), 0, s>>>(X, What, Ppart, T, F, M, Mp, K);
    return hipGetLastError();
}

hipError_t launch_power_mfma(hipStream_t s, const float2* X, const float2* What, float* Ppart, int T, int F, int M, int Mp, int K) {
    // (round 4, VERDICT r03 5(i): X AND W through a global_load_lds ring -- one bin per stage, 2 tiles, 2 or 3 stages, 8-12
    //  waves per CU, no load result held in registers, counted vmcnt waits only -- measured on one box: 254-258 us against
    //  252 for this kernel; the same loop without the matrix instructions 208-219 us, without the DMAs 140-148 us; whole
    //  128-byte lines per DMA instruction, no W traffic, the four waves on adjacent bins: all within 201-232 us memory-only.
    //  The stream of 128-byte runs 262 KB apart is the limit, not the registers.  Dropped.)
    // measured at 2048 x 4000 x 16 / 16 (tiles x bins per group): 4x1 252 us, 2x1 252, 2x2 267, 1x2 279, 1x4 306 -- the
    // W operands come from L2 once per wave and bin, so more frames per wave is less W traffic (W-only 81 us at 1x4);
    // X alone streams in 199 us, the MFMAs alone take 134 us
    return launch_shape<4, 1>(s, X, What, Ppart, T, F, M, Mp, K);
}

}  // namespace oiva
