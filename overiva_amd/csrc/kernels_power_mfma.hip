// Demix + source power on the matrix cores, for 9..16 channels            reference overiva.py:140 + :153/:155
//
//   p[t,k] = sum_f |y_{t,f,k}|^2,   y = W^H x  is per bin a (K x M)(M x frames) product: planar form with
//   v_mfma_f32_16x16x4_f32, sources = rows, 16 frames = columns, channels = contraction in chunks of 4:
//       Y_re += Wr Xr + Wi Xi          Y_im += Wr Xi - Wi Xr          (y = sum_m conj(w_m) x_m)
//   i.e. 4 MFMAs per channel chunk, 16 per bin and 16 frames.  |y|^2 is accumulated over the bins of a batch
//   in the accumulator layout (lane = frame, 4 sources per lane), so nothing is exchanged between lanes and Y is
//   never stored.  The VALU kernel needs K/4 passes over X at K = 16 (register budget); this one reads X once.
#include "oiva_device.h"

#include <cstdlib>

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

// ---------------------------------------------------------------------------------------------------------------------
// The same product with X staged through LDS (16 channels, 64-bin batches; the last batch may be ragged, F >= 64).
// power_mfma_kernel asks memory for 128-byte runs 262 KB apart (lane = frame: 16 frames x one bin per load instruction) and reaches 0.56 of the HBM peak; the vector-ALU
// power_kernel<16, 2>, whose loads are 2 KB runs (lane = bin), moves the same bytes at 0.74.  Here the loads are those 2 KB
// runs -- one frame's 16 bins x 16 channels, by LDS-DMA, no staging registers -- and the matrix cores read their frame-major
// operands back from LDS:
//   workgroup = one 64-bin batch x kPlFrames frames; steps = (up to) 4 sub-batches of 16 bins x kPlTiles tiles of 16 frames;
//   a step: 16 rows of 2 KB -> LDS (pitch 2 KB + 16 bytes: lane (frame j, quarter q) reads 32 bytes at row j -- the 16 frames
//   of a read pass fall into 16 different bank groups); wave w multiplies bins 4w .. 4w + 3 of the sub-batch (W of its four
//   bins in registers for the sub-batch's four steps); two buffers, the DMA of step s + 1 in flight behind the products of s;
//   |y|^2 summed over the wave's bins in the accumulator layout, over the four waves through LDS at the end: the same
//   64-bin parts as every other power kernel.
// Round 5, measured by ablation at 2048 x 4000 x 16 / 16 (220 us): without the matrix instructions (DMAs, barriers, LDS reads)
// 202 us -- with ONE workgroup per CU instead of two still 204, so it is not the bytes in flight that bound the stream --, without
// the DMAs 145 us: the two overlap to within 18 us.  Round 6: what bound the stream at 5.2 TB/s was the ORDER of the sub-batches,
// the same in every workgroup (see `rot` below): rotated, the stream alone runs at 6.3 TB/s (165 us) and the kernel in 193 us.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kPlTiles = 4;                          // frame tiles of 16 per workgroup
constexpr int kPlFrames = 16 * kPlTiles;
constexpr int kPlRow = 16 * 16 * 8;                  // one frame's 16 bins x 16 channels
constexpr int kPlPitch = kPlRow + 16;
constexpr int kPlStage = 16 * kPlPitch;              // bytes per buffer

typedef __attribute__((address_space(1))) const void gvoid_pl_t;
typedef __attribute__((address_space(3))) void lvoid_pl_t;

__global__ __launch_bounds__(kBlock, 2) void power_lds_kernel(const float2* __restrict__ X, const float2* __restrict__ What,
                                                              float* __restrict__ Ppart, int T, int F, int K) {
    constexpr int M = 16;
    __shared__ __attribute__((aligned(16))) unsigned char stage[2][kPlStage];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15;             // A: source (row) | B: frame (column)
    const int q = lane >> 4;             // owner of channels 4q .. 4q + 3 (contraction index)
    const int f0 = blockIdx.x * kBinsPerBatch;
    const int t0 = blockIdx.y * kPlFrames;
    const size_t frame_stride = (size_t)F * M;       // float2 per frame

    // step s = (sub-batch sb = s / kPlTiles, tile tl = s % kPlTiles): rows = frames t0 + 16 tl + r, r < 16; this wave requests rows
    // 4 wave .. 4 wave + 3, each as two 1 KB halves (64 lanes x 16 bytes)
    // The four sub-batches are taken in an order rotated by the workgroup's index.  All workgroups of a launch move at the same
    // pace; with the same order everywhere, the chip asks memory for 2 KB of every 8 KB at any one time and the stream stops at
    // 5.3 TB/s -- rotated, the workgroups in flight cover all four quarters: 6.3 TB/s for the stream alone
    // (tools/r6/dmabench.hip, rows F and M).
    // (a ragged last batch has fewer sub-batches: those that start past the last bin are left out, one that is cut is shifted below)
    const int nsb = min(4, (F - f0 + 15) >> 4);
    const int nsteps = nsb * kPlTiles;
    const int rot = (int)((blockIdx.x + blockIdx.y) % (unsigned)nsb);
    auto sub_batch = [&](int sbi) {
        const int sb = sbi + rot;
        return sb < nsb ? sb : sb - nsb;
    };
    auto issue = [&](int s, int buf) {
        const int sb = sub_batch(s / kPlTiles), tl = s % kPlTiles;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = 4 * wave + rr;
            const int t = t0 + 16 * tl + r;
            // (frames past T: a valid row, never stored; a sub-batch that would run past the last bin starts at bin F - 16 instead and
            //  the bins it then holds twice meet a zero in W below)
            const float2* src = X + (size_t)(t < T ? t : T - 1) * frame_stride + (size_t)min(f0 + 16 * sb, F - 16) * M;
#pragma unroll
            for (int h = 0; h < 2; ++h)
                __builtin_amdgcn_global_load_lds((gvoid_pl_t*)(src + h * 128 + lane * 2), (lvoid_pl_t*)(stage[buf] + r * kPlPitch + h * 1024), 16, 0, 0);
        }
    };
    float P[kPlTiles][4];                // sources 4q .. 4q + 3 at frame t0 + 16 tl + j, summed over this wave's bins
#pragma unroll
    for (int tl = 0; tl < kPlTiles; ++tl)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[tl][r] = 0.f;

    // LDS byte address of this lane's 32 bytes of (row j, first bin of the wave) in buffer 0
    const unsigned rd_base = (unsigned)(uintptr_t)(&stage[0][0]) + j * kPlPitch + wave * 4 * 128 + q * 32;
    issue(0, 0);
    float2 w[4][4];                      // [bin of the wave][channel 4q + c]: W[f][m][k = j]
    for (int sbi = 0; sbi < nsb; ++sbi) {
        const int sb = sub_batch(sbi);
        // W of this wave's four bins of the sub-batch (from L2; once per four steps).  Requested BEFORE the DMAs of the next
        // step, so that the counted wait below covers it.
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int first = f0 + 16 * sb;                                  // the sub-batch's own bins start here ...
                const int bin = min(first, F - 16) + 4 * wave + g;               // ... the run in LDS may start earlier (ragged last batch)
                const bool ok = j < K && bin >= first;
                const float2 wv = What[((size_t)bin * M + 4 * q + c) * M + (j < K ? j : 0)];
                w[g][c] = make_float2(ok ? wv.x : 0.f, ok ? wv.y : 0.f);
            }
        static_for<kPlTiles>([&](auto tc) {
            constexpr int tl = decltype(tc)::value;
            const int s = kPlTiles * sbi + tl;
            // buffer (s + 1) & 1 was read in step s - 1 and every wave has passed that step's second barrier
            if (s + 1 < nsteps) issue(s + 1, (s + 1) & 1);
            // this wave's eight requests of step s have landed (those of s + 1 -- and nothing else -- may still be in flight),
            // then every wave's.  Raw barrier: __syncthreads() carries a release fence, which drains the DMA queue.
            if (s + 1 < nsteps)
                asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            float4 v[8];
            // (the reads in assembly: hipcc puts a vmcnt(0) in front of every LDS read it can see behind an LDS-DMA)
            asm volatile(
                "ds_read_b128 %0, %8\n\t"
                "ds_read_b128 %1, %8 offset:16\n\t"
                "ds_read_b128 %2, %8 offset:128\n\t"
                "ds_read_b128 %3, %8 offset:144\n\t"
                "ds_read_b128 %4, %8 offset:256\n\t"
                "ds_read_b128 %5, %8 offset:272\n\t"
                "ds_read_b128 %6, %8 offset:384\n\t"
                "ds_read_b128 %7, %8 offset:400\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "s_barrier"                   // every wave holds its operands: the buffer may be requested into again
                : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                : "v"(rd_base + (unsigned)((s & 1) * kPlStage))
                : "memory");
            f32x4 yr[4], yi[4];
            float2 x[4][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                x[g][0] = make_float2(v[2 * g].x, v[2 * g].y);
                x[g][1] = make_float2(v[2 * g].z, v[2 * g].w);
                x[g][2] = make_float2(v[2 * g + 1].x, v[2 * g + 1].y);
                x[g][3] = make_float2(v[2 * g + 1].z, v[2 * g + 1].w);
                yr[g] = yi[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // chunk-major over the four bins: eight independent accumulators, an MFMA never waits for the one before it
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    yr[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g][c].x, x[g][c].x, yr[g], 0, 0, 0);
                    yi[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g][c].x, x[g][c].y, yi[g], 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    yr[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g][c].y, x[g][c].y, yr[g], 0, 0, 0);
                    yi[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(-w[g][c].y, x[g][c].x, yi[g], 0, 0, 0);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) P[tl][r] = fmaf(yr[g][r], yr[g][r], fmaf(yi[g][r], yi[g][r], P[tl][r]));
        });
    }
    __syncthreads();
    // the four waves' sums (wave order) -> Ppart; D layout: lane l, register r = [source 4 (l >> 4) + r][frame l & 15]
    float* red = reinterpret_cast<float*>(stage[0]);       // [wave][tile][source 16][frame 16]
#pragma unroll
    for (int tl = 0; tl < kPlTiles; ++tl)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * kPlTiles + tl) * 16 + 4 * q + r) * 16 + j] = P[tl][r];
    __syncthreads();
    for (int e = tid; e < kPlFrames * 16; e += kBlock) {
        const int tl = e >> 8, k = (e >> 4) & 15, jj = e & 15;      // consecutive threads: consecutive frames of one source
        const int t = t0 + 16 * tl + jj;
        float sum = red[((0 * kPlTiles + tl) * 16 + k) * 16 + jj];
#pragma unroll
        for (int wv = 1; wv < kWaves; ++wv) sum += red[((wv * kPlTiles + tl) * 16 + k) * 16 + jj];
        if (t < T && k < K) Ppart[((size_t)blockIdx.x * T + t) * K + k] = sum;
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
    // 16 channels, at least 64 bins: X staged through LDS in 2 KB runs (power_lds_kernel above); $OIVA_POWER_LDS=0: off
    static const bool lds = [] { const char* v = getenv("OIVA_POWER_LDS"); return !(v && v[0] == '0'); }();
    if (lds && M == 16 && Mp == 16 && F >= kBinsPerBatch && T >= 16) {
        power_lds_kernel<<<dim3((F + kBinsPerBatch - 1) / kBinsPerBatch, (T + kPlFrames - 1) / kPlFrames), dim3(kBlock), 0, s>>>(X, What, Ppart, T, F, K);
        return hipGetLastError();
    }
    return launch_shape<4, 1>(s, X, What, Ppart, T, F, M, Mp, K);
}

}  // namespace oiva
