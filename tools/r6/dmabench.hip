// Micro-benchmark (GPU box): the LDS-DMA stream of the 16-channel power pass, memory only (no LDS reads, no arithmetic), at 2048 bins x
// 4000 frames x 16 channels (1.05 GB): workgroup = 64 bins x 64 frames, 16 steps of 32 KB, two buffers, raw barriers --
//   A: a step = 16 frames x 16 bins: runs of 2 KB, 262 KB apart (power_lds_kernel)
//   B: a step = 4 frames x 64 bins:  runs of 8 KB
//   C: as B with 128 frames per workgroup (32 steps)
//   hipcc --offload-arch=gfx950 -O3 tools/r6/dmabench.hip -o /tmp/dmabench && /tmp/dmabench
#include <hip/hip_runtime.h>

#include <cstdio>

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
constexpr int M = 16;

template <int MODE, int FRAMES>
__global__ __launch_bounds__(256, 2) void stream(const float2* __restrict__ X, float* out, int T, int F) {
    __shared__ __attribute__((aligned(16))) unsigned char stage[2][32768 + 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int f0 = blockIdx.x * 64, t0 = blockIdx.y * FRAMES;
    const size_t frame_stride = (size_t)F * M;
    constexpr int steps = FRAMES / 4;
    auto issue = [&](int s, int buf) {
        if (MODE == 0) {
            const int sb = s / (steps / 4), tl = s % (steps / 4);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = 4 * wave + rr, t = t0 + 16 * tl + r;
                const float2* src = X + (size_t)(t < T ? t : T - 1) * frame_stride + (size_t)(f0 + 16 * sb) * M;
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    __builtin_amdgcn_global_load_lds((gvoid_t*)(src + h * 128 + lane * 2), (lvoid_t*)(stage[buf] + r * 2064 + h * 1024), 16, 0, 0);
            }
        } else {
            const int t = t0 + 4 * s + wave;
            const float2* src = X + (size_t)(t < T ? t : T - 1) * frame_stride + (size_t)f0 * M;
#pragma unroll
            for (int h = 0; h < 8; ++h)
                __builtin_amdgcn_global_load_lds((gvoid_t*)(src + h * 128 + lane * 2), (lvoid_t*)(stage[buf] + wave * 8208 + h * 1024), 16, 0, 0);
        }
    };
    issue(0, 0);
    for (int s = 0; s < steps; ++s) {
        if (s + 1 < steps) issue(s + 1, (s + 1) & 1);
        if (s + 1 < steps)
            asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
    }
    if (out != nullptr && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) out[0] = reinterpret_cast<float*>(stage[0])[lane];
}

template <int MODE, int FRAMES>
void run(const char* name, const float2* X, int T, int F) {
    dim3 grid(F / 64, (T + FRAMES - 1) / FRAMES);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((stream<MODE, FRAMES>), grid, dim3(256), 0, 0, X, (float*)nullptr, T, F);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((stream<MODE, FRAMES>), grid, dim3(256), 0, 0, X, (float*)nullptr, T, F);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)T * F * M * 8;
    printf("%-52s %7.1f us  %6.2f TB/s\n", name, ms / 10 * 1e3, bytes / (ms / 10 * 1e-3) / 1e12);
}

// The same stream through REGISTERS: a step's 32 KB as eight 16-byte loads per lane (two steps in flight = 64 registers), written to
// LDS with ds_write_b128 when they have landed, one barrier per step (the write of step s + 2 into a buffer comes after the barrier
// of step s + 1, which every wave passes after its reads of step s).  READ: every wave also reads its 8 x 16 bytes back, as the
// power kernel does.
template <int MODE, int FRAMES, bool READ, bool SYNC = true>
__global__ __launch_bounds__(256, 2) void stream_reg(const float2* __restrict__ X, float* out, int T, int F) {
    __shared__ __attribute__((aligned(16))) unsigned char stage[2][32768 + 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int f0 = blockIdx.x * 64, t0 = blockIdx.y * FRAMES;
    const size_t frame_stride = (size_t)F * M;
    constexpr int steps = FRAMES / 4;
    // address of load e of step s (e = 0..7)
    auto src = [&](int s, int e) -> const float4* {
        if (MODE == 0 || MODE == 2 || MODE == 3) {
            int sb = s / (steps / 4), tl = s % (steps / 4);
            if (MODE == 2) sb = (sb + blockIdx.x + blockIdx.y) & 3;
            if (MODE == 3) sb = s & 3, tl = s >> 2;
            const int r = 4 * wave + (e >> 1), t = t0 + 16 * tl + r;
            return reinterpret_cast<const float4*>(X + (size_t)(t < T ? t : T - 1) * frame_stride + (size_t)(f0 + 16 * sb) * M + (e & 1) * 128 + lane * 2);
        } else {
            const int t = t0 + 4 * s + wave;
            return reinterpret_cast<const float4*>(X + (size_t)(t < T ? t : T - 1) * frame_stride + (size_t)f0 * M + e * 128 + lane * 2);
        }
    };
    auto dst = [&](int buf, int e) -> float4* {
        unsigned char* base = stage[buf] + (MODE != 1 ? 4 * wave * 2064 : wave * 8208) + lane * 16;
        return reinterpret_cast<float4*>(base + (MODE != 1 ? (e >> 1) * 2064 + (e & 1) * 1024 : e * 1024));
    };
#define ISSUE(s, r) r##0 = *src(s, 0), r##1 = *src(s, 1), r##2 = *src(s, 2), r##3 = *src(s, 3), r##4 = *src(s, 4), r##5 = *src(s, 5), r##6 = *src(s, 6), r##7 = *src(s, 7)
#define S4(v) ((v).x + (v).y + (v).z + (v).w)
#define SUM(r) acc.x += S4(r##0) + S4(r##1) + S4(r##2) + S4(r##3) + S4(r##4) + S4(r##5) + S4(r##6) + S4(r##7)
#define PUT(s, r) *dst((s) & 1, 0) = r##0, *dst((s) & 1, 1) = r##1, *dst((s) & 1, 2) = r##2, *dst((s) & 1, 3) = r##3, *dst((s) & 1, 4) = r##4, *dst((s) & 1, 5) = r##5, *dst((s) & 1, 6) = r##6, *dst((s) & 1, 7) = r##7
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto readback = [&](int s) {
        if (READ) {
            const int j = lane & 15, q = lane >> 4;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 v = *reinterpret_cast<const float4*>(stage[s & 1] + j * 2064 + wave * 512 + g * 128 + q * 32 + h * 16);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
        }
    };
    float4 a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7;
    ISSUE(0, a);
#pragma unroll 1
    for (int s = 0; s < steps - 2; s += 2) {
        ISSUE(s + 1, b);
        __builtin_amdgcn_sched_barrier(0);
        if (SYNC) {
            PUT(s, a);
            __syncthreads();
            readback(s);
        } else {
            SUM(a);
        }
        ISSUE(s + 2, a);
        __builtin_amdgcn_sched_barrier(0);
        if (SYNC) {
            PUT(s + 1, b);
            __syncthreads();
            readback(s + 1);
        } else {
            SUM(b);
        }
    }
    ISSUE(steps - 1, b);
    __builtin_amdgcn_sched_barrier(0);
    if (SYNC) {
        PUT(steps - 2, a);
        __syncthreads();
        readback(steps - 2);
        PUT(steps - 1, b);
        __syncthreads();
        readback(steps - 1);
    } else {
        SUM(a);
        SUM(b);
    }
    if (out != nullptr && (acc.x + acc.y + acc.z + acc.w == 12345.f || (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0)))
        out[0] = reinterpret_cast<float*>(stage[0])[lane] + acc.x;
}

template <int MODE, int FRAMES, bool READ, bool SYNC = true>
void run_reg(const char* name, const float2* X, float* out, int T, int F) {
    dim3 grid(F / 64, (T + FRAMES - 1) / FRAMES);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((stream_reg<MODE, FRAMES, READ, SYNC>), grid, dim3(256), 0, 0, X, out, T, F);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((stream_reg<MODE, FRAMES, READ, SYNC>), grid, dim3(256), 0, 0, X, out, T, F);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)T * F * M * 8;
    printf("%-60s %7.1f us  %6.2f TB/s\n", name, ms / 10 * 1e3, bytes / (ms / 10 * 1e-3) / 1e12);
}

// The 8-channel covariance pass's stream (cov_dma_kernel: workgroup = 16 bins x one of four frame splits, a wave = 16 bins x 4
// frames per step, private ring of four stages, no barrier; 2048 x 4000 x 8 = 524 MB), memory only, by the footprint of ONE
// DMA instruction:  PAT 0: piece j of every lane's own 64-byte vector -- 32 lines, a quarter of each (the kernel's form);
// PAT 1: one frame's 1 KB, lane (r, b) -> piece r of bin b (quads of lanes 64 bytes apart, 8 whole lines per instruction);
// PAT 2: one frame's 1 KB, lane l -> bytes 16 l (quads contiguous).
template <int PAT>
__global__ __launch_bounds__(256, 2) void stream8(const float2* __restrict__ X, float* out, int T, int F, int tc) {
    constexpr int M8 = 8;
    __shared__ float4 ring[4 * 4 * 256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t_begin = blockIdx.y * tc, t_end = min(T, t_begin + tc), nsteps = (t_end - t_begin + 15) >> 4;
    const size_t frame_stride = (size_t)F * M8;
    float4* wring = ring + wave * 4 * 256;
    const int b = lane & 15, ql = lane >> 4;
    auto issue = [&](int i, int st) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2* src;
            if (PAT == 0) {
                const int t = t_begin + wave * 4 + ql + 16 * i;
                src = X + (size_t)(i < nsteps && t < t_end ? t : T - 1) * frame_stride + (size_t)(blockIdx.x * 16 + b) * M8 + 2 * j;
            } else {
                const int t = t_begin + wave * 4 + j + 16 * i;
                const int off = PAT == 1 ? (lane & 15) * M8 + 2 * (lane >> 4) : 2 * lane;
                src = X + (size_t)(i < nsteps && t < t_end ? t : T - 1) * frame_stride + (size_t)(blockIdx.x * 16) * M8 + off;
            }
            __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)(wring + st * 256 + j * 64), 16, 0, 0);
        }
    };
    issue(0, 0);
    issue(1, 1);
    issue(2, 2);
    for (int i = 0; i < nsteps; ++i) {
        issue(i + 3, (i + 3) & 3);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (out != nullptr && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) out[1] = reinterpret_cast<float*>(ring)[lane];
}

template <int PAT>
void run8(const char* name, const float2* X, int T, int F) {
    const int tc = (T + 3) / 4;
    dim3 grid(F / 16, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((stream8<PAT>), grid, dim3(256), 0, 0, X, (float*)nullptr, T, F, tc);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((stream8<PAT>), grid, dim3(256), 0, 0, X, (float*)nullptr, T, F, tc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)T * F * 8 * 8;
    printf("%-60s %7.1f us  %6.2f TB/s\n", name, ms / 10 * 1e3, bytes / (ms / 10 * 1e-3) / 1e12);
}

int main() {
    const int T = 4000, F = 2048;
    float2* X;
    (void)hipMalloc(&X, (size_t)T * F * M * 8 + (1 << 20));
    (void)hipMemset(X, 0, (size_t)T * F * M * 8);
    float* out;
    (void)hipMalloc(&out, 4096);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 64>("A: 16 frames x 16 bins per step (2 KB runs)", X, T, F);
        run<1, 64>("B: 4 frames x 64 bins per step (8 KB runs)", X, T, F);
        run<1, 128>("C: as B, 128 frames per workgroup", X, T, F);
        run<0, 128>("D: as A, 128 frames per workgroup", X, T, F);
        run_reg<0, 64, false>("E: as A through registers + ds_write_b128", X, out, T, F);
        run_reg<0, 64, true>("F: as E, operands read back from LDS", X, out, T, F);
        run_reg<1, 64, false>("G: as B through registers + ds_write_b128", X, out, T, F);
        run_reg<0, 128, true>("H: as F, 128 frames per workgroup", X, out, T, F);
        run_reg<2, 64, true>("M: as F, sub-batch order rotated by the workgroup index", X, out, T, F);
        run_reg<3, 64, true>("N: as F, tile-major (the four sub-batches of a tile in turn)", X, out, T, F);
        run_reg<2, 128, true>("O: as M, 128 frames per workgroup", X, out, T, F);
        run_reg<3, 128, true>("P: as N, 128 frames per workgroup", X, out, T, F);
        run_reg<0, 64, false, false>("I: as E, no LDS and no barrier (waves run free)", X, out, T, F);
        run_reg<1, 64, false, false>("J: as G, no LDS and no barrier", X, out, T, F);
        run_reg<1, 128, false, false>("K: as J, 128 frames per workgroup", X, out, T, F);
        run_reg<1, 512, false, false>("L: as J, 512 frames per workgroup", X, out, T, F);
    }
    for (int rep = 0; rep < 2; ++rep) {
        run8<0>("8 ch, Q: a lane's own vector, piece by piece (cov_dma_kernel)", X, T, F);
        run8<1>("8 ch, R: a frame's 1 KB per instruction, lane (r, b)", X, T, F);
        run8<2>("8 ch, S: a frame's 1 KB per instruction, lane l -> 16 l", X, T, F);
    }
    return 0;
}
