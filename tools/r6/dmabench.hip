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

int main() {
    const int T = 4000, F = 2048;
    float2* X;
    (void)hipMalloc(&X, (size_t)T * F * M * 8 + (1 << 20));
    (void)hipMemset(X, 0, (size_t)T * F * M * 8);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 64>("A: 16 frames x 16 bins per step (2 KB runs)", X, T, F);
        run<1, 64>("B: 4 frames x 64 bins per step (8 KB runs)", X, T, F);
        run<1, 128>("C: as B, 128 frames per workgroup", X, T, F);
        run<0, 128>("D: as A, 128 frames per workgroup", X, T, F);
    }
    return 0;
}
