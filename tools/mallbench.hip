// Does alternating the streaming direction between passes let the second pass hit the 256 MB Infinity Cache?
// All workgroups run concurrently (one round); each streams its own contiguous chunk ascending or descending.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void stream(const float4* __restrict__ X, float* out, size_t chunk4, int dir) {
    const float4* p = X + (size_t)blockIdx.x * chunk4;
    float4 acc = make_float4(0, 0, 0, 0);
    const size_t n = chunk4 / 256;   // iterations
    for (size_t i = 0; i < n; i += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t it = dir ? (n - 1 - (i + u)) : (i + u);
            float4 v = p[it * 256 + threadIdx.x];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
int main() {
    for (size_t mb : {128, 256, 384, 512, 768, 1024}) {
        const size_t bytes = mb << 20;
        const int blocks = 2048;
        const size_t chunk4 = bytes / 16 / blocks / 1024 * 1024;
        float4* X; float* out;
        hipMalloc(&X, bytes); hipMalloc(&out, 4); hipMemset(X, 0x3c, bytes);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float t_same = 0, t_alt = 0;
        for (int mode = 0; mode < 2; ++mode) {
            for (int w = 0; w < 4; ++w) stream<<<blocks, 256>>>(X, out, chunk4, mode ? (w & 1) : 0);
            hipDeviceSynchronize();
            hipEventRecord(a);
            const int reps = 20;
            for (int r = 0; r < reps; ++r) stream<<<blocks, 256>>>(X, out, chunk4, mode ? (r & 1) : 0);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            (mode ? t_alt : t_same) = ms / reps;
        }
        const double gb = (double)chunk4 * 16 * blocks / 1e9;
        printf("%4zu MB: same direction %.1f us (%.0f GB/s) | alternating %.1f us (%.0f GB/s)\n", mb, t_same * 1e3, gb / t_same * 1e3,
               t_alt * 1e3, gb / t_alt * 1e3);
        hipFree(X); hipFree(out);
    }
    return 0;
}
