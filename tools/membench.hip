// Micro-benchmark: read-only streaming of a (T, F, M=8) complex64 tensor with the access patterns
// the covariance / power kernels can use.  Prints GB/s per pattern.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int M = 8;
// A: lane = (bin, phase): each lane reads its own 64 contiguous bytes (4 x dwordx4), 16 bins x 4 frames per wave-step
template <int UNROLL>
__global__ __launch_bounds__(256) void pat_a(const float4* __restrict__ X, float* out, int T, int F, int tc) {
    const int tid = threadIdx.x, b = tid & 15, q = tid >> 4;
    const int f = blockIdx.x * 16 + b;
    const int t0 = blockIdx.y * tc;
    const int t1 = min(T, t0 + tc);
    float4 acc = make_float4(0, 0, 0, 0);
    const size_t fs = (size_t)F * (M / 2);
    const float4* p = X + ((size_t)(t0 + q) * F + f) * (M / 2);
    for (int t = t0 + q; t < t1; t += 16 * UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (t + 16 * u < t1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float4 v = p[u * 16 * fs + i];
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            }
        }
        p += 16 * UNROLL * fs;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
// B: fully coalesced: lane l reads 16 B at l*16 of a 1 KB run (16 bins of one frame); wave w takes frame phase
template <int UNROLL>
__global__ __launch_bounds__(256) void pat_b(const float4* __restrict__ X, float* out, int T, int F, int tc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.y * tc;
    const int t1 = min(T, t0 + tc);
    float4 acc = make_float4(0, 0, 0, 0);
    const size_t fs = (size_t)F * (M / 2);
    const float4* p = X + (size_t)(t0 + wave) * fs + (size_t)blockIdx.x * 64 + lane;
    for (int t = t0 + wave; t < t1; t += 4 * UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (t + 4 * u < t1) {
                float4 v = p[u * 4 * fs];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        p += 4 * UNROLL * fs;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
// C: coalesced, wider run: block of 256 lanes reads 4 KB contiguous (64 bins of one frame) per step
template <int UNROLL>
__global__ __launch_bounds__(256) void pat_c(const float4* __restrict__ X, float* out, int T, int F, int tc) {
    const int tid = threadIdx.x;
    const int t0 = blockIdx.y * tc;
    const int t1 = min(T, t0 + tc);
    float4 acc = make_float4(0, 0, 0, 0);
    const size_t fs = (size_t)F * (M / 2);
    const float4* p = X + (size_t)t0 * fs + (size_t)blockIdx.x * 256 + tid;
    for (int t = t0; t < t1; t += UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (t + u < t1) {
                float4 v = p[u * fs];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        p += UNROLL * fs;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
// P: the geometry of power_kernel: workgroup = 4 waves x 16 bins = 64 bins, 4 frame phases per wave, UNROLL steps of 4
//    frames in flight per wave
template <int UNROLL>
__global__ __launch_bounds__(256) void pat_p(const float4* __restrict__ X, float* out, int T, int F, int tc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = lane & 15, q = lane >> 4;
    const int f = (blockIdx.x * 4 + wave) * 16 + b;
    const int t0 = blockIdx.y * tc;
    const int t1 = min(T, t0 + tc);
    float4 acc = make_float4(0, 0, 0, 0);
    const size_t fs = (size_t)F * (M / 2);
    const float4* p = X + ((size_t)(t0 + q) * F + f) * (M / 2);
    for (int t = t0 + q; t < t1; t += 4 * UNROLL) {
        float4 v[UNROLL][4];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) v[u][i] = (t + 4 * u < t1) ? p[u * 4 * fs + i] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc.x += v[u][i].x; acc.y += v[u][i].y; acc.z += v[u][i].z; acc.w += v[u][i].w; }
        p += 4 * UNROLL * fs;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
__global__ void fill_random(unsigned* x, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u ^ (unsigned)(i >> 32);
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        x[i] = (h & 0x007fffffu) | 0x3f000000u | (h & 0x80000000u);      // random floats in +-[0.5, 1)
    }
}
// D: flat grid-stride copy-style read (the usual bandwidth ceiling probe)
__global__ __launch_bounds__(256) void pat_d(const float4* __restrict__ X, float* out, size_t n) {
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = X[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <typename Fn>
float time_it(Fn fn, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    fn(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) fn();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main(int argc, char** argv) {
    const int T = 4000, F = 2048;
    const size_t n4 = (size_t)T * F * M / 2;
    float4* X; float* out;
    CK(hipMalloc(&X, n4 * sizeof(float4))); CK(hipMalloc(&out, 4));
    if (argc > 1 && argv[1][0] == 'c') CK(hipMemset(X, 0x3c, n4 * sizeof(float4)));      // constant data
    else fill_random<<<4096, 256>>>(reinterpret_cast<unsigned*>(X), n4 * 4);             // random data (default)
    CK(hipDeviceSynchronize());
    const double gb = n4 * 16 / 1e9;
    for (int nsplit : {3, 4, 6, 12, 16, 24, 32}) {
        int tc = ((T + nsplit - 1) / nsplit + 15) / 16 * 16;
        int ns = (T + tc - 1) / tc;
        float a1 = time_it([&] { pat_a<1><<<dim3(F / 16, ns), 256>>>(X, out, T, F, tc); }, 20);
        float a2 = time_it([&] { pat_a<2><<<dim3(F / 16, ns), 256>>>(X, out, T, F, tc); }, 20);
        float a4 = time_it([&] { pat_a<4><<<dim3(F / 16, ns), 256>>>(X, out, T, F, tc); }, 20);
        float b1 = time_it([&] { pat_b<1><<<dim3(F / 16, ns), 256>>>(X, out, T, F, tc); }, 20);
        float b4 = time_it([&] { pat_b<4><<<dim3(F / 16, ns), 256>>>(X, out, T, F, tc); }, 20);
        float c4 = time_it([&] { pat_c<4><<<dim3(F / 64, ns * 4), 256>>>(X, out, T, F, (tc + 3) / 4); }, 20);
        float p2 = time_it([&] { pat_p<2><<<dim3(F / 64, ns), 256>>>(X, out, T, F, tc); }, 20);
        float p4 = time_it([&] { pat_p<4><<<dim3(F / 64, ns), 256>>>(X, out, T, F, tc); }, 20);
        printf("nsplit %2d (tc %4d): A1 %.0f  A2 %.0f  A4 %.0f | B1 %.0f  B4 %.0f | C4 %.0f | P2 %.0f  P4 %.0f GB/s\n", ns, tc, gb / a1 * 1e3,
               gb / a2 * 1e3, gb / a4 * 1e3, gb / b1 * 1e3, gb / b4 * 1e3, gb / c4 * 1e3, gb / p2 * 1e3, gb / p4 * 1e3);
    }
    for (int blocks : {1024, 2048, 4096, 8192}) {
        float d = time_it([&] { pat_d<<<blocks, 256>>>(X, out, n4); }, 20);
        printf("flat grid-stride %5d blocks: %.0f GB/s\n", blocks, gb / d * 1e3);
    }
    return 0;
}
