// Feasibility probe for a push-style exchange between processes: process A exports a FINE-GRAINED device buffer
// (data + counter) through hipIpcMemHandle; process B opens it, and ONE kernel writes the data and then -- last
// workgroup, after a system-scope fence -- adds 1 to the counter; A waits for the counter with hipStreamWaitValue32
// and reads the data with a kernel.   usage: ipc_probe a <dir> | ipc_probe b <dir>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void push(float* dst, unsigned* counter, unsigned* ticket, int n, float v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = v + i;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
            *ticket = 0;
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void sum(const float* p, int n, double* out) {
    double s = 0;
    for (int i = threadIdx.x; i < n; i += 256) s += p[i];
    atomicAdd(out, s);
}
static bool exists(const std::string& f) { return access(f.c_str(), F_OK) == 0; }
int main(int argc, char** argv) {
    const std::string dir = argv[2];
    const int n = 1 << 16;
    CK(hipSetDevice(0));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (argv[1][0] == 'a') {
        float* buf = nullptr;
        hipError_t ea = hipExtMallocWithFlags((void**)&buf, n * 4 + 256, hipDeviceMallocFinegrained);
        printf("A: hipExtMallocWithFlags(finegrained) -> %s\n", hipGetErrorString(ea));
        if (ea != hipSuccess) return 4;
        CK(hipMemset(buf, 0, n * 4 + 256));
        hipIpcMemHandle_t h;
        hipError_t eh = hipIpcGetMemHandle(&h, buf);
        printf("A: hipIpcGetMemHandle(finegrained) -> %s\n", hipGetErrorString(eh));
        if (eh != hipSuccess) return 5;
        FILE* f = fopen((dir + "/handle.tmp").c_str(), "wb"); fwrite(&h, sizeof(h), 1, f); fclose(f);
        rename((dir + "/handle.tmp").c_str(), (dir + "/handle").c_str());
        unsigned* counter = reinterpret_cast<unsigned*>(buf + n);
        double* out; CK(hipMalloc(&out, 8)); CK(hipMemset(out, 0, 8));
        for (int round = 1; round <= 3; ++round) {
            CK(hipStreamWaitValue32(s, counter, round, hipStreamWaitValueGte, 0xffffffffu));
            sum<<<1, 256, 0, s>>>(buf, n, out);
        }
        CK(hipStreamSynchronize(s));
        double host; CK(hipMemcpy(&host, out, 8, hipMemcpyDeviceToHost));
        printf("A: sum over 3 rounds %.1f (round values differ; nonzero expected)\n", host);
        FILE* d = fopen((dir + "/done").c_str(), "w"); fclose(d);
        return 0;
    }
    while (!exists(dir + "/handle")) usleep(1000);
    hipIpcMemHandle_t h; FILE* f = fopen((dir + "/handle").c_str(), "rb"); fread(&h, sizeof(h), 1, f); fclose(f);
    void* peer = nullptr; CK(hipIpcOpenMemHandle(&peer, h, hipIpcMemLazyEnablePeerAccess));
    float* pb = static_cast<float*>(peer);
    unsigned* ticket; CK(hipMalloc(&ticket, 4)); CK(hipMemset(ticket, 0, 4));
    usleep(100000);
    for (int round = 1; round <= 3; ++round) {
        push<<<n / 256, 256, 0, s>>>(pb, reinterpret_cast<unsigned*>(pb + n), ticket, n, 1.0f * round);
        CK(hipStreamSynchronize(s));
        usleep(20000);
    }
    while (!exists(dir + "/done")) usleep(1000);
    CK(hipIpcCloseMemHandle(peer));
    printf("B: done\n");
    return 0;
}
