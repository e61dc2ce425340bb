// Probe: can hipStreamWaitValue32 / hipStreamWriteValue32 be captured into a hipGraph on this ROCm?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* p) { if (threadIdx.x == 0) atomicAdd(p, 1u); }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned* flag; hipMalloc(&flag, 8); hipMemset(flag, 0, 8);
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    printf("begin capture: %s\n", hipGetErrorString(e));
    k<<<1, 64, 0, s>>>(flag);
    e = hipStreamWriteValue32(s, flag + 1, 5, 0);
    printf("write value in capture: %s\n", hipGetErrorString(e));
    e = hipStreamWaitValue32(s, flag + 1, 5, hipStreamWaitValueGte, 0xffffffffu);
    printf("wait value in capture: %s\n", hipGetErrorString(e));
    k<<<1, 64, 0, s>>>(flag);
    e = hipStreamEndCapture(s, &g);
    printf("end capture: %s\n", hipGetErrorString(e));
    if (e == hipSuccess && g) {
        e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        printf("instantiate: %s\n", hipGetErrorString(e));
        if (e == hipSuccess) {
            e = hipGraphLaunch(ge, s); printf("launch: %s\n", hipGetErrorString(e));
            e = hipStreamSynchronize(s); printf("sync: %s\n", hipGetErrorString(e));
            unsigned h[2]; hipMemcpy(h, flag, 8, hipMemcpyDeviceToHost); printf("counter %u flag %u\n", h[0], h[1]);
        }
    }
    return 0;
}
