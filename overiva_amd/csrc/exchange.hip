// Push exchange of the per-rank partial source powers between the GPUs of a node     (reference overiva.py:152-155)
//
// The activation r[t,k] needs sum_f |y|^2 over ALL bins, i.e. one all-gather of (T, K) float32 parts per iteration
// when the bins are sharded over ranks.  The payload is tiny (128 KB per rank at 8 GPUs), so a library collective is
// pure latency: launch + protocol + synchronisation, 17 us measured for RCCL with one rank.  This is the same
// all-gather written for that regime: every rank owns a gather buffer in FINE-GRAINED device memory exported through
// hipIpcMemHandle; after its power pass a rank runs ONE kernel that stores its part into its slot of every rank's
// buffer (peer stores travel over xGMI) and whose last workgroup, after a system-scope fence, adds 1 to a counter in
// every rank's memory; a rank's stream then waits (hipStreamWaitValue32, a command-processor wait that holds no
// compute unit) until its counter says that all parts of this epoch have landed, and the activation kernel reads the
// buffer.  Buffers and counters are double-buffered by epoch parity: a rank can be at most one epoch ahead of the
// slowest reader of its stores (it cannot finish epoch e + 1 before every rank has pushed e + 1, which a rank does only
// after it has consumed epoch e).
//
// Result layout = rank-major concatenation of the parts = what all_gather_into_tensor produces, so the consumers do
// not know which transport ran.  The host side (overiva_amd/exchange.py) checks the transport against
// torch.distributed's all-gather with host-side time-outs before it is used, and falls back to the collective.
#include "oiva_internal.h"

#include <algorithm>
#include <cstring>
#include <ctime>

extern "C" {
struct oiva_xchg {
    int device = 0, rank = 0, world = 0;
    size_t slot_bytes = 0;       // one rank's part (slots are contiguous: the result is a plain concatenation)
    char* block = nullptr;       // fine-grained: [2][world][slot_bytes] then 2 counters (64 bytes apart)
    unsigned* ticket = nullptr;  // workgroups of the push kernel that have stored their chunk
    char* peer[OIVA_XCHG_MAX_RANKS] = {};   // every rank's block as mapped here (own = block)
    bool opened[OIVA_XCHG_MAX_RANKS] = {};
    bool connected = false;
};
}

namespace oiva {
namespace {

constexpr size_t kCounterStride = 64;

struct PushArgs {
    char* peer[OIVA_XCHG_MAX_RANKS];
};

// the counters sit behind the two gather buffers, on a 64-byte boundary
__host__ __device__ inline size_t buffers_bytes(int world, size_t slot) { return ((size_t)2 * world * slot + 63) & ~(size_t)63; }

// grid (chunks, world): workgroup (c, s) stores chunk c of the part into slot `rank` of rank s's buffer of this parity
template <typename V>   // float4 when the part is a whole number of 16-byte pieces, else float
__global__ __launch_bounds__(kBlock) void push_kernel(PushArgs a, const V* __restrict__ part, unsigned* ticket, int rank, int world,
                                                      long long slot_bytes, long long n, int parity) {
    const int s = blockIdx.y;
    V* dst = reinterpret_cast<V*>(a.peer[s] + ((size_t)parity * world + rank) * (size_t)slot_bytes);
    for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < n; e += (long long)gridDim.x * kBlock) dst[e] = part[e];
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();                                   // this workgroup's stores have reached their rank
        if (atomicAdd(ticket, 1u) == gridDim.x * gridDim.y - 1) {  // last one: every store of this push has
            *ticket = 0;
            const size_t counters = buffers_bytes(world, (size_t)slot_bytes);
            for (int r = 0; r < world; ++r)
                __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(a.peer[r] + counters + parity * kCounterStride), 1u,
                                       __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace
}  // namespace oiva

using namespace oiva;

// the mapped gather buffers of a connected exchange, for the X-resident kernel's in-kernel exchange (plan.hip)
int oiva::xchg_peers(oiva_xchg* x, char** peers, int* rank, int* world, size_t* slot_bytes) {
    if (!x || !(x->connected || x->world == 1)) return -1;
    for (int r = 0; r < OIVA_XCHG_MAX_RANKS; ++r) peers[r] = r < x->world ? x->peer[r] : nullptr;
    *rank = x->rank;
    *world = x->world;
    *slot_bytes = x->slot_bytes;
    return 0;
}

#define XNEED(cond, code, msg) \
    do {                       \
        if (!(cond)) return fail_with(code, msg); \
    } while (0)
#define XHIP(expr)                                                                                       \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return fail_with(OIVA_ERR_HIP, std::string(#expr ": ") + hipGetErrorString(e_)); \
    } while (0)

extern "C" {

int oiva_xchg_create(oiva_xchg** out, int device, int rank, int world, long long part_bytes) {
    XNEED(out, OIVA_ERR_ARG, "null output");
    XNEED(world >= 1 && world <= OIVA_XCHG_MAX_RANKS && rank >= 0 && rank < world && part_bytes > 0 && part_bytes % 4 == 0,
          OIVA_ERR_ARG, "bad rank / world / part size (a whole number of floats)");
    static_assert(sizeof(hipIpcMemHandle_t) == OIVA_XCHG_HANDLE_BYTES, "handle size");
    XHIP(hipSetDevice(device));
    auto* x = new oiva_xchg;
    x->device = device;
    x->rank = rank;
    x->world = world;
    x->slot_bytes = (size_t)part_bytes;
    const size_t total = buffers_bytes(world, x->slot_bytes) + 2 * kCounterStride;
    hipError_t e = hipExtMallocWithFlags((void**)&x->block, total, hipDeviceMallocFinegrained);
    if (e == hipSuccess) e = hipMemset(x->block, 0, total);
    if (e == hipSuccess) e = hipMalloc((void**)&x->ticket, sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(x->ticket, 0, sizeof(unsigned));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        if (x->block) (void)hipFree(x->block);
        if (x->ticket) (void)hipFree(x->ticket);
        delete x;
        return fail_with(OIVA_ERR_HIP, std::string("exchange allocation: ") + hipGetErrorString(e));
    }
    x->peer[rank] = x->block;
    *out = x;
    return OIVA_OK;
}

int oiva_xchg_export(oiva_xchg* x, void* handle) {
    XNEED(x && handle, OIVA_ERR_ARG, "null argument");
    XHIP(hipSetDevice(x->device));
    hipIpcMemHandle_t h;
    XHIP(hipIpcGetMemHandle(&h, x->block));
    std::memcpy(handle, &h, sizeof(h));
    return OIVA_OK;
}

int oiva_xchg_connect(oiva_xchg* x, const void* handles) {
    XNEED(x && handles, OIVA_ERR_ARG, "null argument");
    XNEED(!x->connected, OIVA_ERR_STATE, "already connected");
    XHIP(hipSetDevice(x->device));
    for (int r = 0; r < x->world; ++r) {
        if (r == x->rank) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, static_cast<const char*>(handles) + (size_t)r * sizeof(h), sizeof(h));
        void* p = nullptr;
        XHIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        x->peer[r] = static_cast<char*>(p);
        x->opened[r] = true;
    }
    x->connected = true;
    return OIVA_OK;
}

int oiva_xchg_gathered(oiva_xchg* x, int epoch, void** gathered) {
    XNEED(x && gathered && epoch >= 1, OIVA_ERR_ARG, "bad arguments");
    *gathered = x->block + (size_t)(epoch & 1) * x->world * x->slot_bytes;
    return OIVA_OK;
}

int oiva_xchg_push(oiva_xchg* x, void* stream, const void* part_dev, long long part_bytes, int epoch) {
    XNEED(x && part_dev && epoch >= 1, OIVA_ERR_ARG, "bad arguments");
    XNEED(x->connected || x->world == 1, OIVA_ERR_STATE, "exchange not connected");
    XNEED((size_t)part_bytes == x->slot_bytes && ((uintptr_t)part_dev & 15) == 0, OIVA_ERR_ARG,
          "part must be 16-byte aligned and of the size given at creation");
    XHIP(hipSetDevice(x->device));
    PushArgs a;
    for (int r = 0; r < OIVA_XCHG_MAX_RANKS; ++r) a.peer[r] = r < x->world ? x->peer[r] : nullptr;
    const bool vec = part_bytes % 16 == 0;
    const long long n = vec ? part_bytes / 16 : part_bytes / 4;
    const unsigned chunks = (unsigned)std::min<long long>(std::max<long long>((n + 4 * kBlock - 1) / (4 * kBlock), 1), 64);
    const dim3 grid(chunks, (unsigned)x->world);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long slot4 = (long long)x->slot_bytes;
    if (vec)
        hipLaunchKernelGGL(push_kernel<float4>, grid, dim3(kBlock), 0, st, a, static_cast<const float4*>(part_dev), x->ticket, x->rank,
                           x->world, slot4, n, epoch & 1);
    else
        hipLaunchKernelGGL(push_kernel<float>, grid, dim3(kBlock), 0, st, a, static_cast<const float*>(part_dev), x->ticket, x->rank,
                           x->world, slot4, n, epoch & 1);
    XHIP(hipGetLastError());
    return OIVA_OK;
}

static unsigned expected_count(const oiva_xchg* x, int epoch) { return (unsigned)((epoch + 1) >> 1) * (unsigned)x->world; }

int oiva_xchg_wait(oiva_xchg* x, void* stream, int epoch) {
    XNEED(x && epoch >= 1, OIVA_ERR_ARG, "bad arguments");
    XHIP(hipSetDevice(x->device));
    unsigned* counter = reinterpret_cast<unsigned*>(x->block + buffers_bytes(x->world, x->slot_bytes) + (epoch & 1) * kCounterStride);
    XHIP(hipStreamWaitValue32(static_cast<hipStream_t>(stream), counter, expected_count(x, epoch), hipStreamWaitValueGte, 0xffffffffu));
    return OIVA_OK;
}

int oiva_xchg_poll(oiva_xchg* x, int epoch, int timeout_ms, int* arrived) {
    XNEED(x && arrived && epoch >= 1, OIVA_ERR_ARG, "bad arguments");
    XHIP(hipSetDevice(x->device));
    const unsigned* counter =
        reinterpret_cast<const unsigned*>(x->block + buffers_bytes(x->world, x->slot_bytes) + (epoch & 1) * kCounterStride);
    *arrived = 0;
    for (int waited = 0;; ++waited) {
        unsigned v = 0;
        XHIP(hipMemcpy(&v, counter, sizeof(v), hipMemcpyDeviceToHost));
        if (v >= expected_count(x, epoch)) {
            *arrived = 1;
            return OIVA_OK;
        }
        if (waited >= timeout_ms) return OIVA_OK;
        struct timespec ts = {0, 1000000};
        nanosleep(&ts, nullptr);
    }
}

int oiva_xchg_force(oiva_xchg* x, int epoch) {
    XNEED(x && epoch >= 1, OIVA_ERR_ARG, "bad arguments");
    XHIP(hipSetDevice(x->device));
    // a host store of the value the stream waits for, through a stream of its own (the waiting stream is blocked, and
    // nothing here may synchronise with it)
    unsigned* counter = reinterpret_cast<unsigned*>(x->block + buffers_bytes(x->world, x->slot_bytes) + (epoch & 1) * kCounterStride);
    const unsigned v = expected_count(x, epoch);
    hipStream_t s = nullptr;
    XHIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipError_t e = hipMemcpyAsync(counter, &v, sizeof(v), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipStreamDestroy(s);
    XHIP(e);
    return OIVA_OK;
}

int oiva_xchg_destroy(oiva_xchg* x) {
    if (!x) return OIVA_OK;
    (void)hipSetDevice(x->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < x->world; ++r)
        if (x->opened[r]) (void)hipIpcCloseMemHandle(x->peer[r]);
    if (x->block) (void)hipFree(x->block);
    if (x->ticket) (void)hipFree(x->ticket);
    delete x;
    return OIVA_OK;
}

}  // extern "C"
