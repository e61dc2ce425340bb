// C ABI of liboveriva_hip.so (see include/overiva_hip.h): plan life cycle, prologue, iteration,
// epilogue, measurement and test-only stage access.  Host code only; kernels live in kernels_*.hip.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "host_io.h"
#include "oiva_internal.h"
#include "resident.h"

using namespace oiva;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

}  // namespace

int oiva::fail_with(int code, const std::string& msg) { return fail(code, msg); }
oiva::KernelTimer& oiva::kernel_timer() {
    static thread_local KernelTimer t;
    return t;
}

namespace {

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(OIVA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define NEED(cond, code, msg) \
    do {                      \
        if (!(cond)) return fail(code, msg); \
    } while (0)

int round_up(int a, int b) { return (a + b - 1) / b * b; }
int ceil_div(int a, int b) { return (a + b - 1) / b; }

constexpr int kGraphBatch = 8;       // iterations of the graph captured ahead of time (oiva_plan_use_graph)
constexpr int kGraphMaxIters = 32;   // longest graph captured on demand: an iterate(n) call is ceil(n / 32) replays
constexpr int kGraphCache = 6;       // captured lengths kept per plan

}  // namespace

// ---- large device buffers (X, Y, staging) come from a process-wide pool -------------------------------------------
// The drop-in call creates and destroys a plan per call; hipMalloc + hipFree of the 524 MB of X and the 131 MB of Y at the
// headline shape were ~2.5 ms of a 22 ms call.  Buffers of >= 16 MB go back to the pool instead of the driver (exact-size
// reuse, per device), at most $OIVA_POOL_MB (default 2048) held; oiva_pool_trim() releases them.
namespace {
struct BigPool {
    struct Entry {
        void* p;
        size_t bytes;
        int dev;
    };
    std::mutex m;
    std::vector<Entry> held;      // oldest first
    size_t total = 0;
};
BigPool& big_pool() {
    static BigPool* bp = new BigPool;      // (never destroyed: the HIP runtime may be gone when static destructors run)
    return *bp;
}
constexpr size_t kPoolMinBytes = (size_t)16 << 20;
size_t pool_cap_bytes() {
    static const size_t cap = [] {
        const char* v = std::getenv("OIVA_POOL_MB");
        return (size_t)(v ? std::max(0, std::atoi(v)) : 2048) << 20;
    }();
    return cap;
}
// every buffer the pool holds goes back to the driver (oiva_pool_trim; dev_malloc when the driver is out of memory)
void pool_release_all() {
    BigPool& bp = big_pool();
    std::lock_guard<std::mutex> g(bp.m);
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (auto& e : bp.held) {
        (void)hipSetDevice(e.dev);
        (void)hipFree(e.p);
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    bp.held.clear();
    bp.total = 0;
}
// hipMalloc of the library: the pool's idle buffers are memory the caller thinks is free, so before an allocation fails for
// want of memory they are handed back and the allocation is tried once more
hipError_t dev_malloc(void** out, size_t bytes) {
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipErrorOutOfMemory) return e;
    (void)hipGetLastError();
    pool_release_all();
    return hipMalloc(out, bytes);
}
template <class P>
hipError_t dev_malloc(P** out, size_t bytes) {
    return dev_malloc(reinterpret_cast<void**>(out), bytes);
}
hipError_t big_alloc(int dev, void** out, size_t bytes) {
    if (bytes >= kPoolMinBytes) {
        BigPool& bp = big_pool();
        std::lock_guard<std::mutex> g(bp.m);
        for (size_t i = bp.held.size(); i-- > 0;)
            if (bp.held[i].dev == dev && bp.held[i].bytes == bytes) {
                *out = bp.held[i].p;
                bp.total -= bytes;
                bp.held.erase(bp.held.begin() + (long)i);
                return hipSuccess;
            }
    }
    return dev_malloc(out, bytes);
}
void big_free(int dev, void* ptr, size_t bytes) {
    if (!ptr) return;
    if (bytes >= kPoolMinBytes && bytes <= pool_cap_bytes()) {
        BigPool& bp = big_pool();
        std::lock_guard<std::mutex> g(bp.m);
        while (!bp.held.empty() && bp.total + bytes > pool_cap_bytes()) {
            (void)hipFree(bp.held.front().p);
            bp.total -= bp.held.front().bytes;
            bp.held.erase(bp.held.begin());
        }
        bp.held.push_back({ptr, bytes, dev});
        bp.total += bytes;
        return;
    }
    (void)hipFree(ptr);
}
}  // namespace

struct oiva_plan {
    int device = 0;
    int T = 0, F = 0, M = 0, K = 0, model = 0, F_total = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    const float2* X = nullptr;  // (T,F,M)
    float2* X_owned = nullptr;
    size_t x_owned_bytes = 0, y_bytes = 0;      // (big_alloc / big_free)
    float2* X_pad = nullptr;    // (T, F, M + 1): X with a zero channel behind every bin's M, for the vector-ALU covariance kernels at 9/11/13/15 channels
    float2* What = nullptr;     // (F,M,M) complex64: what the streaming kernels read
    double2* What64 = nullptr;  // (F,M,M) complex128: carried between iterations by the float64 update
    bool what64_valid = false;
    double* Cx = nullptr;       // [F][M*M] packed, / T
    void* Vpart = nullptr;      // [nsplit][F][K][M*M] packed partial sums, float32 or float64 (cov_f64())
    float* Ppart = nullptr;     // [nb (or more, zero padded)][T][K]
    int ppart_alloc = 0;
    float* Plocal = nullptr;    // (T,K)
    float* R = nullptr;         // (T,K)
    float* wscale = nullptr;    // (K)
    float* Spart = nullptr;     // [nsplit][F][K][3]
    float2* Y = nullptr;        // (T,F,K), allocated on first demix
    double2* scratch_c = nullptr;  // K*F*M*M complex128, for getters
    double* scratch_p = nullptr;   // K*F*M*M packed float64, for getters

    CovGeom cov{};
    bool cov_quad_on = true;      // oiva_plan_set_cov_quad
    bool cov_hmfma_on = true;     // oiva_plan_set_cov_hmfma ($OIVA_COV_HMFMA=0: off)
    bool fuse_cov_update = false; // oiva_plan_set_fuse_cov_update ($OIVA_COV_UPDATE=1: on wherever the shape qualifies)
    CovGeom stg{};              // geometry of the projection-back statistics pass (16-bin groups, independent of cov)
    PowGeom pw{};
    int n_cu = 256;
    int vpart_splits_alloc = 0;

    bool have_x = false, have_cx = false, have_w = false;
    bool pad_valid = false;       // X_pad holds the current X
    bool wscale_pending = false;  // wscale computed (by the covariance pass), update not yet applied
    int raw_weights = 0;          // test hook: R holds final 1/weights, no gamma normalisation
    int prec = 0;                 // OIVA_PREC_* bits (oiva_plan_set_precision)
    bool upd_f64() const { return prec & OIVA_PREC_UPDATE_F64; }
    bool cov_f64() const { return prec & OIVA_PREC_COV_F64; }
    // element type of the covariance partials: the vector-ALU kernels (<= 8 channels) always sum their float32 lane
    // chains across lanes in float64 and store float64 partials; the 9..16-channel matrix-core kernel stores its
    // accumulator type
    bool vpart_f64() const { return vpart_f64_of(cov); }
    bool vpart_f64_of(const CovGeom& g) const { return cov_f64() || M <= 8 || g.quad || (g.half16 && !g.part32); }
    int use_graph = 0;
    // OGIVE (ive.py): per-bin state, allocated by oiva_plan_ogive_begin
    OgiveState og{};
    std::vector<void*> og_bufs;
    bool og_ready = false;
    int og_mode = 0, og_model = 0;
    // captured chunk of OGIVE epochs (five launches per epoch are latency bound on the reference's problem sizes)
    hipGraphExec_t og_graph = nullptr;
    int og_graph_n = 0, og_graph_phase = -1;
    double og_graph_mu = 0., og_graph_tol = 0.;
    // X-resident iteration (resident.h): geometry, exchange buffers, epoch of the last launch
    ResidentGeom rg{};
    bool res_ok = false;           // the shape qualifies
    bool res_on = false;
    void* res_block = nullptr;     // one allocation: parts | vpart | rsum | wpub | flags | ctrl | stamps
    float* res_parts = nullptr;
    float* res_psum = nullptr;
    double* res_vpart = nullptr;
    double* res_rsum = nullptr;
    float2* res_wpub = nullptr;
    unsigned* res_flags = nullptr; // ctrl words ([0] give-up code)
    float2* res_what = nullptr;    // second copies of W_hat / its complex128 form: a launch writes its final state there, and the
    double2* res_what64 = nullptr; //   plan swaps them with What / What64 when nobody gave up (own allocations)
    unsigned* res_code_host = nullptr;   // pinned: the give-up code of a launch, copied behind it on the stream
    unsigned long long* res_stamps = nullptr;
    size_t res_block_bytes = 0;
    unsigned res_epoch = 0;
    int res_last_code = 0, res_launches = 0, res_fallbacks = 0, res_stamped = 0;
    int res_timeout_ms = 0, res_stall = -1, res_stall_iter = 0;
    char* res_loop_buf = nullptr;  // loop-back: this plan plays all res_world ranks on one GPU (oiva_plan_resident_loopback)
    bool res_loopback = false;
    bool res_trace = false;        // timestamps of every workgroup (oiva_plan_resident_trace)
    unsigned long long* res_trace_buf = nullptr;
    int res_trace_iters = 0;
    char* res_gath[OIVA_XCHG_MAX_RANKS] = {};   // every rank's gather buffer (bins sharded over GPUs), else unused
    int res_rank = 0, res_world = 1;
    // exchange of the ranks' partial powers inside the activation kernel of the four-launch path (oiva_plan_fused_connect)
    bool fx_on = false, fx_loopback = false;
    char* fx_gath[OIVA_XCHG_MAX_RANKS] = {};
    int fx_rank = 0, fx_world = 1;
    int fx_nblk_own = 1, fx_nblk_peer = 1;      // block sums this rank forms / words per other rank's slot
    char* fx_loop_buf = nullptr;                // loop-back: this plan's own gather buffer
    unsigned* fx_state = nullptr;               // [0] give-up flag, [16 ...] one epoch counter per workgroup of the activation kernel
    unsigned* fx_flag_host = nullptr;           // pinned: fx_state[0] as copied behind the work on the plan's stream (check_fused)
    int fx_timeout_ms = 0, fx_stall = 0;
    // hand-over of Y to a host array in slabs of frames (demix_to_host): a second stream for the device-to-host copies, one
    // event pair per slot of the pinned ring, a device staging ring for complex128 output
    hipStream_t io_stream = nullptr;
    hipEvent_t io_written[kHostRingSlots] = {}, io_copied[kHostRingSlots] = {};
    double2* io_c128[kHostRingSlots] = {};
    size_t io_c128_bytes = 0;
    size_t io_slab_bytes = 0;                   // oiva_plan_set_io_slab (0: 8 MB)
    float2* ck_what = nullptr;                  // oiva_plan_save_w: a copy of W_hat (and of its complex128 form) on the device
    double2* ck_what64 = nullptr;
    bool ck_valid = false, ck_what64_valid = false;
    std::vector<std::pair<int, hipGraphExec_t>> graphs;   // (iterations, executable graph), most recently used last
    hipEvent_t ev[2] = {};
};

namespace {

// Frame splits are chosen so that the grid is a whole number of "rounds" of what the chip holds at
// once (CUs x resident workgroups per CU): a grid of 1.5 rounds runs as long as one of 2.
int pick_splits(int capacity, int blocks_per_split, int T, int min_frames) {
    int ns = std::max(1, capacity / std::max(1, blocks_per_split));
    ns = std::min(ns, std::max(1, T / std::max(1, min_frames)));   // every split keeps >= min_frames frames
    return ns;
}

// Kernels with four frame phases per workgroup, float64 per-bin algebra behind them (`mixed`): 8 frame splits on a long
// frame axis, 4 on a short one (their float32 chains are then T / 32 resp. T / 16 frames); on a short axis never more than
// one round of workgroups (a workgroup's fixed costs dominate there) or splits of fewer than 16 frames.
int mixed_min_splits(int capacity, int blocks_per_split, int T) {
    if (T >= 1024) return 8;       // (a second round of workgroups costs little there: +18 us of 270 at 2048 x 4000 x 16 / 2)
    const int one_round = std::max(1, capacity / std::max(1, blocks_per_split));
    return std::max(1, std::min(std::min(4, one_round), T / 16));
}

void choose_cov_geom(oiva_plan* p, int nsplit_req) {
    CovGeom g;
    // float64, 8 channels: two lanes per (bin, frame), 32 bins per workgroup (kernels_cov_pair64.hip); else 16 bins
    // (same geometry, float32: 8 channels with three or more sources, four per pass -- kernels_cov_pair32.hip)
    g.pair32 = !p->cov_f64() && cov_pair32_supported(p->M, p->K);
    const bool pair = g.pair32 || (p->cov_f64() && cov_pair64_supported(p->M));
    g.nbg = ceil_div(p->F, pair ? cov_pair64_bins_per_block() : kBinsPerWave);
    g.kc = cov_sources_per_pass(p->M, p->K, p->cov_f64(), p->T < 1024);
    const int nz = ceil_div(p->K, g.kc);
    int nsplit = nsplit_req;
    // few sources: the Hermitian half on the vector ALU, four lanes per (bin, frame).  One or two sources are one pass
    // over X and faster than the matrix-core kernel in any mode; three or four are two passes, slower than its float32
    // form (measured 486-517 against 459 us at 2048 x 4000 x 16) but with float64 partial sums of short float32 chains,
    // which is what the float64 per-bin algebra of the `mixed` mode needs -- and 1.8 times faster than the float64
    // matrix-core pass that mode would otherwise have to be replaced by
    // (9, 11, 13, 15 channels: the same kernels on the copy of X padded by one zero channel, plan_covariance)
    const int Mc = p->M + (p->M > 8 && p->M % 2 && p->X_pad != nullptr ? 1 : 0);
    g.pad = Mc != p->M;
    if (p->M > 8 && !p->cov_f64() && p->cov_quad_on && cov_quad_supported(Mc, p->K) && (p->K <= 2 || p->upd_f64())) {
        // one round of two workgroups per CU
        g.quad = 1;
        g.kc = cov_quad_sources_per_pass(p->K);
        if (nsplit <= 0) {
            const int blocks = g.nbg * ceil_div(p->K, g.kc);
            nsplit = std::min(32, pick_splits(p->n_cu * 2, blocks, p->T, p->T >= 1024 ? 128 : 64));
            // only 4 frame phases per workgroup: a lane's float32 chain is T / (4 nsplit) frames, four times that of the
            // 8-channel kernel at equal splits, and the error of the result grows linearly with it (measured against
            // the reference's own complex64 floor, 16 channels / 2 sources x 20 iterations: T = 4000: 4 splits 0.8-1.0
            // floors, 8 splits 0.5-0.6, 16 splits 0.3; T = 163: 1 split 1.4, 4 splits 0.8, 8 splits 0.6).  With the
            // float64 per-bin algebra (`mixed`, the default arithmetic of these shapes) the chains are what is left of
            // the error, so that mode takes 8 splits (+18 us on the pass, +8 us in the update at 2048 x 4000 x 16 / 2).
            // ... as long as that is still one round of workgroups (few frames: 4 splits = chains of T / 16, 0.8 floors)
            if (p->upd_f64()) nsplit = std::max(nsplit, mixed_min_splits(p->n_cu * 2, blocks, p->T));
        }
        g.tc = round_up(ceil_div(p->T, nsplit), 8);
        g.nsplit = ceil_div(p->T, g.tc);
        p->cov = g;
        return;
    }
    // many sources (5..16): the Hermitian half over 32 lanes per (bin, frame), every source in one pass; a wave's float32
    // chain is T / (4 nsplit) frames: <= 256 in `fast`, <= 128 with the float64 per-bin algebra behind it
    // (float64 sums, `precise`, 3..16 sources: the same lanes with four or eight sources per pass; splits only to fill the chip)
    if (p->M > 8 && p->cov_quad_on &&
        (p->cov_f64() ? cov_half16_f64_supported(Mc, p->K) : p->K > 4 && cov_half16_supported(Mc, p->K))) {
        g.half16 = 1;
        g.hmfma = (p->cov_hmfma_on && (p->cov_f64() ? cov_hmfma64_supported(Mc, p->K) : cov_hmfma_supported(Mc, p->K))) ? 1 : 0;
        g.nbg = ceil_div(p->F, 2);
        g.kc = p->cov_f64() ? cov_half16_f64_sources_per_pass(p->K) : cov_half16_sources_per_pass(p->K);
        if (nsplit <= 0) {
            // (float64: no chain to bound; the four-source form runs a little faster in two rounds of workgroups -- 2048 x 4000
            //  x 16 / 4: 1 split 744 us, 2 splits 691, 4 splits 693; the eight-source form does not care)
            // (the matrix-core kernel: one bin per workgroup and eight float32 chains -- half the splits for the same chain)
            const int chains = g.hmfma ? 8 : 4;
            nsplit = p->cov_f64() ? (!g.hmfma && g.kc == 4 && p->T >= 1024 ? 2 : 1) : ceil_div(p->T, chains * (p->upd_f64() ? 128 : 256));
            const int per_cu = g.hmfma ? (p->cov_f64() ? 2 : 3) : p->cov_f64() && g.kc == 4 ? 4 : 2;      // workgroups a CU holds (registers / launch bounds)
            const int groups = g.hmfma ? p->F : g.nbg * ceil_div(p->K, g.kc);
            while (groups * nsplit < per_cu * p->n_cu && ceil_div(p->T, nsplit + 1) >= 64) ++nsplit;
        }
        g.tc = round_up(ceil_div(p->T, nsplit), g.hmfma ? 32 : 16);
        g.nsplit = ceil_div(p->T, g.tc);
        // (round 6, opt-in: $OIVA_HMFMA_PART32=1, read whenever the geometry is chosen) the float32 matrix-core kernel hands its
        // partial blocks over as float32 -- half the bytes the per-bin update is bound by at 16 x 16 (update 66 -> 50 us, iteration
        // 931 -> 903 us) for one more rounding per block: W moves by 1e-8 on i.i.d. input and by 3e-6 .. 1e-5 (0.3-0.5 reference
        // floors) on a 16-source mixture at full size, and the fixtures of <= 1024 frames land up to twice as far from the
        // complex128 result (tools/r6/part32_ab.py, tools/r6/NOTES.md).  Not the default: parity first.
        if (g.hmfma && !p->cov_f64()) {
            const char* pv = std::getenv("OIVA_HMFMA_PART32");
            g.part32 = pv && pv[0] == '1';
        }
        p->cov = g;
        return;
    }
    if (p->M > 8) {
        // planar matrix-core path: one wave per (bin, split); splits bound the length of the fp32
        // accumulation chain (<= 512 frames) and keep >= 2 waves per SIMD when there are few bins
        if (nsplit <= 0) {
            nsplit = ceil_div(p->T, 512);
            const int waves = p->F * ceil_div(p->K, 4);
            while (waves * nsplit < 2 * 4 * p->n_cu && ceil_div(p->T, nsplit + 1) >= 64) ++nsplit;
        }
        g.tc = round_up(ceil_div(p->T, nsplit), 4);
        g.nsplit = ceil_div(p->T, g.tc);
        p->cov = g;
        return;
    }
    const int quantum = pair ? 8 : 16;      // frames per step of a workgroup
    if (nsplit <= 0) {
        int bpc = 2;
        if (cov_blocks_per_cu(p->M, g.kc, p->cov_f64(), &bpc) != hipSuccess || bpc < 1) bpc = 2;
        // one round: the grid is what the chip holds at once (CUs x resident workgroups); every workgroup pays a
        // fixed cost (gamma prologue, ring fill, epilogue), so fewer, longer workgroups win as long as the chip
        // is full, and 1.5 rounds run as long as 2
        // (on a short frame axis at least 64 frames per split instead of 128: at the reference's 160-235 frames the floor of
        //  128 left the chip to one split -- 2049 x 235 x 8 / 2: 1 split 18.9 / 29.2 us (float32 / float64), 3 splits 14.7 /
        //  16.8.  On a long axis the floor of 128 stays: a 256-bin shard of 4000 frames takes 28 splits in 20.4 us, 32 splits
        //  -- exactly the chip's 512 workgroup slots, which the dispatcher does not fill evenly -- 26.3)
        nsplit = pick_splits(p->n_cu * bpc, g.nbg * nz, p->T, p->T >= 1024 ? 128 : 64);
        // a grid that fills the chip's workgroup slots EXACTLY runs slower than one an eighth short of it (the dispatcher does
        // not fill the CUs evenly): 512 bins x 4000 frames, 16 splits = 512 workgroups 31.4 us, 14 splits 29.5 us; 256 bins: 32
        // splits 26.3 us, 28 splits 20.4 us
        if (nsplit >= 12 && g.nbg * nz * nsplit >= p->n_cu * bpc) nsplit = nsplit * 7 / 8;
        // the update kernel adds the nsplit partials of every matrix element in one round of loads per 16 splits
        // (sum_vpart); more than 32 splits cost more there than the fuller grid saves here (measured on a
        // 256-bin shard: 16 splits 25.2 + 8.0 us, 28 splits 21.4 + 9.2 us, 42 splits 25.5 + 10.3 us)
        // ... unless 32 splits would leave most of the chip idle (few bins, very long frame axis): then the
        // streaming pass dominates and up to 64 splits are allowed
        const int cap = (g.nbg * nz * 32 >= p->n_cu) ? 32 : 64;
        nsplit = std::min(nsplit, cap);
        // four frame phases per workgroup instead of 16: float32 chains four times as long at equal splits; with the
        // float64 per-bin algebra behind it the pass takes at least 8 splits (see the 10..16-channel kernel above)
        // (round 5: on a short frame axis the bound is the chain itself -- T / (4 nsplit) <= 64 frames, what 4 splits give just below
        //  1024 frames -- not 4 splits whatever T: at the reference's 2049 bins x 235 frames the forced fourth split cost 8 / 4
        //  sources 26.8 against 21.2 us on the pass (iteration 74.4 -> 69.2 us), 8 / 3 23.0 against 19.2 (63.5 -> 59.4))
        if (g.pair32 && p->upd_f64()) {
            const int by_chain = p->T >= 1024 ? 8 : std::min(4, ceil_div(p->T, 256));
            nsplit = std::max(nsplit, std::min(by_chain, mixed_min_splits(p->n_cu * bpc, g.nbg * nz, p->T)));
        }
    }
    g.tc = round_up(ceil_div(p->T, nsplit), quantum);
    g.nsplit = ceil_div(p->T, g.tc);
    p->cov = g;
}

void choose_stats_geom(oiva_plan* p) {
    CovGeom g;
    g.nbg = ceil_div(p->F, kBinsPerWave);
    g.kc = 2;
    const int nz = ceil_div(p->K, g.kc);
    const int nsplit = std::min(16, pick_splits(p->n_cu * 4, g.nbg * nz, p->T, 128));
    g.tc = round_up(ceil_div(p->T, nsplit), 16);
    g.nsplit = ceil_div(p->T, g.tc);
    p->stg = g;
}

void choose_pow_geom(oiva_plan* p, int nsplit_req) {
    PowGeom g;
    g.nb = ceil_div(p->F, kBinsPerWave * kWaves);
    g.kp = pow_sources_per_pass(p->M, p->K);
    int nsplit = nsplit_req;
    if (nsplit <= 0) {
        // Workgroups so that about 48 KB of X are in flight per CU: a wave keeps two steps (2 x 4 frames x 16 bins x 8M
        // bytes) in flight, i.e. 12 / M workgroups per CU -- 1.5 at 8 channels, 3 at 4, 0.75 at 16.  Measured at 2048 bins
        // x 4000 frames on the kernel itself and on the pure-read form of its geometry (tools/membench.hip, pattern P):
        // 8 channels 12 splits (384 workgroups) 88-90 us, 16: 91, 24: 92, 8: 106 (four steps in flight and 24 splits, as in
        // round 1: 94-96 us); 4 channels 24 splits 40.7 us, 12: 46; 2 channels 24 splits 21.8, 12: 35; 16 channels / 2
        // sources 6 splits 172 us, 12: 234.  More resident waves thrash the 32 KB L1, fewer expose HBM latency.  Each
        // workgroup loads its W first (256-bin shard: 62 splits 13.5 us, 167 splits 15.9 us).
        // (counted per source pass: with 8 sources in two passes, 12 splits 112 us, 6 splits 127-137 us)
        // ONE source per pass halves the arithmetic per byte and a wave runs through its two steps in flight before the next
        // ones arrive: twice the workgroups (8 channels / 1 source: 12 splits 111 us, 24 splits 92.5; with 2-4 sources 24
        // splits measure the same as 12)
        // (not beyond 8 channels: 16 channels / 1 source 6 splits 226 us, 12 splits -- 1.5 rounds of workgroups -- 276)
        const int per_cu_x12 = (g.kp == 1 && p->M <= 8) ? 24 : 12;
        // (at least 32 frames per workgroup: with 64, 2049 x 235 x 8 / 2 ran 3 splits in 16.0 us where 6-8 take 11.9-12.1,
        //  2049 x 160 x 4 / 2 2 splits in 12.0 us where 5-8 take 8.2-8.4)
        nsplit = pick_splits(std::max(p->n_cu / 2, p->n_cu * per_cu_x12 / std::max(p->M, 1)), g.nb, p->T, 32);
    }
    int tcp = round_up(ceil_div(p->T, nsplit), 4);
    tcp = std::min(std::max(tcp, 4), kPowMaxFrames);
    g.tcp = tcp;
    g.nsplit = ceil_div(p->T, tcp);
    p->pw = g;
}

int drop_graph(oiva_plan* p) {
    while (!p->graphs.empty()) {
        hipGraphExec_t g = p->graphs.back().second;
        p->graphs.pop_back();
        HIP_TRY(hipGraphExecDestroy(g));
    }
    if (p->og_graph) {
        HIP_TRY(hipGraphExecDestroy(p->og_graph));
        p->og_graph = nullptr;
    }
    return OIVA_OK;
}

size_t vpart_floats(const oiva_plan* p, int nsplit) { return (size_t)nsplit * p->F * p->K * p->M * p->M; }

int ensure_vpart(oiva_plan* p) {
    if (p->cov.nsplit > p->vpart_splits_alloc) {
        if (p->Vpart) HIP_TRY(hipFree(p->Vpart));
        p->Vpart = nullptr;
        HIP_TRY(dev_malloc(&p->Vpart, (vpart_floats(p, p->cov.nsplit) + 2) * sizeof(double)));   // either element type; sum_vpart reads idx + 1
        p->vpart_splits_alloc = p->cov.nsplit;
    }
    return OIVA_OK;
}

// ---- the five stages of one iteration -----------------------------------------------------------
int stage_power(oiva_plan* p) {
    HIP_TRY(launch_power(p->stream, p->X, p->X_pad && p->pad_valid ? p->X_pad : nullptr, p->What, p->Ppart, p->T, p->F, p->M, p->K, p->pw));
    return OIVA_OK;
}
int stage_activation(oiva_plan* p, const float* parts, int nparts) {
    if (p->fx_on && parts == p->Ppart) {
        // bins sharded over GPUs, the ranks' sums exchanged by the activation kernel itself (no collective, no host in the loop)
        HIP_TRY(launch_activation_xchg(p->stream, parts, nparts, p->fx_gath, p->fx_rank, p->fx_world, p->fx_loopback ? (p->fx_stall ? 2 : 1) : 0, p->fx_nblk_own, p->fx_nblk_peer, p->fx_state + 16,
                                       p->fx_state, (long long)(p->fx_timeout_ms > 0 ? p->fx_timeout_ms : 2000) * 100000, p->R, p->T, p->K,
                                       p->model, p->F_total));
        p->raw_weights = 0;
        return OIVA_OK;
    }
    HIP_TRY(launch_activation(p->stream, parts, nparts, p->R, p->T, p->K, p->model, p->F_total));
    p->raw_weights = 0;
    return OIVA_OK;
}
int stage_cov(oiva_plan* p) {
    HIP_TRY(launch_cov(p->stream, p->X, p->X_pad, p->R, p->Plocal /* weights scratch, (T,16) */, p->wscale, p->model,
                       p->raw_weights, p->Vpart, p->cov_f64(), p->T, p->F, p->M, p->K, p->cov));
    p->wscale_pending = !p->raw_weights;
    return OIVA_OK;
}
int stage_update(oiva_plan* p, bool init_only) {
    UpdateArgs a;
    a.What = p->What;
    a.What64 = p->upd_f64() ? p->What64 : nullptr;
    a.Cx = p->Cx;
    a.Vpart = p->Vpart;
    a.vpart_f64 = p->vpart_f64() ? 1 : 0;
    a.wscale = (!init_only && p->wscale_pending) ? p->wscale : nullptr;
    a.nsplit = p->cov.nsplit;
    a.T = p->T;
    a.F = p->F;
    a.M = p->M;
    a.K = p->K;
    a.init_only = init_only ? 1 : 0;
    a.use_double = p->upd_f64() ? 1 : 0;
    a.layout = (p->prec & OIVA_PREC_UPDATE_ROWS) ? 1 : 0;
    HIP_TRY(launch_update(p->stream, a));
    p->what64_valid = a.What64 != nullptr;      // the float32 variants leave the complex128 copy behind
    if (!init_only) p->wscale_pending = false;
    return OIVA_OK;
}

// covariance and per-bin update as one launch (kernels_cov_update.hip) where the plan's geometry is the kernel's: the headline
// shape and its neighbours (8 channels, 2 sources, four frame splits)
bool cov_update_applies(const oiva_plan* p) {
    return p->fuse_cov_update && !p->cov_f64() && !p->cov.pair32 && !p->raw_weights && !(p->prec & OIVA_PREC_UPDATE_ROWS) && p->K < p->M &&
           cov_update_supported(p->M, p->K, p->T, p->F, p->cov.nsplit, p->cov.tc);
}
int stage_cov_update(oiva_plan* p) {
    UpdateArgs a;
    a.What = p->What;
    a.What64 = p->upd_f64() ? p->What64 : nullptr;
    a.Cx = p->Cx;
    a.Vpart = nullptr;
    a.vpart_f64 = 1;
    a.wscale = nullptr;
    a.nsplit = p->cov.nsplit;
    a.T = p->T, a.F = p->F, a.M = p->M, a.K = p->K;
    a.init_only = 0;
    a.use_double = p->upd_f64() ? 1 : 0;
    a.layout = 0;
    HIP_TRY(launch_cov_update(p->stream, p->X, p->R, p->wscale, p->model, a, p->cov.tc));
    p->what64_valid = a.What64 != nullptr;
    p->wscale_pending = false;
    return OIVA_OK;
}

int one_iteration(oiva_plan* p) {
    int rc;
    if ((rc = stage_power(p))) return rc;
    if ((rc = stage_activation(p, p->Ppart, p->pw.nb))) return rc;
    if (cov_update_applies(p)) return stage_cov_update(p);
    if ((rc = stage_cov(p))) return rc;
    return stage_update(p, false);
}

// ---- X-resident iteration (resident.h) -------------------------------------------------------------
constexpr int kResidentStampIters = 256;

bool resident_applies(const oiva_plan* p) {
    // (the float64 covariance of `precise` exists in the kernel for 4 channels: 32 float64 accumulators per lane)
    const bool arith_ok = !p->cov_f64() || (p->M == 4 && p->upd_f64());
    return p->res_on && p->res_ok && (p->F == p->F_total || p->res_world > 1) && arith_ok && !p->raw_weights && !p->wscale_pending;
}

int resident_alloc(oiva_plan* p) {
    if (p->res_block) return OIVA_OK;
    const ResidentGeom& g = p->rg;
    const size_t Fp = (size_t)g.NB * 16, NA = (size_t)p->M * p->M, K = p->K;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_parts = up((size_t)2 * g.NB * g.NS * g.TW * K * sizeof(float));
    const size_t b_psum = up((size_t)2 * g.NS * g.TW * K * sizeof(float));
    const size_t b_vpart = up(((size_t)g.NS * Fp * K * NA + 2) * sizeof(double));
    const size_t b_rsum = up((size_t)g.NB * ((g.NS * K + 1) & ~(size_t)1) * sizeof(double) + 16);      // rows of an even number of words
    const size_t b_wpub = up(Fp * K * p->M * sizeof(float2));
    const size_t b_flags = up((16 + (size_t)g.NB * g.NS) * sizeof(unsigned));      // ctrl words, then the XCD table
    const size_t b_stamps = up((size_t)kResidentStampIters * kResidentStamps * sizeof(unsigned long long));
    const size_t total = b_parts + b_psum + b_vpart + b_rsum + b_wpub + b_flags + b_stamps;
    if (!p->res_what) HIP_TRY(dev_malloc(&p->res_what, (size_t)p->F * NA * sizeof(float2)));
    if (!p->res_what64) HIP_TRY(dev_malloc(&p->res_what64, (size_t)p->F * NA * sizeof(double2)));
    if (!p->res_code_host) HIP_TRY(hipHostMalloc((void**)&p->res_code_host, sizeof(unsigned), hipHostMallocDefault));
    HIP_TRY(dev_malloc(&p->res_block, total));
    HIP_TRY(hipMemsetAsync(p->res_block, 0, total, p->stream));
    char* c = static_cast<char*>(p->res_block);
    p->res_parts = reinterpret_cast<float*>(c);
    c += b_parts;
    p->res_psum = reinterpret_cast<float*>(c);
    c += b_psum;
    p->res_vpart = reinterpret_cast<double*>(c);
    c += b_vpart;
    p->res_rsum = reinterpret_cast<double*>(c);
    c += b_rsum;
    p->res_wpub = reinterpret_cast<float2*>(c);
    c += b_wpub;
    p->res_flags = reinterpret_cast<unsigned*>(c);
    c += b_flags;
    p->res_stamps = reinterpret_cast<unsigned long long*>(c);
    p->res_block_bytes = total;
    p->res_epoch = 0;
    return OIVA_OK;
}

// n iterations in one persistent launch.  Synchronous.  *ran = false when the launch gave up (nothing changed):
// the caller then runs the four-launch path.
int run_resident(oiva_plan* p, int n, bool* ran) {
    *ran = false;
    int rc = resident_alloc(p);
    if (rc) return rc;
    const ResidentGeom& g = p->rg;
    ResidentArgs a{};
    a.X = p->X;
    a.What = p->What;
    a.What64 = p->upd_f64() ? p->What64 : nullptr;
    a.What_out = p->res_what;
    a.What64_out = p->upd_f64() ? p->res_what64 : nullptr;
    a.what64_valid = p->what64_valid ? 1 : 0;
    a.Cx = p->Cx;
    a.parts = p->res_parts;
    a.psum = p->res_psum;
    a.vpart = p->res_vpart;
    a.rsum = p->res_rsum;
    a.wpub = p->res_wpub;
    a.ctrl = p->res_flags;
    a.xcc_tab = p->res_flags + 16;
    a.stamps = n <= kResidentStampIters ? p->res_stamps : nullptr;
    a.stamp_all = 0;
    if (p->res_trace && n <= 64) {
        const size_t bytes = (size_t)g.NB * g.NS * n * kResidentStamps * sizeof(unsigned long long);
        if (p->res_trace_buf) HIP_TRY(hipFree(p->res_trace_buf));
        p->res_trace_buf = nullptr;
        HIP_TRY(dev_malloc(&p->res_trace_buf, bytes));
        HIP_TRY(hipMemsetAsync(p->res_trace_buf, 0, bytes, p->stream));
        a.stamps = p->res_trace_buf;
        a.stamp_all = 1;
        p->res_trace_iters = n;
    }
    a.T = p->T;
    a.F = p->F;
    a.F_total = p->F_total;
    a.model = p->model;
    a.g = g;
    a.n_iter = n;
    a.epoch0 = p->res_epoch;
    a.timeout_ticks = (long long)(p->res_timeout_ms > 0 ? p->res_timeout_ms : (p->res_world > 1 && !p->res_loopback ? 2000 : 250)) * 100000;   // 100 MHz clock; an iteration takes tens of microseconds (other ranks may start late: 2 s)
    a.stall_block = p->res_stall;
    a.stall_iter = p->res_stall_iter;
    a.rank = p->res_rank;
    a.world = p->res_world;
    a.loopback = p->res_loopback ? 1 : 0;
    for (int r = 0; r < OIVA_XCHG_MAX_RANKS; ++r) a.gath[r] = reinterpret_cast<float*>(p->res_gath[r]);
    HIP_TRY(launch_resident(p->stream, a, p->M, p->K, p->upd_f64(), p->cov_f64()));
    p->res_launches++;
    HIP_TRY(hipMemcpyAsync(p->res_code_host, a.ctrl, sizeof(unsigned), hipMemcpyDeviceToHost, p->stream));   // (pinned: one wait for both)
    HIP_TRY(hipStreamSynchronize(p->stream));
    const unsigned code = *p->res_code_host;
    if (code != 0) {
        // some wait ran into its time-out (workgroups not co-resident, or the test hook): whatever the workgroups that did
        // finish wrote went to the staging copy, W_hat itself is untouched.
        // Clear the flags, start the epochs over and stay on the four-launch path from here on.
        HIP_TRY(hipMemset(p->res_block, 0, p->res_block_bytes));
        p->res_epoch = 0;
        p->res_last_code = (int)code;
        p->res_fallbacks++;
        p->res_on = false;
        p->res_stamped = 0;
        if (p->res_world > 1)      // the other ranks' state is unknown: no silent fall-back, the caller has to decide for all of them
            return fail(OIVA_ERR_STATE, "the X-resident launch gave up waiting (code " + std::to_string(code) +
                                        "): a rank did not deliver its parts in time; W_hat of this rank is unchanged");
        return OIVA_OK;
    }
    // nobody gave up: the staged W_hat becomes the state -- the buffers change places (captured graphs of the four-launch path
    // hold the old addresses: dropped, they are rebuilt if that path is ever used)
    if (!p->graphs.empty() || p->og_graph) {
        int rcg = drop_graph(p);
        if (rcg) return rcg;
    }
    std::swap(p->What, p->res_what);
    if (a.What64_out) std::swap(p->What64, p->res_what64);
    p->og.What = p->What;
    p->og.What64 = p->What64;
    p->res_epoch += (unsigned)n;
    p->res_stamped = (a.stamps && !a.stamp_all) ? n : 0;
    p->what64_valid = a.What64 != nullptr;      // the float32 update leaves the complex128 copy behind
    p->wscale_pending = false;
    *ran = true;
    return OIVA_OK;
}

// The executable graph of `iters` iterations (stream capture records, it does not execute): taken from the plan's cache or
// captured, instantiated and uploaded now.  An iterate(n) call replays ONE graph of n iterations (n <= 32; else 32 at a time):
// a replay costs 10-16 us of which little hides behind the previous one, so the 20 steps of the driver's bench line as
// 8 + 8 + 1 + 1 + 1 + 1 (rounds 1-4) ran 4.7 us per iteration above the steady state of the same kernels (205.1 against 200.4
// us).  oiva_plan_use_graph captures the 8-iteration graph ahead of time, so that short calls never capture inside a
// caller's timed region; other lengths are captured by the first call that asks for them (0.1-0.4 ms) and kept.
int graph_for(oiva_plan* p, int iters, hipGraphExec_t* out) {
    for (size_t i = 0; i < p->graphs.size(); ++i)
        if (p->graphs[i].first == iters) {
            auto hit = p->graphs[i];
            p->graphs.erase(p->graphs.begin() + (long)i);
            p->graphs.push_back(hit);
            *out = hit.second;
            return OIVA_OK;
        }
    const bool pending = p->wscale_pending;
    const int raw = p->raw_weights;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    HIP_TRY(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
    int r = OIVA_OK;
    for (int i = 0; i < iters && r == OIVA_OK; ++i) r = one_iteration(p);
    hipError_t e = hipStreamEndCapture(p->stream, &graph);
    p->wscale_pending = pending;   // capturing toggled the host-side flags without running anything
    p->raw_weights = raw;
    if (r) {
        if (graph) (void)hipGraphDestroy(graph);
        return r;
    }
    HIP_TRY(e);
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    HIP_TRY(e);
    // move the executable graph to the device now: otherwise its FIRST launch pays for that, inside whatever the
    // caller is timing (a few percent of a 20-iteration run)
    HIP_TRY(hipGraphUpload(exec, p->stream));
    if ((int)p->graphs.size() >= kGraphCache) {
        HIP_TRY(hipStreamSynchronize(p->stream));          // (the evicted graph may still be replaying)
        HIP_TRY(hipGraphExecDestroy(p->graphs.front().second));
        p->graphs.erase(p->graphs.begin());
    }
    p->graphs.emplace_back(iters, exec);
    *out = exec;
    return OIVA_OK;
}

int build_graphs(oiva_plan* p) {
    hipGraphExec_t g = nullptr;
    return graph_for(p, kGraphBatch, &g);
}

// W_hat lives twice on the device: complex64 for the streaming kernels, complex128 for the float64 update
int upload_what(oiva_plan* p, const std::vector<double2>& wh) {
    std::vector<float2> w32(wh.size());
    for (size_t i = 0; i < wh.size(); ++i) w32[i] = make_float2((float)wh[i].x, (float)wh[i].y);
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemcpy(p->What, w32.data(), w32.size() * sizeof(float2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->What64, wh.data(), wh.size() * sizeof(double2), hipMemcpyHostToDevice));
    p->what64_valid = true;
    return OIVA_OK;
}

// the current W_hat in float64: the complex128 copy when the float64 update maintains it, else the complex64 one
int download_what(oiva_plan* p, std::vector<double2>& wh) {
    wh.resize((size_t)p->F * p->M * p->M);
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->what64_valid) {
        HIP_TRY(hipMemcpy(wh.data(), p->What64, wh.size() * sizeof(double2), hipMemcpyDeviceToHost));
        return OIVA_OK;
    }
    std::vector<float2> w32(wh.size());
    HIP_TRY(hipMemcpy(w32.data(), p->What, w32.size() * sizeof(float2), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < wh.size(); ++i) wh[i] = make_double2(w32[i].x, w32[i].y);
    return OIVA_OK;
}

// the padded copy of X the vector-ALU covariance kernels read at 9 / 11 / 13 / 15 channels
int ensure_pad(oiva_plan* p) {
    if (p->X_pad == nullptr || p->pad_valid) return OIVA_OK;
    HIP_TRY(launch_pad_channels(p->stream, p->X, p->X_pad, (long long)p->T * p->F, p->M));
    p->pad_valid = true;
    return OIVA_OK;
}

int check_ready(oiva_plan* p) {
    NEED(p != nullptr, OIVA_ERR_ARG, "null plan");
    NEED(p->have_x, OIVA_ERR_STATE, "X not set (oiva_plan_set_x_host/_dev)");
    NEED(p->have_cx, OIVA_ERR_STATE, "input covariance not computed (oiva_plan_covariance)");
    NEED(p->have_w, OIVA_ERR_STATE, "demixing matrix not set (oiva_plan_set_w)");
    return OIVA_OK;
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        (void)hipGetDevice(&prev);
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() {
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (prev >= 0 && cur != prev) (void)hipSetDevice(prev);
    }
};

}  // namespace

extern "C" {

int oiva_version(void) { return 100; }

const char* oiva_last_error(void) { return g_err.c_str(); }

int oiva_device_count(int* n) {
    NEED(n != nullptr, OIVA_ERR_ARG, "null pointer");
    HIP_TRY(hipGetDeviceCount(n));
    return OIVA_OK;
}

int oiva_plan_create(oiva_plan** out, int device, int T, int F, int M, int K, int model, int F_total, void* stream) {
    NEED(out != nullptr, OIVA_ERR_ARG, "null out pointer");
    *out = nullptr;
    NEED(T >= 1 && F >= 1, OIVA_ERR_ARG, "T and F must be >= 1");
    NEED(M >= 1 && M <= OIVA_MAX_CHANNELS, OIVA_ERR_ARG, "number of channels must be in 1..16");
    NEED(K >= 1 && K <= M, OIVA_ERR_ARG, "n_src must be in 1..n_chan");
    NEED(model == OIVA_MODEL_LAPLACE || model == OIVA_MODEL_GAUSS, OIVA_ERR_ARG, "unknown model");
    NEED(F_total >= F, OIVA_ERR_ARG, "F_total must be >= F");
    NEED(cov_supported(M), OIVA_ERR_ARG, "unsupported number of channels");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    NEED(device >= 0 && device < ndev, OIVA_ERR_ARG, "no such device");
    DeviceGuard guard(device);

    oiva_plan* p = new oiva_plan();
    p->device = device;
    p->T = T;
    p->F = F;
    p->M = M;
    p->K = K;
    p->model = model;
    p->F_total = F_total;
    if (stream) {
        p->stream = (hipStream_t)stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete p;
            return fail(OIVA_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
        }
        p->own_stream = true;
    }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            p->n_cu = prop.multiProcessorCount;
    }
    if (M > 8 && M % 2 == 1) {
        // 9 / 11 / 13 / 15 channels: the vector-ALU covariance kernels read 16-byte pieces at an even channel pitch, so
        // they get their own copy of X with one zero channel per bin (filled by oiva_plan_covariance; + (M + 1) / M of X)
        hipError_t ep = dev_malloc((void**)&p->X_pad, (size_t)T * F * (M + 1) * sizeof(float2));
        if (ep != hipSuccess) {
            oiva_plan_destroy(p);
            return fail(OIVA_ERR_HIP, std::string("allocation of the padded copy of X failed: ") + hipGetErrorString(ep));
        }
    }
    {
        const char* v = std::getenv("OIVA_COV_HMFMA");
        p->cov_hmfma_on = !(v && v[0] == '0');
        const char* u = std::getenv("OIVA_COV_UPDATE");
        p->fuse_cov_update = u && u[0] == '1';
    }
    choose_cov_geom(p, 0);
    choose_pow_geom(p, 0);
    choose_stats_geom(p);
    p->res_ok = resident_geometry(T, F, M, K, p->n_cu, 0, &p->rg);
    const size_t nTK = (size_t)T * K;
    const size_t nFMM = (size_t)F * M * M;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** ptr, size_t bytes) {
        if (e == hipSuccess) e = dev_malloc(ptr, bytes);
    };
    alloc((void**)&p->What, nFMM * sizeof(float2));
    alloc((void**)&p->What64, nFMM * sizeof(double2));
    alloc((void**)&p->Cx, nFMM * sizeof(double));
    alloc((void**)&p->Ppart, (size_t)p->pw.nb * nTK * sizeof(float));
    p->ppart_alloc = p->pw.nb;
    alloc((void**)&p->Plocal, std::max(nTK, ((size_t)T + 1) * 32) * sizeof(float));   // also the (T + 1, 16) weights scratch (floats or doubles)
    alloc((void**)&p->R, r_buffer_bytes(T, K));   // activations, zeroed pad rows, per-block sums (rsum_offset_floats)
    if (e == hipSuccess) e = hipMemset(p->R, 0, r_buffer_bytes(T, K));
    alloc((void**)&p->wscale, (size_t)K * sizeof(float));
    alloc((void**)&p->Spart, (size_t)p->stg.nsplit * F * K * 3 * sizeof(float));
    alloc((void**)&p->scratch_c, (size_t)K * nFMM * sizeof(double2));
    alloc((void**)&p->scratch_p, std::max((size_t)K * nFMM, nTK) * sizeof(double));
    for (auto& ev : p->ev) {
        if (e == hipSuccess) e = hipEventCreate(&ev);
    }
    if (e != hipSuccess) {
        oiva_plan_destroy(p);
        return fail(OIVA_ERR_HIP, std::string("allocation failed: ") + hipGetErrorString(e));
    }
    int rc = ensure_vpart(p);
    if (rc) {
        oiva_plan_destroy(p);
        return rc;
    }
    *out = p;
    return OIVA_OK;
}

int oiva_plan_destroy(oiva_plan* p) {
    if (!p) return OIVA_OK;
    DeviceGuard guard(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    for (auto& g : p->graphs) (void)hipGraphExecDestroy(g.second);
    if (p->og_graph) (void)hipGraphExecDestroy(p->og_graph);
    if (p->res_code_host) (void)hipHostFree(p->res_code_host);
    if (p->io_stream) (void)hipStreamSynchronize(p->io_stream);
    big_free(p->device, p->X_owned, p->x_owned_bytes);
    big_free(p->device, p->Y, p->y_bytes);
    for (int i = 0; i < kHostRingSlots; ++i) {
        big_free(p->device, p->io_c128[i], p->io_c128_bytes);
        p->io_c128[i] = nullptr;
    }
    void* bufs[] = {p->X_pad, p->What, p->What64, p->Cx,        p->Vpart,    p->Ppart, p->Plocal, p->res_block, p->res_trace_buf, p->res_what, p->res_what64,
                    p->R,       p->wscale, p->Spart, p->scratch_c, p->scratch_p};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    for (void* b : p->og_bufs)
        if (b) (void)hipFree(b);
    if (p->res_loop_buf) (void)hipFree(p->res_loop_buf);
    if (p->fx_loop_buf) (void)hipFree(p->fx_loop_buf);
    if (p->fx_state) (void)hipFree(p->fx_state);
    if (p->fx_flag_host) (void)hipHostFree(p->fx_flag_host);
    if (p->io_stream) (void)hipStreamDestroy(p->io_stream);
    for (int i = 0; i < kHostRingSlots; ++i) {
        if (p->io_written[i]) (void)hipEventDestroy(p->io_written[i]);
        if (p->io_copied[i]) (void)hipEventDestroy(p->io_copied[i]);
    }
    if (p->ck_what) (void)hipFree(p->ck_what);
    if (p->ck_what64) (void)hipFree(p->ck_what64);
    for (auto& ev : p->ev)
        if (ev) (void)hipEventDestroy(ev);
    if (p->own_stream && p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return OIVA_OK;
}

int oiva_plan_set_x_host(oiva_plan* p, const void* X, long long row_pitch_bytes) {
    NEED(p && X, OIVA_ERR_ARG, "null argument");
    DeviceGuard guard(p->device);
    const size_t row = (size_t)p->F * p->M * sizeof(float2);
    const size_t pitch = row_pitch_bytes > 0 ? (size_t)row_pitch_bytes : row;
    NEED(pitch >= row, OIVA_ERR_ARG, "row pitch smaller than one frame of this plan's bins");
    if (!p->X_owned) {
        HIP_TRY(big_alloc(p->device, (void**)&p->X_owned, row * p->T));
        p->x_owned_bytes = row * p->T;
    }
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemcpy2D(p->X_owned, row, X, pitch, row, p->T, hipMemcpyHostToDevice));
    if (p->X != p->X_owned) {             // switching from a borrowed array: captured graphs hold its pointer
        int rc = drop_graph(p);
        if (rc) return rc;
    }
    p->X = p->X_owned;
    p->have_x = true;
    p->have_cx = false;
    p->pad_valid = false;
    return OIVA_OK;
}

int oiva_plan_set_x_host_c128(oiva_plan* p, const void* X, long long row_pitch_bytes) {
    NEED(p && X, OIVA_ERR_ARG, "null argument");
    DeviceGuard guard(p->device);
    const size_t n_row = (size_t)p->F * p->M;
    const size_t row = n_row * sizeof(double2);
    const size_t pitch = row_pitch_bytes > 0 ? (size_t)row_pitch_bytes : row;
    NEED(pitch >= row, OIVA_ERR_ARG, "row pitch smaller than one frame of this plan's bins");
    if (!p->X_owned) {
        HIP_TRY(big_alloc(p->device, (void**)&p->X_owned, n_row * sizeof(float2) * p->T));
        p->x_owned_bytes = n_row * sizeof(float2) * p->T;
    }
    HIP_TRY(hipStreamSynchronize(p->stream));
    // in slabs of frames through a staging buffer (<= 256 MB): the conversion of one slab overlaps nothing, but the
    // footprint stays bounded and the host never touches the data (a NumPy astype of 1 GB costs 60 ms)
    const int slab = (int)std::max<size_t>(1, std::min<size_t>((size_t)p->T, ((size_t)256 << 20) / row));
    double2* stage = nullptr;
    HIP_TRY(big_alloc(p->device, (void**)&stage, row * slab));
    hipError_t e = hipSuccess;
    for (int t0 = 0; t0 < p->T && e == hipSuccess; t0 += slab) {
        const int nt = std::min(slab, p->T - t0);
        e = hipMemcpy2D(stage, row, static_cast<const char*>(X) + (size_t)t0 * pitch, pitch, row, nt, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = launch_cast_c128_to_c64(p->stream, stage, p->X_owned + (size_t)t0 * n_row, (long long)nt * n_row);
        if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    }
    if (e == hipSuccess) big_free(p->device, stage, row * slab); else (void)hipFree(stage);
    HIP_TRY(e);
    if (p->X != p->X_owned) {             // switching from a borrowed array: captured graphs hold its pointer
        int rc = drop_graph(p);
        if (rc) return rc;
    }
    p->X = p->X_owned;
    p->have_x = true;
    p->have_cx = false;
    p->pad_valid = false;
    return OIVA_OK;
}

int oiva_plan_set_x_dev(oiva_plan* p, const void* X_dev) {
    NEED(p && X_dev, OIVA_ERR_ARG, "null argument");
    NEED(((uintptr_t)X_dev & 15) == 0, OIVA_ERR_ARG, "device X must be 16-byte aligned");
    if (p->X != (const float2*)X_dev) {   // captured graphs hold the old pointer
        DeviceGuard guard(p->device);
        HIP_TRY(hipStreamSynchronize(p->stream));
        int rc = drop_graph(p);
        if (rc) return rc;
    }
    p->X = (const float2*)X_dev;
    p->have_x = true;
    p->have_cx = false;
    p->pad_valid = false;
    return OIVA_OK;
}

int oiva_plan_covariance(oiva_plan* p) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    NEED(p->have_x, OIVA_ERR_STATE, "X not set");
    DeviceGuard guard(p->device);
    // (the padded copy follows X here: every path that changes X clears have_cx, and the iteration needs Cx; a borrowed X
    //  rewritten in place needs a new Cx too)
    p->pad_valid = false;
    int rcp = ensure_pad(p);
    if (rcp) return rcp;
    CovGeom g = p->cov;
    g.kc = 1;
    g.part32 = 0;          // (one "source": never the matrix-core kernel)
    // unit weights, one "source": partials land in Vpart laid out as [nsplit][F][1][M*M]
    HIP_TRY(launch_cov(p->stream, p->X, p->X_pad, nullptr, nullptr, nullptr, p->model, 0, p->Vpart, p->cov_f64(), p->T, p->F, p->M, 1, g));
    HIP_TRY(launch_sum_parts(p->stream, p->Vpart, p->vpart_f64_of(g), g.nsplit, p->Cx, (long long)p->F * p->M * p->M, 1. / (double)p->T));
    p->have_cx = true;
    return OIVA_OK;
}

int oiva_plan_get_cx(oiva_plan* p, void* Cx_host, int f64) {
    NEED(p && Cx_host, OIVA_ERR_ARG, "null argument");
    NEED(p->have_cx, OIVA_ERR_STATE, "input covariance not computed");
    DeviceGuard guard(p->device);
    const size_t n = (size_t)p->F * p->M * p->M;
    HIP_TRY(launch_unpack_herm(p->stream, p->Cx, p->scratch_c, f64 != 0, p->F, p->M));
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemcpy(Cx_host, p->scratch_c, n * (f64 ? sizeof(double2) : sizeof(float2)), hipMemcpyDeviceToHost));
    return OIVA_OK;
}

int oiva_plan_set_w(oiva_plan* p, const void* W0_host, int f64) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    NEED(p->have_cx, OIVA_ERR_STATE, "input covariance not computed (needed for the orthogonality constraint)");
    DeviceGuard guard(p->device);
    const int F = p->F, M = p->M, K = p->K;
    std::vector<double2> wh((size_t)F * M * M, make_double2(0., 0.));
    for (int f = 0; f < F; ++f) {
        double2* m = wh.data() + (size_t)f * M * M;
        for (int r = 0; r < M; ++r)
            for (int k = 0; k < K; ++k) {
                const size_t i = ((size_t)f * M + r) * K + k;
                if (!W0_host) {
                    m[r * M + k] = make_double2(r == k ? 1. : 0., 0.);           // overiva.py:113-114
                } else if (f64) {
                    m[r * M + k] = static_cast<const double2*>(W0_host)[i];       // overiva.py:116-117
                } else {
                    const float2 v = static_cast<const float2*>(W0_host)[i];
                    m[r * M + k] = make_double2(v.x, v.y);
                }
            }
        for (int r = K; r < M; ++r) m[r * M + r] = make_double2(-1., 0.);  // overiva.py:122-123
    }
    int rc = upload_what(p, wh);
    if (rc) return rc;
    p->wscale_pending = false;
    p->have_w = true;
    if (K < M) return stage_update(p, true);  // J from the orthogonality constraint, overiva.py:120-121
    return OIVA_OK;
}

static int set_w_from_eigenvectors(oiva_plan* p, double* evals_host, bool lapack_phase);

int oiva_plan_set_w_pca(oiva_plan* p, double* evals_host) { return set_w_from_eigenvectors(p, evals_host, false); }
int oiva_plan_set_w_eig(oiva_plan* p) { return set_w_from_eigenvectors(p, nullptr, true); }

static int set_w_from_eigenvectors(oiva_plan* p, double* evals_host, bool lapack_phase) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    NEED(p->have_cx, OIVA_ERR_STATE, "input covariance not computed (oiva_plan_covariance)");
    DeviceGuard guard(p->device);
    double* evals = evals_host ? p->scratch_p : nullptr;       // scratch_p holds at least K * F * M * M doubles
    HIP_TRY(launch_pca_subspace(p->stream, p->Cx, p->What, p->What64, evals, p->F, p->M, p->K, lapack_phase));
    p->what64_valid = true;
    p->wscale_pending = false;
    p->have_w = true;
    if (evals_host) {
        HIP_TRY(hipStreamSynchronize(p->stream));
        HIP_TRY(hipMemcpy(evals_host, evals, (size_t)p->F * p->M * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (p->K < p->M) return stage_update(p, true);             // J from the orthogonality constraint
    return OIVA_OK;
}

static int check_fused(oiva_plan* p);

int oiva_plan_demix_dev(oiva_plan* p, int proj_back, void** Y_dev) {
    int rc = check_ready(p);
    if (rc) return rc;
    NEED(Y_dev, OIVA_ERR_ARG, "null output");
    DeviceGuard guard(p->device);
    if ((rc = check_fused(p))) return rc;
    const size_t row = (size_t)p->F * p->K * sizeof(float2);
    if (!p->Y) {
        HIP_TRY(big_alloc(p->device, (void**)&p->Y, row * p->T));
        p->y_bytes = row * p->T;
    }
    const float* sp = nullptr;
    if (proj_back) {
        HIP_TRY(launch_demix_stats(p->stream, p->X, p->What, p->Spart, p->T, p->F, p->M, p->K, p->stg));
        sp = p->Spart;
    }
    HIP_TRY(launch_demix_write(p->stream, p->X, p->What, sp, p->stg.nsplit, p->Y, p->T, p->F, p->M, p->K));
    HIP_TRY(hipStreamSynchronize(p->stream));
    *Y_dev = p->Y;
    return OIVA_OK;
}

int oiva_plan_iterate(oiva_plan* p, int n) {
    int rc = check_ready(p);
    if (rc) return rc;
    NEED(n >= 0, OIVA_ERR_ARG, "negative iteration count");
    NEED(p->F == p->F_total || resident_applies(p) || p->fx_on, OIVA_ERR_STATE,
         "plan owns a bin shard: drive it with oiva_plan_power / all-gather / oiva_plan_update (or connect an in-kernel exchange: "
         "oiva_plan_fused_connect, oiva_plan_resident_connect)");
    DeviceGuard guard(p->device);
    if (n == 0) return OIVA_OK;
    if (resident_applies(p)) {
        bool ran = false;
        if ((rc = run_resident(p, n, &ran))) return rc;
        if (ran) return OIVA_OK;
    }
    if (p->use_graph) {
        for (int left = n; left > 0;) {
            const int m = std::min(left, kGraphMaxIters);
            hipGraphExec_t g = nullptr;
            if ((rc = graph_for(p, m, &g))) return rc;
            HIP_TRY(hipGraphLaunch(g, p->stream));
            left -= m;
        }
        // the host-side flags after n iterations, whether the graph was captured now (capturing runs stage_update, which sets
        // them) or taken from the cache (nothing runs on the host): the float32 update leaves the complex128 copy behind
        p->wscale_pending = false;
        p->raw_weights = 0;
        p->what64_valid = p->upd_f64();
        return OIVA_OK;
    }
    for (int i = 0; i < n; ++i)
        if ((rc = one_iteration(p))) return rc;
    return OIVA_OK;
}

int oiva_plan_power(oiva_plan* p) {
    int rc = check_ready(p);
    if (rc) return rc;
    DeviceGuard guard(p->device);
    return stage_power(p);
}

int oiva_plan_power_buffer(oiva_plan* p, int parts_per_rank, void** parts_dev, long long* bytes) {
    NEED(p && parts_dev, OIVA_ERR_ARG, "null argument");
    NEED(parts_per_rank >= p->pw.nb, OIVA_ERR_ARG, "parts_per_rank smaller than this plan's own bin batches");
    DeviceGuard guard(p->device);
    const size_t part = (size_t)p->T * p->K * sizeof(float);
    if (parts_per_rank > p->ppart_alloc) {
        HIP_TRY(hipStreamSynchronize(p->stream));
        int rc = drop_graph(p);
        if (rc) return rc;
        if (p->Ppart) HIP_TRY(hipFree(p->Ppart));
        p->Ppart = nullptr;
        HIP_TRY(dev_malloc(&p->Ppart, part * parts_per_rank));
        // parts beyond nb stay zero; on the plan's own stream, so that it is ordered before the next power pass
        HIP_TRY(hipMemsetAsync(p->Ppart, 0, part * parts_per_rank, p->stream));
        p->ppart_alloc = parts_per_rank;
    }
    *parts_dev = p->Ppart;
    if (bytes) *bytes = (long long)(part * parts_per_rank);
    return OIVA_OK;
}

int oiva_plan_update(oiva_plan* p, const void* parts_dev, int nparts) {
    int rc = check_ready(p);
    if (rc) return rc;
    NEED(parts_dev && nparts >= 1, OIVA_ERR_ARG, "need at least one part");
    DeviceGuard guard(p->device);
    if ((rc = stage_activation(p, (const float*)parts_dev, nparts))) return rc;
    if (cov_update_applies(p)) return stage_cov_update(p);
    if ((rc = stage_cov(p))) return rc;
    return stage_update(p, false);
}

// Y = demix(X, W) (+ projection back) into a host array, complex64 or complex128 (overiva.py:192-204).
//   $OIVA_DEMIX_IO = ring (default): slabs of frames; slab k is computed (write_kernel; for complex128 also widened on the
//       device), copied into a slot of the process-wide pinned ring on a second stream and moved from there into the
//       caller's array by the copy threads, all three overlapping across slabs;
//   register: the caller's array is page-locked for the call (hipHostRegister) and the slabs are copied straight into it;
//   legacy: one kernel over all frames, one synchronous hipMemcpy2D (the form up to round 4).
// Same bits in every mode: the kernel does the same arithmetic per (frame, bin) whatever the slab.
static int demix_to_host(oiva_plan* p, void* Y_host, long long row_pitch_bytes, int proj_back, bool c128) {
    int rc = check_ready(p);
    if (rc) return rc;
    NEED(Y_host, OIVA_ERR_ARG, "null output");
    DeviceGuard guard(p->device);
    const size_t n_row = (size_t)p->F * p->K;
    const size_t row_dev = n_row * sizeof(float2);
    const size_t row = c128 ? n_row * sizeof(double2) : row_dev;
    const size_t pitch = row_pitch_bytes > 0 ? (size_t)row_pitch_bytes : row;
    NEED(pitch >= row, OIVA_ERR_ARG, "row pitch smaller than one frame of this plan's bins");
    if ((rc = check_fused(p))) return rc;
    if (!p->Y) {
        HIP_TRY(big_alloc(p->device, (void**)&p->Y, row_dev * p->T));
        p->y_bytes = row_dev * p->T;
    }
    const float* sp = nullptr;
    if (proj_back) {
        HIP_TRY(launch_demix_stats(p->stream, p->X, p->What, p->Spart, p->T, p->F, p->M, p->K, p->stg));
        sp = p->Spart;
    }
    static const int mode = [] {
        const char* v = std::getenv("OIVA_DEMIX_IO");
        if (v && !std::strcmp(v, "legacy")) return 0;
        if (v && !std::strcmp(v, "register")) return 2;
        return 1;
    }();
    const size_t slab_target = p->io_slab_bytes ? p->io_slab_bytes : ((size_t)8 << 20);
    if (mode == 0 || (size_t)p->T * row < slab_target / 2) {
        // small outputs (and the legacy form): one kernel, one copy
        HIP_TRY(launch_demix_write(p->stream, p->X, p->What, sp, p->stg.nsplit, p->Y, p->T, p->F, p->M, p->K));
        if (!c128) {
            HIP_TRY(hipStreamSynchronize(p->stream));
            HIP_TRY(hipMemcpy2D(Y_host, pitch, p->Y, row, row, p->T, hipMemcpyDeviceToHost));
            return OIVA_OK;
        }
        const int slab = (int)std::max<size_t>(1, std::min<size_t>((size_t)p->T, ((size_t)256 << 20) / row));
        double2* stage = nullptr;
        HIP_TRY(dev_malloc(&stage, row * slab));
        hipError_t e = hipSuccess;
        for (int t0 = 0; t0 < p->T && e == hipSuccess; t0 += slab) {
            const int nt = std::min(slab, p->T - t0);
            e = launch_cast_c64_to_c128(p->stream, p->Y + (size_t)t0 * n_row, stage, (long long)nt * n_row);
            if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
            if (e == hipSuccess)
                e = hipMemcpy2D(static_cast<char*>(Y_host) + (size_t)t0 * pitch, pitch, stage, row, row, nt, hipMemcpyDeviceToHost);
        }
        (void)hipFree(stage);
        HIP_TRY(e);
        return OIVA_OK;
    }
    // ---- slabs of about 8 MB of output
    const int slab = (int)std::max<size_t>(1, std::min<size_t>((size_t)p->T, slab_target / row));
    const int nslab = ceil_div(p->T, slab);
    const size_t slab_bytes = (size_t)slab * row;
    if (!p->io_stream) HIP_TRY(hipStreamCreateWithFlags(&p->io_stream, hipStreamNonBlocking));
    for (int i = 0; i < kHostRingSlots; ++i) {
        if (!p->io_written[i]) HIP_TRY(hipEventCreateWithFlags(&p->io_written[i], hipEventDisableTiming));
        if (!p->io_copied[i]) HIP_TRY(hipEventCreateWithFlags(&p->io_copied[i], hipEventDisableTiming));
    }
    if (c128 && p->io_c128_bytes < slab_bytes) {
        HIP_TRY(hipStreamSynchronize(p->stream));
        // all three or none: the recorded size is that of every non-null slot, also after a failed allocation
        for (auto& b : p->io_c128) {
            big_free(p->device, b, p->io_c128_bytes);
            b = nullptr;
        }
        p->io_c128_bytes = 0;
        for (auto& b : p->io_c128) {
            hipError_t e = big_alloc(p->device, (void**)&b, slab_bytes);
            if (e != hipSuccess) {
                for (auto& c : p->io_c128) {
                    if (c) (void)hipFree(c);
                    c = nullptr;
                }
                HIP_TRY(e);
            }
        }
        p->io_c128_bytes = slab_bytes;
    }
    bool registered = false;
    void* pinned[kHostRingSlots] = {};
    if (mode == 2) {
        registered = hipHostRegister(Y_host, pitch * (size_t)(p->T - 1) + row, hipHostRegisterDefault) == hipSuccess;
        if (!registered) (void)hipGetLastError();      // (an unaligned or foreign range: the ring serves)
    }
    // (the pinned ring and the copy threads are process-wide: one hand-over at a time, whatever thread or plan asks)
    static std::mutex io_mutex;
    std::unique_lock<std::mutex> io_lock(io_mutex, std::defer_lock);
    if (!registered) {
        io_lock.lock();
        HIP_TRY(host_ring_slots(slab_bytes, pinned));
    }
    auto finish = [&](hipError_t e) -> int {
        // nothing of this call is in flight when the ring (io_mutex) passes to the next caller, also after an error
        (void)hipStreamSynchronize(p->io_stream);
        if (registered) (void)hipHostUnregister(Y_host);
        if (e != hipSuccess) return fail(OIVA_ERR_HIP, std::string("final demix: ") + hipGetErrorString(e));
        return OIVA_OK;
    };
    auto issue = [&](int k) -> hipError_t {
        const int s = k % kHostRingSlots;
        const int t0 = k * slab, nt = std::min(slab, p->T - t0);
        float2* ydev = p->Y + (size_t)t0 * n_row;
        hipError_t e = launch_demix_write(p->stream, p->X + (size_t)t0 * p->F * p->M, p->What, sp, p->stg.nsplit, ydev, nt, p->F, p->M, p->K);
        const void* src = ydev;
        if (e == hipSuccess && c128) {
            e = launch_cast_c64_to_c128(p->stream, ydev, p->io_c128[s], (long long)nt * n_row);
            src = p->io_c128[s];
        }
        if (e == hipSuccess) e = hipEventRecord(p->io_written[s], p->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(p->io_stream, p->io_written[s], 0);
        if (e == hipSuccess) {
            if (registered)
                e = hipMemcpy2DAsync(static_cast<char*>(Y_host) + (size_t)t0 * pitch, pitch, src, row, row, nt, hipMemcpyDeviceToHost, p->io_stream);
            else
                e = hipMemcpyAsync(pinned[s], src, (size_t)nt * row, hipMemcpyDeviceToHost, p->io_stream);
        }
        if (e == hipSuccess) e = hipEventRecord(p->io_copied[s], p->io_stream);
        // (slot s -- the pinned buffer and, for complex128, the device staging buffer -- is rewritten by slab k + slots, which
        //  the loop below issues only after it has waited for THIS copy and moved its data on)
        return e;
    };
    int issued = 0;
    for (int k = 0; k < nslab; ++k) {
        hipError_t e = hipSuccess;
        while (issued < nslab && issued - k < kHostRingSlots && e == hipSuccess) e = issue(issued++);
        if (k == 0 && !registered && e == hipSuccess) {
            // the copy threads must not meet pages that were never touched (a fresh NumPy array): eight threads faulting the
            // same mapping in ran at 8 GB/s where resident pages take 41.  Populating them costs 1-2 ms for 131 MB -- next to
            // nothing when the caller already did (overiva() does, behind its iterations) -- and runs while the first slabs
            // are computed and cross PCIe.
            host_prefault(Y_host, pitch * (size_t)(p->T - 1) + row, /*may_touch*/ pitch == row);
        }
        if (e == hipSuccess) e = hipEventSynchronize(p->io_copied[k % kHostRingSlots]);
        if (e != hipSuccess) return finish(e);
        if (!registered) {
            const int t0 = k * slab, nt = std::min(slab, p->T - t0);
            host_copy_rows(static_cast<char*>(Y_host) + (size_t)t0 * pitch, pitch, pinned[k % kHostRingSlots], row, row, nt);
        }
    }
    return finish(hipStreamSynchronize(p->stream));
}

int oiva_pool_trim(void) {
    pool_release_all();
    return OIVA_OK;
}

int oiva_plan_set_io_slab(oiva_plan* p, long long bytes) {
    NEED(p && bytes >= 0, OIVA_ERR_ARG, "bad arguments");
    p->io_slab_bytes = (size_t)bytes;
    return OIVA_OK;
}

int oiva_plan_demix(oiva_plan* p, void* Y_host, long long row_pitch_bytes, int proj_back) {
    return demix_to_host(p, Y_host, row_pitch_bytes, proj_back, false);
}

int oiva_plan_demix_c128(oiva_plan* p, void* Y_host, long long row_pitch_bytes, int proj_back) {
    return demix_to_host(p, Y_host, row_pitch_bytes, proj_back, true);
}

int oiva_plan_get_w(oiva_plan* p, void* W_host, int f64) {
    NEED(p && W_host, OIVA_ERR_ARG, "null argument");
    NEED(p->have_w, OIVA_ERR_STATE, "demixing matrix not set");
    DeviceGuard guard(p->device);
    const int F = p->F, M = p->M, K = p->K;
    int rc = check_fused(p);
    if (rc) return rc;
    std::vector<double2> wh;
    rc = download_what(p, wh);
    if (rc) return rc;
    bool finite = true;
    for (int f = 0; f < F; ++f)
        for (int r = 0; r < M; ++r)
            for (int k = 0; k < K; ++k) {
                const double2 v = wh[((size_t)f * M + r) * M + k];
                finite = finite && std::isfinite(v.x) && std::isfinite(v.y);
                const size_t i = ((size_t)f * M + r) * K + k;
                if (f64)
                    static_cast<double2*>(W_host)[i] = v;
                else
                    static_cast<float2*>(W_host)[i] = make_float2((float)v.x, (float)v.y);
            }
    if (!finite) return fail(OIVA_ERR_NUMERIC, "demixing matrix holds non-finite values (singular W_hat^H V)");
    return OIVA_OK;
}

// the in-kernel exchange of the four-launch path gave up waiting for a rank: recorded on the device, looked at by everything
// that hands results to the caller (sync, get_w, the demix entry points).  The flag is copied into a pinned word behind the
// work already queued on the plan's stream and read after ONE wait for that stream.
static int check_fused(oiva_plan* p) {
    if (!p->fx_state || !p->fx_on) return OIVA_OK;
    if (!p->fx_flag_host) HIP_TRY(hipHostMalloc((void**)&p->fx_flag_host, sizeof(unsigned), hipHostMallocDefault));
    HIP_TRY(hipMemcpyAsync(p->fx_flag_host, p->fx_state, sizeof(unsigned), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    const unsigned code = *p->fx_flag_host;
    if (code != 0)
        return fail(OIVA_ERR_STATE, "the exchange inside the activation kernel gave up waiting for a rank's partial powers (workgroup " +
                                        std::to_string(code - 1) + "): the state of this plan is undefined (oiva_plan_restore_w brings back the "
                                        "demixing matrices saved by oiva_plan_save_w; oiva_plan_fused_connect(p, NULL) clears the condition)");
    return OIVA_OK;
}

int oiva_plan_sync(oiva_plan* p) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    DeviceGuard guard(p->device);
    HIP_TRY(hipStreamSynchronize(p->stream));
    return check_fused(p);
}

int oiva_plan_save_w(oiva_plan* p) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    NEED(p->have_w, OIVA_ERR_STATE, "demixing matrix not set");
    DeviceGuard guard(p->device);
    const size_t n = (size_t)p->F * p->M * p->M;
    if (!p->ck_what) HIP_TRY(dev_malloc((void**)&p->ck_what, n * sizeof(float2)));
    if (!p->ck_what64) HIP_TRY(dev_malloc((void**)&p->ck_what64, n * sizeof(double2)));
    // on the plan's stream: ordered behind the iterations already queued, in front of the ones that follow
    HIP_TRY(hipMemcpyAsync(p->ck_what, p->What, n * sizeof(float2), hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->ck_what64, p->What64, n * sizeof(double2), hipMemcpyDeviceToDevice, p->stream));
    p->ck_what64_valid = p->what64_valid;
    p->ck_valid = true;
    return OIVA_OK;
}

int oiva_plan_restore_w(oiva_plan* p) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    NEED(p->ck_valid, OIVA_ERR_STATE, "nothing saved (oiva_plan_save_w)");
    DeviceGuard guard(p->device);
    const size_t n = (size_t)p->F * p->M * p->M;
    HIP_TRY(hipMemcpyAsync(p->What, p->ck_what, n * sizeof(float2), hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->What64, p->ck_what64, n * sizeof(double2), hipMemcpyDeviceToDevice, p->stream));
    p->what64_valid = p->ck_what64_valid;
    p->wscale_pending = false;
    p->have_w = true;
    return OIVA_OK;
}

static int fused_setup(oiva_plan* p) {
    const size_t words = 16 + (size_t)rsum_blocks(p->T) * p->K;
    if (!p->fx_state) HIP_TRY(dev_malloc((void**)&p->fx_state, words * sizeof(unsigned)));
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemset(p->fx_state, 0, words * sizeof(unsigned)));
    return drop_graph(p);                       // captured graphs hold the other activation kernel
}

int oiva_plan_fused_connect(oiva_plan* p, oiva_xchg* x) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    NEED(!p->fx_loopback, OIVA_ERR_STATE, "the plan runs the loop-back exchange (oiva_plan_fused_loopback(p, 0) switches it off)");
    DeviceGuard guard(p->device);
    if (!x) {
        HIP_TRY(hipStreamSynchronize(p->stream));
        p->fx_on = false;
        p->fx_world = 1;
        p->fx_rank = 0;
        for (auto& g : p->fx_gath) g = nullptr;
        if (p->fx_state) HIP_TRY(hipMemset(p->fx_state, 0, sizeof(unsigned)));     // a wait that gave up is history now
        return drop_graph(p);
    }
    char* peers[OIVA_XCHG_MAX_RANKS];
    int rank = 0, world = 1;
    size_t slot = 0;
    NEED(xchg_peers(x, peers, &rank, &world, &slot) == 0, OIVA_ERR_STATE, "exchange not connected");
    // slot = nblk * T * K * 8 bytes: nblk block sums per rank (every rank the same; 1 = the rank's sum)
    const size_t word_bytes = (size_t)p->T * p->K * 8;
    NEED(slot % word_bytes == 0 && slot / word_bytes >= 1 && slot / word_bytes <= (size_t)kCanonBlocks, OIVA_ERR_ARG,
         "exchange slot size must be nblk * T * K * 8 bytes, nblk = 1 ... 8 block sums per rank");
    const int nblk = (int)(slot / word_bytes);
    NEED(p->pw.nb % nblk == 0, OIVA_ERR_ARG, "the plan's 64-bin parts do not divide into that many blocks");
    NEED((world - 1) * nblk <= kCanonBlocks, OIVA_ERR_ARG, "the activation kernel polls at most 8 words of the other ranks per frame and source (9 ranks of one sum each)");
    int rc = fused_setup(p);
    if (rc) return rc;
    // the epochs of this connection start at 1 again: words left in the own gather buffer by an earlier connection (tagged
    // 1 ... n) must not pass for current ones.  The other ranks store into this buffer from their first iteration on, so
    // the callers rendezvous between connecting and iterating (include/overiva_hip.h; sharded.py does).
    HIP_TRY(hipMemset(peers[rank], 0, (size_t)2 * world * slot));
    HIP_TRY(hipDeviceSynchronize());
    p->fx_nblk_own = p->fx_nblk_peer = nblk;
    for (int r = 0; r < OIVA_XCHG_MAX_RANKS; ++r) p->fx_gath[r] = peers[r];
    p->fx_rank = rank;
    p->fx_world = world;
    p->fx_on = true;
    return OIVA_OK;
}

int oiva_plan_fused_loopback(oiva_plan* p, int world) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    NEED(world >= 0 && world <= OIVA_XCHG_MAX_RANKS, OIVA_ERR_ARG, "bad number of ranks");
    DeviceGuard guard(p->device);
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->fx_loop_buf) {
        HIP_TRY(hipFree(p->fx_loop_buf));
        p->fx_loop_buf = nullptr;
    }
    if (p->fx_loopback) {
        p->fx_on = p->fx_loopback = false;
        p->fx_world = 1;
        for (auto& g : p->fx_gath) g = nullptr;
        int rc = drop_graph(p);
        if (rc) return rc;
    }
    if (world <= 1) return OIVA_OK;
    NEED(!p->fx_on, OIVA_ERR_STATE, "the plan is connected to other ranks (oiva_plan_fused_connect(p, NULL) disconnects)");
    NEED(p->F == p->F_total, OIVA_ERR_STATE, "loop-back plays the other ranks with zeros: the plan must own all bins");
    // own blocks: those of the single-GPU sum (the result keeps its bits); per phantom rank the words a real rank of that
    // world would send: 8 / world blocks
    const int bsz = (p->pw.nb + kCanonBlocks - 1) / kCanonBlocks;
    p->fx_nblk_own = (p->pw.nb + bsz - 1) / bsz;
    p->fx_nblk_peer = std::max(1, kCanonBlocks / world);
    NEED((world - 1) * p->fx_nblk_peer <= kCanonBlocks, OIVA_ERR_ARG, "at most 9 ranks");
    const size_t bytes = (size_t)2 * world * p->fx_nblk_peer * p->T * p->K * 8;
    HIP_TRY(hipExtMallocWithFlags((void**)&p->fx_loop_buf, bytes, hipDeviceMallocFinegrained));
    HIP_TRY(hipMemset(p->fx_loop_buf, 0, bytes));
    int rc = fused_setup(p);
    if (rc) return rc;
    for (int r = 0; r < world; ++r) p->fx_gath[r] = p->fx_loop_buf;
    p->fx_rank = 0;
    p->fx_world = world;
    p->fx_loopback = true;
    p->fx_on = true;
    return OIVA_OK;
}

int oiva_plan_fused_debug(oiva_plan* p, int timeout_ms, int stall) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    DeviceGuard guard(p->device);
    p->fx_timeout_ms = timeout_ms;
    p->fx_stall = stall;
    HIP_TRY(hipStreamSynchronize(p->stream));
    return drop_graph(p);                       // (both are kernel arguments of captured launches)
}

int oiva_plan_iterate_timed(oiva_plan* p, int n, float* total_ms, float* per_kernel_ms) {
    int rc = check_ready(p);
    if (rc) return rc;
    NEED(n >= 1 && total_ms, OIVA_ERR_ARG, "bad arguments");
    NEED(p->F == p->F_total, OIVA_ERR_STATE, "timed iterate needs a plan that owns all bins");
    DeviceGuard guard(p->device);
    hipEvent_t e_begin = p->ev[0], e_end = p->ev[1];
    if (!per_kernel_ms) {
        HIP_TRY(hipEventRecord(e_begin, p->stream));
        if ((rc = oiva_plan_iterate(p, n))) return rc;
        HIP_TRY(hipEventRecord(e_end, p->stream));
        HIP_TRY(hipEventSynchronize(e_end));
        HIP_TRY(hipEventElapsedTime(total_ms, e_begin, e_end));
        return OIVA_OK;
    }
    // per-kernel: an event before every launch and one after the last launch of each iteration, all
    // recorded without draining the stream; elapsed times are read after one final synchronisation.
    for (int s = 0; s < OIVA_N_STAGES; ++s) per_kernel_ms[s] = 0.f;
    *total_ms = 0.f;
    const int per_it = OIVA_N_STAGES + 1;
    std::vector<hipEvent_t> pool((size_t)n * per_it, nullptr);
    auto destroy = [&]() {
        for (hipEvent_t e : pool)
            if (e) (void)hipEventDestroy(e);
    };
    for (auto& e : pool) {
        hipError_t err = hipEventCreate(&e);
        if (err != hipSuccess) {
            destroy();
            return fail(OIVA_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(err));
        }
    }
    std::vector<hipEvent_t> kev((size_t)2 * n, nullptr);
    for (auto& e : kev)
        if (hipEventCreate(&e) != hipSuccess) e = nullptr;
    hipError_t err = hipSuccess;
    for (int it = 0; it < n && err == hipSuccess && rc == OIVA_OK; ++it) {
        hipEvent_t* e = pool.data() + (size_t)it * per_it;
        err = hipEventRecord(e[0], p->stream);
        if (err == hipSuccess && !(rc = stage_power(p))) err = hipEventRecord(e[1], p->stream);
        if (err == hipSuccess && !rc && !(rc = stage_activation(p, p->Ppart, p->pw.nb))) err = hipEventRecord(e[2], p->stream);
        const bool fused = cov_update_applies(p);          // covariance + update as one launch: all of it is stage 2, stage 3 is empty
        if (err == hipSuccess && !rc) {
            arm_kernel_timer(kev[2 * it], kev[2 * it + 1]);    // the covariance kernel's own start / stop (see launch_dominant)
            rc = fused ? stage_cov_update(p) : stage_cov(p);
            arm_kernel_timer(nullptr, nullptr);
            if (!rc) err = hipEventRecord(e[3], p->stream);
        }
        if (err == hipSuccess && !rc && !(fused ? OIVA_OK : (rc = stage_update(p, false)))) err = hipEventRecord(e[4], p->stream);
    }
    if (err == hipSuccess && rc == OIVA_OK) err = hipStreamSynchronize(p->stream);
    for (int it = 0; it < n && err == hipSuccess && rc == OIVA_OK; ++it) {
        hipEvent_t* e = pool.data() + (size_t)it * per_it;
        for (int s = 0; s < OIVA_N_STAGES && err == hipSuccess; ++s) {
            float ms = 0.f;
            err = hipEventElapsedTime(&ms, e[s], e[s + 1]);
            // the covariance stage is reported as the duration of its kernel proper (events attached to the dispatch),
            // which is what the roofline is about and what rocprofv3 shows; the bracketing events add ~3 us of gaps
            float kms = 0.f;
            // (where the stage is two launches -- a weights pre-pass in front of the kernel: 9..16 channels, the float64 kernel
            //  of 8 channels, 8 channels with three or more sources -- it keeps its bracketed time, so that the stages still add
            //  up to the total)
            const bool one_launch = p->M <= 8 && !p->cov.pair32 && !(p->cov_f64() && cov_pair64_supported(p->M));
            if (s == 2 && one_launch && kev[2 * it] && kev[2 * it + 1] && hipEventElapsedTime(&kms, kev[2 * it], kev[2 * it + 1]) == hipSuccess &&
                kms > 0.f && kms <= ms)
                ms = kms;
            else if (s == 2)
                (void)hipGetLastError();
            per_kernel_ms[s] += ms;
        }
    }
    if (err == hipSuccess && rc == OIVA_OK) err = hipEventElapsedTime(total_ms, pool[0], pool[(size_t)n * per_it - 1]);
    destroy();
    for (hipEvent_t e : kev)
        if (e) (void)hipEventDestroy(e);
    if (rc) return rc;
    HIP_TRY(err);
    return OIVA_OK;
}

int oiva_plan_get_cov_splits(oiva_plan* p, int* nsplit) {
    NEED(p && nsplit, OIVA_ERR_ARG, "null argument");
    *nsplit = p->cov.nsplit;
    return OIVA_OK;
}

int oiva_plan_set_cov_splits(oiva_plan* p, int nsplit) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    NEED(nsplit >= 0 && nsplit <= p->T, OIVA_ERR_ARG, "bad split count");
    DeviceGuard guard(p->device);
    HIP_TRY(hipStreamSynchronize(p->stream));
    int rc = drop_graph(p);
    if (rc) return rc;
    choose_cov_geom(p, nsplit);
    return ensure_vpart(p);
}

int oiva_plan_set_cov_quad(oiva_plan* p, int enable, int* active) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    DeviceGuard guard(p->device);
    HIP_TRY(hipStreamSynchronize(p->stream));
    int rc = drop_graph(p);
    if (rc) return rc;
    p->cov_quad_on = enable != 0;
    choose_cov_geom(p, 0);
    if (active) *active = p->cov.quad || p->cov.half16;
    return ensure_vpart(p);
}

int oiva_plan_set_fuse_cov_update(oiva_plan* p, int enable, int* active) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    DeviceGuard guard(p->device);
    if (enable >= 0) {
        HIP_TRY(hipStreamSynchronize(p->stream));
        int rc = drop_graph(p);
        if (rc) return rc;
        p->fuse_cov_update = enable != 0;
    }
    if (active) *active = cov_update_applies(p) ? 1 : 0;
    return OIVA_OK;
}

int oiva_plan_set_cov_hmfma(oiva_plan* p, int enable) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    DeviceGuard guard(p->device);
    HIP_TRY(hipStreamSynchronize(p->stream));
    int rc = drop_graph(p);
    if (rc) return rc;
    p->cov_hmfma_on = enable != 0;
    choose_cov_geom(p, 0);
    return ensure_vpart(p);
}

int oiva_plan_set_pow_splits(oiva_plan* p, int nsplit) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    NEED(nsplit >= 0 && nsplit <= p->T, OIVA_ERR_ARG, "bad split count");
    DeviceGuard guard(p->device);
    HIP_TRY(hipStreamSynchronize(p->stream));
    int rc = drop_graph(p);
    if (rc) return rc;
    choose_pow_geom(p, nsplit);
    return OIVA_OK;
}

int oiva_plan_use_graph(oiva_plan* p, int enable) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    DeviceGuard guard(p->device);
    p->use_graph = enable ? 1 : 0;
    if (!enable) return drop_graph(p);
    if (p->have_x && p->have_cx && p->have_w && (p->F == p->F_total || p->fx_on)) return build_graphs(p);
    return OIVA_OK;
}

int oiva_plan_set_precision(oiva_plan* p, int flags) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    NEED((flags & ~(OIVA_PREC_UPDATE_F64 | OIVA_PREC_UPDATE_ROWS | OIVA_PREC_COV_F64)) == 0, OIVA_ERR_ARG, "unknown precision flag");
    DeviceGuard guard(p->device);
    HIP_TRY(hipStreamSynchronize(p->stream));
    int rc = drop_graph(p);
    if (rc) return rc;
    if ((flags & OIVA_PREC_UPDATE_F64) && p->have_w && !p->what64_valid) {
        // switching the update to float64 mid-run: seed the complex128 copy from the complex64 state
        std::vector<double2> wh;
        if ((rc = download_what(p, wh)) || (rc = upload_what(p, wh))) return rc;
    }
    // (more than 8 channels: which float32 kernel takes the pass also depends on the arithmetic of the per-bin algebra;
    // 8 channels with three or more sources: the number of frame splits does)
    const bool cov_changed = ((flags ^ p->prec) & (OIVA_PREC_COV_F64 | ((p->M > 8 || p->cov.pair32) ? OIVA_PREC_UPDATE_F64 : 0))) != 0;
    p->prec = flags;
    if (cov_changed) {
        choose_cov_geom(p, 0);            // sources per pass and residency depend on the accumulator type
        if ((rc = ensure_vpart(p))) return rc;
    }
    return OIVA_OK;
}

// ---- X-resident iteration ------------------------------------------------------------------------------
int oiva_plan_set_resident(oiva_plan* p, int enable) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    if (!enable) {
        p->res_on = false;
        return OIVA_OK;
    }
    NEED(p->res_ok, OIVA_ERR_ARG,
         "shape does not qualify for the X-resident iteration (4 or 8 channels, 1 or 2 sources with background channels, "
         "16 bins x <= 256 frames per compute unit)");
    DeviceGuard guard(p->device);
    int rc = resident_alloc(p);
    if (rc) return rc;
    p->res_on = true;
    return OIVA_OK;
}

int oiva_plan_set_resident_splits(oiva_plan* p, int nsplit) {
    NEED(p && nsplit >= 0, OIVA_ERR_ARG, "bad arguments");
    NEED(!p->res_on, OIVA_ERR_STATE, "switch the resident iteration off before changing its geometry");
    DeviceGuard guard(p->device);
    ResidentGeom g;
    const bool ok = resident_geometry(p->T, p->F, p->M, p->K, p->n_cu, nsplit, &g);
    NEED(ok || nsplit == 0, OIVA_ERR_ARG, "the shape does not fit on chip with that many frame splits");
    if (p->res_block) {                        // buffers were sized for the old geometry
        HIP_TRY(hipStreamSynchronize(p->stream));
        HIP_TRY(hipFree(p->res_block));
        p->res_block = nullptr;
    }
    p->res_ok = ok;
    if (ok) p->rg = g;
    return OIVA_OK;
}

int oiva_plan_resident_loopback(oiva_plan* p, int world) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    NEED(world >= 0 && world <= OIVA_XCHG_MAX_RANKS, OIVA_ERR_ARG, "bad number of ranks");
    NEED(!p->res_on, OIVA_ERR_STATE, "switch the resident iteration off before changing its exchange");
    DeviceGuard guard(p->device);
    if (p->res_loop_buf) {
        HIP_TRY(hipStreamSynchronize(p->stream));
        HIP_TRY(hipFree(p->res_loop_buf));
        p->res_loop_buf = nullptr;
    }
    p->res_loopback = false;
    p->res_world = 1;
    p->res_rank = 0;
    for (auto& g : p->res_gath) g = nullptr;
    if (world <= 1) return OIVA_OK;
    NEED(p->res_ok, OIVA_ERR_ARG, "shape does not qualify for the X-resident iteration");
    NEED(p->F == p->F_total, OIVA_ERR_STATE, "loop-back plays the other ranks with zeros: the plan must own all bins");
    // the gather buffer of the multi-GPU exchange, same kind of memory (fine-grained, system-scope atomics), but nobody else
    // maps it: [2 (epoch parity)][world][NS * TW][K] floats
    const size_t bytes = (size_t)2 * world * p->rg.NS * p->rg.TW * p->K * sizeof(float);
    HIP_TRY(hipExtMallocWithFlags((void**)&p->res_loop_buf, bytes, hipDeviceMallocFinegrained));
    HIP_TRY(hipMemset(p->res_loop_buf, 0, bytes));
    for (int r = 0; r < world; ++r) p->res_gath[r] = p->res_loop_buf;
    p->res_world = world;
    p->res_loopback = true;
    if (p->res_block) {                        // epochs restart with the fresh buffer
        HIP_TRY(hipStreamSynchronize(p->stream));
        HIP_TRY(hipMemset(p->res_block, 0, p->res_block_bytes));
        p->res_epoch = 0;
    }
    return OIVA_OK;
}

int oiva_plan_resident_connect(oiva_plan* p, oiva_xchg* x) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    NEED(!p->res_loopback, OIVA_ERR_STATE, "the plan runs the loop-back exchange (oiva_plan_resident_loopback(p, 0) switches it off)");
    if (!x) {                                  // back to a single rank
        p->res_world = 1;
        p->res_rank = 0;
        for (auto& g : p->res_gath) g = nullptr;
        return OIVA_OK;
    }
    NEED(p->res_ok, OIVA_ERR_ARG, "shape does not qualify for the X-resident iteration");
    char* peers[OIVA_XCHG_MAX_RANKS];
    int rank = 0, world = 1;
    size_t slot = 0;
    NEED(xchg_peers(x, peers, &rank, &world, &slot) == 0, OIVA_ERR_STATE, "exchange not connected");
    const size_t want = (size_t)p->rg.NS * p->rg.TW * p->K * sizeof(float);
    NEED(slot == want, OIVA_ERR_ARG, "exchange slot size must be frame_splits * frames_per_split * K * 4 bytes of THIS plan "
                                     "(every rank must run the same split geometry)");
    for (int r = 0; r < OIVA_XCHG_MAX_RANKS; ++r) p->res_gath[r] = peers[r];
    p->res_rank = rank;
    p->res_world = world;
    return OIVA_OK;
}

int oiva_plan_resident_info(oiva_plan* p, int* info) {
    NEED(p && info, OIVA_ERR_ARG, "null argument");
    const ResidentGeom& g = p->rg;
    const int v[OIVA_RESIDENT_INFO] = {p->res_ok ? 1 : 0, p->res_on ? 1 : 0, g.NB, g.NS, g.TW, g.J, g.JR, g.lds_bytes,
                                       p->res_last_code, p->res_launches, p->res_fallbacks, g.J * kBlock * p->M * 8};
    for (int i = 0; i < OIVA_RESIDENT_INFO; ++i) info[i] = v[i];
    return OIVA_OK;
}

int oiva_plan_resident_phases(oiva_plan* p, double* phase_us, int* n_iter) {
    NEED(p && phase_us && n_iter, OIVA_ERR_ARG, "null argument");
    for (int i = 0; i < OIVA_RESIDENT_PHASES; ++i) phase_us[i] = 0.;
    *n_iter = 0;
    if (!p->res_stamps || p->res_stamped <= 0) return OIVA_OK;
    DeviceGuard guard(p->device);
    const int n = p->res_stamped;
    std::vector<unsigned long long> st((size_t)n * kResidentStamps);
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemcpy(st.data(), p->res_stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    static_assert(OIVA_RESIDENT_PHASES <= kResidentStamps - 1, "one phase between consecutive stamps");
    for (int it = 0; it < n; ++it)
        for (int i = 0; i < OIVA_RESIDENT_PHASES; ++i)
            phase_us[i] += (double)(st[(size_t)it * kResidentStamps + i + 1] - st[(size_t)it * kResidentStamps + i]) * 0.01 / n;   // 100 MHz ticks
    *n_iter = n;
    return OIVA_OK;
}

int oiva_plan_resident_trace(oiva_plan* p, int enable, unsigned long long* stamps_host, int* n_wg, int* n_iter) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    DeviceGuard guard(p->device);
    p->res_trace = enable != 0;
    if (n_wg) *n_wg = p->rg.NB * p->rg.NS;
    if (n_iter) *n_iter = p->res_trace_buf ? p->res_trace_iters : 0;
    if (stamps_host && p->res_trace_buf) {
        HIP_TRY(hipStreamSynchronize(p->stream));
        HIP_TRY(hipMemcpy(stamps_host, p->res_trace_buf,
                          (size_t)p->rg.NB * p->rg.NS * p->res_trace_iters * kResidentStamps * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    return OIVA_OK;
}

int oiva_plan_resident_debug(oiva_plan* p, int timeout_ms, int stall_block) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    p->res_timeout_ms = timeout_ms;
    p->res_stall = stall_block;
    p->res_stall_iter = 0;
    return OIVA_OK;
}

int oiva_plan_resident_debug_from(oiva_plan* p, int timeout_ms, int stall_block, int first_stalled_iteration) {
    NEED(p && first_stalled_iteration >= 0, OIVA_ERR_ARG, "bad arguments");
    p->res_timeout_ms = timeout_ms;
    p->res_stall = stall_block;
    p->res_stall_iter = first_stalled_iteration;
    return OIVA_OK;
}

// ---- OGIVE (reference ive.py:33-256) ----------------------------------------------------------------
int oiva_plan_ogive_begin(oiva_plan* p, int update_mode, int model) {
    int rc = check_ready(p);
    if (rc) return rc;
    NEED(p->K == 1, OIVA_ERR_ARG, "OGIVE extracts one source: create the plan with K = 1");
    NEED(p->F == p->F_total, OIVA_ERR_STATE, "OGIVE is not bin-sharded");
    NEED(update_mode >= OIVA_OGIVE_DEMIX && update_mode <= OIVA_OGIVE_SWITCHING, OIVA_ERR_ARG, "unknown update mode");
    NEED(model == OIVA_MODEL_LAPLACE || model == OIVA_MODEL_GAUSS, OIVA_ERR_ARG, "unknown model");
    DeviceGuard guard(p->device);
    const size_t F = p->F, M = p->M;
    if (p->og_bufs.empty()) {
        hipError_t e = hipSuccess;
        auto alloc = [&](size_t bytes) -> void* {
            void* ptr = nullptr;
            if (e == hipSuccess) e = dev_malloc(&ptr, bytes);
            if (ptr) p->og_bufs.push_back(ptr);
            return ptr;
        };
        p->og.CxInv = (double2*)alloc(F * M * M * sizeof(double2));
        p->og.CxNorm = (double*)alloc(F * sizeof(double));
        p->og.A = (double2*)alloc(F * M * sizeof(double2));
        p->og.Delta = (double2*)alloc(F * M * sizeof(double2));
        p->og.Lambda = (double*)alloc(F * sizeof(double));
        p->og.DoA = (int*)alloc(F * sizeof(int));
        p->og.DoW = (int*)alloc(F * sizeof(int));
        p->og.Dnorm = (double*)alloc(F * sizeof(double));
        p->og.ctrl = (int*)alloc(4 * sizeof(int));
        p->og.maxdelta = (double*)alloc(2 * sizeof(double));
        if (e != hipSuccess) return fail(OIVA_ERR_HIP, std::string("allocation failed: ") + hipGetErrorString(e));
    }
    if (!p->what64_valid) {              // the step kernel reads and writes the complex128 copy of w
        std::vector<double2> wh;
        if ((rc = download_what(p, wh)) || (rc = upload_what(p, wh))) return rc;
    }
    if (p->og_graph) {                   // a cached chunk of epochs was captured for the previous update mode / model
        HIP_TRY(hipStreamSynchronize(p->stream));
        HIP_TRY(hipGraphExecDestroy(p->og_graph));
        p->og_graph = nullptr;
    }
    p->og.Cx = p->Cx;
    p->og.What = p->What;
    p->og.What64 = p->What64;
    p->og_mode = update_mode;
    p->og_model = model;
    HIP_TRY(launch_ogive_init(p->stream, p->og, p->F, p->M, update_mode));
    p->og_ready = true;
    return OIVA_OK;
}

int oiva_plan_ogive_iterate(oiva_plan* p, int first_epoch, int n, double step_size, double tol, int* epochs_run,
                            int* converged, double* max_delta) {
    int rc = check_ready(p);
    if (rc) return rc;
    NEED(p->og_ready, OIVA_ERR_STATE, "call oiva_plan_ogive_begin first");
    NEED(n >= 0 && first_epoch >= 0, OIVA_ERR_ARG, "negative epoch count");
    DeviceGuard guard(p->device);
    int before[2] = {0, 0};
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemcpy(before, p->og.ctrl, sizeof(before), hipMemcpyDeviceToHost));
    const int amodel = p->og_model == OIVA_MODEL_LAPLACE ? kModelOgiveLaplace : OIVA_MODEL_GAUSS;
    auto epochs = [&](int e0, int count) -> int {
        for (int e = e0; e < e0 + count; ++e) {
            if (p->og_mode == OIVA_OGIVE_SWITCHING && e % 10 == 0) HIP_TRY(launch_ogive_switch(p->stream, p->og, p->F, p->M));   // ive.py:192-193
            int r = stage_power(p);                                                             // ive.py:196 + the norm of :210/:213
            if (r) return r;
            HIP_TRY(launch_activation(p->stream, p->Ppart, p->pw.nb, p->R, p->T, 1, amodel, p->F));   // ive.py:209-217 (floor + 1/r in the consumer)
            HIP_TRY(launch_cov(p->stream, p->X, p->X_pad, p->R, p->Plocal, p->wscale, p->model, /*raw: weights 1 / max(r, eps)*/ 1, p->Vpart,
                               p->cov_f64(), p->T, p->F, p->M, 1, p->cov));                    // ive.py:221-227
            HIP_TRY(launch_ogive_step(p->stream, p->og, p->Vpart, p->vpart_f64(), p->cov.nsplit, p->T, p->F, p->M, step_size,
                                      tol));                                                     // ive.py:228-246
        }
        return OIVA_OK;
    };
    constexpr int kMinGraphEpochs = 8;
    if (n >= kMinGraphEpochs) {
        // a chunk of n epochs as one hipGraph, cached while (n, position in the 10-epoch switching cadence, step
        // size, tolerance) stay the same -- the usual case: the host runs equal chunks until the rule is met
        const int phase = first_epoch % 10;
        if (!p->og_graph || p->og_graph_n != n || p->og_graph_phase != phase || p->og_graph_mu != step_size ||
            p->og_graph_tol != tol) {
            if (p->og_graph) HIP_TRY(hipGraphExecDestroy(p->og_graph));
            p->og_graph = nullptr;
            hipGraph_t graph = nullptr;
            HIP_TRY(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
            rc = epochs(phase, n);
            hipError_t e = hipStreamEndCapture(p->stream, &graph);
            if (rc) {
                if (graph) (void)hipGraphDestroy(graph);
                return rc;
            }
            HIP_TRY(e);
            e = hipGraphInstantiate(&p->og_graph, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            HIP_TRY(e);
            p->og_graph_n = n;
            p->og_graph_phase = phase;
            p->og_graph_mu = step_size;
            p->og_graph_tol = tol;
        }
        HIP_TRY(hipGraphLaunch(p->og_graph, p->stream));
    } else if ((rc = epochs(first_epoch, n))) {
        return rc;
    }
    p->wscale_pending = false;
    int after[2] = {0, 0};
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemcpy(after, p->og.ctrl, sizeof(after), hipMemcpyDeviceToHost));
    if (epochs_run) *epochs_run = after[1] - before[1];
    if (converged) *converged = after[0];
    if (max_delta) HIP_TRY(hipMemcpy(max_delta, p->og.maxdelta, sizeof(double), hipMemcpyDeviceToHost));
    return OIVA_OK;
}

// ---- test-only stage access -----------------------------------------------------------------------
int oiva_test_set_rinv(oiva_plan* p, const float* rinv_host) {
    NEED(p && rinv_host, OIVA_ERR_ARG, "null argument");
    DeviceGuard guard(p->device);
    // the device keeps r, not 1/r: store the reciprocal and tell the covariance pass to skip gamma
    std::vector<float> r((size_t)p->T * p->K);
    for (size_t i = 0; i < r.size(); ++i) r[i] = 1.f / rinv_host[i];
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemcpy(p->R, r.data(), r.size() * sizeof(float), hipMemcpyHostToDevice));
    p->raw_weights = 1;
    return OIVA_OK;
}

int oiva_test_get_rinv(oiva_plan* p, float* rinv_host, float* wscale_host) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    DeviceGuard guard(p->device);
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (rinv_host) {
        // same formula as oiva::activation_weight on the r the device holds
        const int T = p->T, K = p->K;
        std::vector<float> r((size_t)T * K);
        HIP_TRY(hipMemcpy(r.data(), p->R, r.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int k = 0; k < K; ++k) {
            double s = 0.;
            for (int t = 0; t < T; ++t) s += (double)r[(size_t)t * K + k];
            const float ginv = p->raw_weights ? 1.f : 1.f / (float)(s / (double)T);
            for (int t = 0; t < T; ++t) {
                float rn = r[(size_t)t * K + k] * ginv;
                rn = rn < 1e-15f ? 1e-15f : rn;
                rinv_host[(size_t)t * K + k] = 1.f / rn;
            }
        }
    }
    if (wscale_host) HIP_TRY(hipMemcpy(wscale_host, p->wscale, (size_t)p->K * sizeof(float), hipMemcpyDeviceToHost));
    return OIVA_OK;
}

int oiva_test_run_weighted_cov(oiva_plan* p) {
    NEED(p, OIVA_ERR_ARG, "null plan");
    NEED(p->have_x, OIVA_ERR_STATE, "X not set");
    DeviceGuard guard(p->device);
    int rc = ensure_pad(p);
    if (rc) return rc;
    return stage_cov(p);
}

int oiva_test_get_v(oiva_plan* p, void* V_host, int f64) {
    NEED(p && V_host, OIVA_ERR_ARG, "null argument");
    DeviceGuard guard(p->device);
    const int F = p->F, M = p->M, K = p->K;
    const long long nfk = (long long)F * K * M * M;
    HIP_TRY(launch_sum_parts(p->stream, p->Vpart, p->vpart_f64(), p->cov.nsplit, p->scratch_p, nfk, 1. / (double)p->T));
    HIP_TRY(launch_unpack_herm(p->stream, p->scratch_p, p->scratch_c, f64 != 0, (long long)F * K, M));
    HIP_TRY(hipStreamSynchronize(p->stream));
    const size_t esz = f64 ? sizeof(double2) : sizeof(float2);
    std::vector<char> fk((size_t)nfk * esz);
    HIP_TRY(hipMemcpy(fk.data(), p->scratch_c, fk.size(), hipMemcpyDeviceToHost));
    // device order is [F][K][M][M]; the oracle's is (K, F, M, M)
    char* out = (char*)V_host;
    const size_t mm = (size_t)M * M * esz;
    for (int f = 0; f < F; ++f)
        for (int k = 0; k < K; ++k)
            std::memcpy(out + ((size_t)k * F + f) * mm, fk.data() + ((size_t)f * K + k) * mm, mm);
    return OIVA_OK;
}

int oiva_test_run_update(oiva_plan* p) {
    int rc = check_ready(p);
    if (rc) return rc;
    DeviceGuard guard(p->device);
    return stage_update(p, false);
}

int oiva_test_get_what(oiva_plan* p, void* What_host, int f64) {
    NEED(p && What_host, OIVA_ERR_ARG, "null argument");
    DeviceGuard guard(p->device);
    std::vector<double2> wh;
    int rc = download_what(p, wh);
    if (rc) return rc;
    if (f64) {
        std::memcpy(What_host, wh.data(), wh.size() * sizeof(double2));
    } else {
        float2* o = static_cast<float2*>(What_host);
        for (size_t i = 0; i < wh.size(); ++i) o[i] = make_float2((float)wh[i].x, (float)wh[i].y);
    }
    return OIVA_OK;
}

int oiva_test_set_what(oiva_plan* p, const void* What_host, int f64) {
    NEED(p && What_host, OIVA_ERR_ARG, "null argument");
    DeviceGuard guard(p->device);
    std::vector<double2> wh((size_t)p->F * p->M * p->M);
    if (f64) {
        std::memcpy(wh.data(), What_host, wh.size() * sizeof(double2));
    } else {
        const float2* in = static_cast<const float2*>(What_host);
        for (size_t i = 0; i < wh.size(); ++i) wh[i] = make_double2(in[i].x, in[i].y);
    }
    int rc = upload_what(p, wh);
    if (rc) return rc;
    p->have_w = true;
    p->wscale_pending = false;
    return OIVA_OK;
}

int oiva_test_time_stage(oiva_plan* p, int stage, int reps, float* avg_ms) {
    int rc = check_ready(p);
    if (rc) return rc;
    NEED(reps >= 1 && avg_ms && stage >= 0 && stage < OIVA_N_STAGES, OIVA_ERR_ARG, "bad arguments");
    DeviceGuard guard(p->device);
    auto run = [&]() -> int {
        switch (stage) {
            case 0: return stage_power(p);
            case 1: return stage_activation(p, p->Ppart, p->pw.nb);
            case 2: return cov_update_applies(p) ? stage_cov_update(p) : stage_cov(p);      // (one launch for both where the plan runs it so)
            default: return cov_update_applies(p) ? OIVA_OK : stage_update(p, false);
        }
    };
    if ((rc = run())) return rc;  // warm
    HIP_TRY(hipEventRecord(p->ev[0], p->stream));
    for (int i = 0; i < reps; ++i)
        if ((rc = run())) return rc;
    HIP_TRY(hipEventRecord(p->ev[1], p->stream));
    HIP_TRY(hipEventSynchronize(p->ev[1]));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, p->ev[0], p->ev[1]));
    *avg_ms = ms / reps;
    return OIVA_OK;
}

int oiva_test_run_power(oiva_plan* p, float* p_host) {
    NEED(p && p_host, OIVA_ERR_ARG, "null argument");
    NEED(p->have_x && p->have_w, OIVA_ERR_STATE, "X / W not set");
    DeviceGuard guard(p->device);
    int rc = stage_power(p);
    if (rc) return rc;
    const size_t n = (size_t)p->T * p->K;
    HIP_TRY(launch_sum_parts(p->stream, p->Ppart, false, p->pw.nb, p->scratch_p, (long long)n, 1.));
    HIP_TRY(hipStreamSynchronize(p->stream));
    std::vector<double> sum(n);
    HIP_TRY(hipMemcpy(sum.data(), p->scratch_p, n * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) p_host[i] = (float)sum[i];
    return OIVA_OK;
}

}  // extern "C"
