// Per-bin sequential algebra of determined AuxIVA at 9..16 channels (BASELINE configs[4]: 16 x 16), float64,
// ONE MATRIX ROW PER LANE: a wavefront works on FOUR bins, one per 16-lane DPP row.          reference overiva.py:176-190
//
// Same mathematics as update_det16_kernel (kernels_update16.hip): per source s
//     w = V_s^-1 u,   u = column s of C = (W_hat^H)^-1,   d = w^H u,   w /= sqrt(d)              (overiva.py:181-186)
//     C' = C - (u / y_s) (y - sqrt(d) e_s^T),   y = w^H C                                        (the new row s of W_hat^H)
// with C from ONE pivoted elimination per bin and iteration.  What changed is who holds what.  update_det16_kernel spreads
// one 16 x 16 matrix over the 64 lanes of a wave (4 entries per lane): every elimination step publishes the pivot row through
// LDS and repeats its scalar work (reciprocal + two Newton steps, selects, waits) on all 64 lanes -- 75 instructions per
// step, of which 27 update entries; 16 sources x 16 steps of that are 142 us at 2048 bins, the wave issuing for two thirds of
// the time.  Here lane (g, i) holds ROW i of the matrices of bin 4 b + g:
//   * the multiplier of a row operation, A[i][k] / A[k][k], is the lane's own;
//   * the pivot row reaches the 16 lanes of its bin by the DPP control row_newbcast:k (lane k of every 16-lane row; the
//     64-bit form v_mov_b64_dpp exists for exactly this control) -- no LDS, no wait;
//   * eliminated columns are skipped exactly (15 - k entries in step k, not "elements that still hold a live column");
//   * the scalar work of a step is paid once for four bins.
// About 45 instructions per step and FOUR bins: a sixth of the instructions per bin.
// C is kept by COLUMNS (lane c holds C[0..15][c]): y = w^H C and the rank-one update then need only broadcasts of w and
// u / y_s (again row_newbcast), and u = C e_s -- the 16 registers of lane s -- reaches the rows through 1 KB of LDS.
//
// THREE WAVES per four bins.  The elimination of V_s does not depend on the chain through the sources -- only its right-hand
// side u does -- so it is split off: waves 1 and 2 (each takes the next source nobody has, from a counter) add the frame splits'
// partial covariances (loaded straight into registers, a source ahead; float64, or -- opt-in -- the float32 blocks of the
// matrix-core covariance kernel), eliminate V_s with the multipliers A[i][k] / A[k][k] RECORDED, and hand the lanes' 16
// multipliers + reciprocal pivot to wave 0 through LDS (17 KB per source, three buffers, a flag word each way per buffer);
// wave 0 inverts W_hat^H (a loop of four steps: sixteen unrolled ones were bound by instruction fetch), then per source
// applies the recorded row operations to u (64 multiply-adds), forms y, updates C and stores the new row of W_hat.  Same
// operations on the same numbers as one wave doing all of it in turn.  142 -> 59 us at 2048 bins; 46 us of that is reading the
// 277 MB of float64 partials.
#include "oiva_device.h"

#include <cstdint>
#include <cstdlib>
#include <type_traits>

namespace oiva {
namespace {

constexpr int N = 16;
constexpr int kBinsPerWaveR = 4;
constexpr int kMaxSplitsR = 4;      // frame splits the eliminating waves add, two in flight at a time (more: the one-matrix-per-wave kernel)

struct alignas(16) Z {       // (16-byte alignment: one ds_read_b128 / ds_write_b128 with a 16-bit immediate offset per value in LDS)
    double re, im;
};

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

// value of lane K of the caller's 16-lane row (v_mov_b64_dpp row_newbcast:K)
template <int K>
__device__ __forceinline__ double bc(double x) {
    long long b = __builtin_bit_cast(long long, x);
    long long r = __builtin_amdgcn_update_dpp(b, b, 0x150 + K, 0xf, 0xf, false);
    return __builtin_bit_cast(double, r);
}
// acc += (lane K of the row: b) * m  /  acc -= ... : v_fmac_f64_dpp, the broadcast as the DPP control of the multiply-add itself
// (64-bit DPP exists for row_newbcast only).  Inline assembly: hipcc forms v_mov_b64_dpp + v_fmac from the builtin, with a
// copy of the source in front of every move (the move's destination is tied to its "old" operand) and a wait state -- ten
// instructions per complex entry where four do.  No wait state is needed here: the hazard "vector write, then DPP read of the
// same register within two instructions" would hand the reader the register's PREVIOUS content -- and the only lane whose
// content is read, lane K, has the multiplier 0 in every use below, so its registers never change.
template <int K>
__device__ __forceinline__ void fmacb(double& acc, double b, double m) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(m), "n"(K));
}
template <int K>
__device__ __forceinline__ void fmacb_neg(double& acc, double b, double m) {
    asm("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(m), "n"(K));
}
// a -= m * (lane K's r)      (r may be a itself: see above)
template <int K>
__device__ __forceinline__ void zsubmul_b(Z& a, Z m, const Z& r) {
    fmacb_neg<K>(a.re, r.re, m.re);
    fmacb<K>(a.re, r.im, m.im);
    fmacb_neg<K>(a.im, r.im, m.re);
    fmacb_neg<K>(a.im, r.re, m.im);
}
// two wait states between whatever vector instruction the compiler placed last and the DPP reads of the assembly that follows
// (hipcc's hazard recogniser does not look into inline assembly): in front of a block whose broadcast operands were just computed
__device__ __forceinline__ void dpp_fence() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
}
// value of lane K of the row (v_mov_b64_dpp behind its two wait states; the builtin costs a register copy more)
template <int K>
__device__ __forceinline__ double bcm(double x) {
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(K));
    return r;
}
template <int CTRL>
__device__ __forceinline__ unsigned dppu(unsigned x) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xf, 0xf, false);
}

// 1 / a: hardware seed + two Newton steps (tools/probe/rcp_rsq_accuracy.hip: last-bit accurate)
__device__ __forceinline__ double rcp_nr(double a) {
    double d = __builtin_amdgcn_rcp(a);
    d = fma(fma(-a, d, 1.0), d, d);
    d = fma(fma(-a, d, 1.0), d, d);
    return d;
}
// 1 / sqrt(a), the same way (the IEEE division and square root are about sixty dependent instructions on the chain's critical path)
__device__ __forceinline__ double rsq_nr(double a) {
    double r = __builtin_amdgcn_rsq(a);
    r = fma(0.5 * r, fma(-a * r, r, 1.0), r);
    return fma(0.5 * r, fma(-a * r, r, 1.0), r);
}
__device__ __forceinline__ Z zmul(Z a, Z b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ Z zinv_fast(Z a) {
    const double d = rcp_nr(a.re * a.re + a.im * a.im);
    return {a.re * d, -a.im * d};
}
// a -= m * r
__device__ __forceinline__ void zsubmul(Z& a, Z m, Z r) {
    a.re = fma(-m.re, r.re, fma(m.im, r.im, a.re));
    a.im = fma(-m.re, r.im, fma(-m.im, r.re, a.im));
}

// one wavefront, LDS operations complete in order: waiting for the wave's own LDS traffic and pinning the instruction
// order is all the synchronisation a write -> read exchange needs
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
}


constexpr int kFacPlanes = N + 1;       // 16 multipliers + the reciprocal pivot, per lane

// (tools/build_variant.py ... -DOIVA_R16_TRACE: clock stamps of workgroup 0, read by tools/r6/det16_trace.py)
#ifdef OIVA_R16_TRACE
__device__ unsigned long long g_r16_trace[3 * 80];
__device__ unsigned g_r16_hwid[2 * 3 * 1024];           // (HW_ID, XCC_ID) of every wave
#define R16_STAMP(slot)                                                                                   \
    do {                                                                                                  \
        if (blockIdx.x == 0 && lane == 0) g_r16_trace[wave * 80 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define R16_STAMP(slot) \
    do {                \
    } while (0)
#endif

constexpr int kFacBufs = 3;
struct LdsR {
    // (the small, busy arrays first: an LDS instruction's immediate offset reaches 64 KB)
    Z prow[kBinsPerWaveR][N];                           // C-phase: pivot row
    Z ppiv[kBinsPerWaveR];                              //          pivot element
    Z ucol[kBinsPerWaveR][N];                           // u = C e_s by row index
    Z ys[kBinsPerWaveR];                                // y_s
    int ready[kFacBufs], consumed[kFacBufs];            // sources handed over / taken, + 1, per buffer
    int claim;                                          // next source nobody eliminates yet
    int sq_done;                                        // wave 0 has C in registers: the last buffer is free
    int w_requested;                                    // wave 0 has requested W_hat: the eliminating waves may load
    // [buffer = source mod 3][plane][lane]: the recorded elimination of V_s.  Three buffers: the eliminating waves run up to three
    // sources ahead of the chain, which starts late (behind the inversion of W_hat^H).  The LAST one doubles as `sq` of that
    // inversion -- the inverse's rows, filed under the column they pivoted, [bin][16][16] --: source 2 waits for sq_done.
    Z fac[kFacBufs][kFacPlanes][64];
    double vsum[2][kBinsPerWaveR * N * N + 2];          // per eliminating wave: sum of the splits / T (packed Hermitian blocks of 256; + 2: the neighbour read of the last diagonal)
};
static_assert(sizeof(Z) * kBinsPerWaveR * N * N <= sizeof(Z) * kFacPlanes * 64, "sq fits a buffer");

// one wavefront against LDS: LDS operations of a wave complete in order
__device__ __forceinline__ void lds_wait() { __builtin_amdgcn_s_waitcnt(0xc07f); }   // lgkmcnt(0)

__device__ __forceinline__ void spin_until(const int* flag, int value) {
#ifndef OIVA_R16_NOSYNC          // (variant builds: each role alone, no hand-shake -- timing only, wrong results)
    while (*const_cast<const volatile int*>(flag) < value) __builtin_amdgcn_s_sleep(1);
#endif
}

template <typename VT>
__global__ __launch_bounds__(192) void update_det16r_kernel(UpdateArgs a) {
    __shared__ LdsR s;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
    const int M = a.M, NA = M * M;
    const int fraw = blockIdx.x * kBinsPerWaveR + g;
    const bool live = fraw < a.F;                          // (the last workgroup of a bin count that is no multiple of 4)
    const int f = live ? fraw : a.F - 1;
    const Z zero = {0., 0.};
#ifdef OIVA_R16_TRACE
    if (lane == 0 && blockIdx.x < 1024) {
        g_r16_hwid[(blockIdx.x * 3 + wave) * 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        g_r16_hwid[(blockIdx.x * 3 + wave) * 2 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
    if (threadIdx.x < kFacBufs) {
        s.ready[threadIdx.x] = 0;
        s.consumed[threadIdx.x] = 0;
        s.claim = 0;
        s.sq_done = 0;
        s.w_requested = 0;
    }
    __syncthreads();

#ifdef OIVA_R16_ONLY_A
    if (wave != 0) return;
#endif
    if (wave != 0) {
        // =============== waves 1, 2: V_s of the sources b, b + 2, ... summed over the splits and eliminated ===============
        const int b = wave - 1;
        // (VT: float64 partials, or -- the matrix-core covariance kernel, round 6 -- float32 ones: each the float64 sum of 8 float32
        //  chains rounded once; either way added here in float64, in split order)
        const VT* vbase = static_cast<const VT*>(a.Vpart);
        const size_t vstride = (size_t)a.F * M * NA;
        const int nsplit = a.nsplit;
        const unsigned blk_bytes = (unsigned)NA * (unsigned)sizeof(VT);
        constexpr int kElems = 16 / (int)sizeof(VT);                   // values per 16-byte piece (2 | 4)
        constexpr int kSub = N * N / kElems / 64;                      // 64-piece (1 KB) parts of a bin's block of 256 values (2 | 1)
        constexpr int kPieces = kBinsPerWaveR * kSub;                  // pieces per lane and split (8 | 4): piece lane + 64 j
        // piece lane + 64 j of a split: bin j / kSub of the workgroup, 1 KB j % kSub of its block -- as the lane's byte offset from the
        // block of the workgroup's first bin (bins past F: the last bin's block again), so that a split of a source has ONE uniform
        // base address and a load is one instruction (scalar base + 32-bit lane offset)
        const int f_first = blockIdx.x * kBinsPerWaveR;
        unsigned poff[kPieces];
#pragma unroll
        for (int j = 0; j < kPieces; ++j) {
            const int fr = f_first + j / kSub;
            const unsigned off = (unsigned)((j % kSub) * 64 + lane) * 16u;
            // lanes past the block re-read its start (their sums are not used); odd M: the last piece runs over (the next block, or the buffer's slack)
            poff[j] = (unsigned)((fr < a.F ? fr : a.F - 1) - f_first) * (unsigned)(M * NA * (int)sizeof(VT)) + (off < blk_bytes ? off : 0u);
        }
        const char* wg_base = reinterpret_cast<const char*>(vbase + (size_t)f_first * M * NA);
        // The partials travel in two batches of (at most) two splits, 64 registers: splits 0, 1 of the next source of this wave are
        // requested before the elimination of the current one and added in the middle of it, where splits 2, 3 are requested; those
        // are added when the elimination is over.  (All four at once: 128 registers in flight, 324 with the rest -- one wave per SIMD.)
        using piece_t = std::conditional_t<sizeof(VT) == 8, double2, float4>;
        piece_t P[2][kPieces];
        double acc[kPieces][kElems];
        auto load_pair = [&](int src, int sp0) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (sp0 + u < nsplit) {                     // (uniform)
                    const char* sbase = wg_base + ((size_t)(sp0 + u) * vstride + (size_t)src * NA) * sizeof(VT);
#pragma unroll
                    for (int j = 0; j < kPieces; ++j) P[u][j] = *reinterpret_cast<const piece_t*>(sbase + poff[j]);
                }
        };
        // acc (+)= the batch, in split order (first: acc = split 0)
        auto add_pair = [&](int sp0) {
            __builtin_amdgcn_s_waitcnt(0x0f70);            // vmcnt(0)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (sp0 + u < nsplit) {
#pragma unroll
                    for (int j = 0; j < kPieces; ++j) {
                        double v[kElems];
                        if constexpr (sizeof(VT) == 8) {
                            v[0] = P[u][j].x, v[1] = P[u][j].y;
                        } else {
                            v[0] = (double)P[u][j].x, v[1] = (double)P[u][j].y, v[2] = (double)P[u][j].z, v[3] = (double)P[u][j].w;
                        }
#pragma unroll
                        for (int e = 0; e < kElems; ++e) acc[j][e] = sp0 + u == 0 ? v[e] : acc[j][e] + v[e];
                    }
                }
        };
        // where row i of a packed Hermitian block lies: index of (re, im) of entry (i, c); the imaginary part changes sign below the diagonal
        int voff[N];
#pragma unroll
        for (int c = 0; c < N; ++c) {
            const int lo = i < c ? i : c, hi = i < c ? c : i;
            voff[c] = g * (N * N) + ((i == c || hi >= M) ? (i < M ? i : 0) : herm_pair_index(M, lo, hi));
        }
        const double invT = 1.0 / (double)a.T;
        double* vs = &s.vsum[b][0];
        // The sources are not dealt out in advance: a wave takes the next one nobody has (a counter in LDS) when it starts on its
        // current one -- the hardware places the six waves of two workgroups on a CU's four SIMDs as {A}, {A, B}, {B, B}, {B}, and a
        // wave alone on its SIMD eliminates a source in half the time of one that shares it (2.5 against 5 us).  Source n goes
        // through buffer n & 1 whoever eliminated it; wave 0 takes them in order.
        auto claim = [&]() {
            int nx = 0;
            if (lane == 0) nx = atomicAdd(&s.claim, 1);
            return __builtin_amdgcn_readfirstlane(nx);
        };
        // (wave 0's 8 KB of W_hat first: behind the 32 KB every eliminating wave requests at once -- 34 MB on the chip -- its loads
        //  took 13 us to come back, and the chain starts only when the inversion is done)
        spin_until(&s.w_requested, 1);
        int src = claim();
        if (src < M) {
            load_pair(src, 0);
            add_pair(0);
            if (nsplit > 2) load_pair(src, 2);
        }
        while (src < M) {
            const int fb = src % kFacBufs;
            // the sum over the splits, in order, times 1 / T, in the layout the partials arrived in
            R16_STAMP(src * 4 + 0);
            if (nsplit > 2) add_pair(2);
            R16_STAMP(src * 4 + 1);
            lds_wait();
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < kPieces; ++j)
#pragma unroll
                for (int e = 0; e < kElems; e += 2)
                    reinterpret_cast<double2*>(vs)[((lane + 64 * j) * kElems + e) / 2] = make_double2(acc[j][e] * invT, acc[j][e + 1] * invT);
            const int nxt = claim();
            const bool more = nxt < M;
            if (more) load_pair(nxt, 0);                   // in flight while this source is eliminated
            lds_wait();
            __builtin_amdgcn_wave_barrier();
            // V_s, row i
            // (the imaginary part of the diagonal entry is whatever lies behind it in the block: no step reads it -- the pivot is
            //  V[k].re, and the pivot row's own multiplier is multiplied by 0)
            Z V[N];
            if (M == N) {                                   // (uniform)
#pragma unroll
                for (int c = 0; c < N; ++c) {
                    const double vr = vs[voff[c]], vi = vs[voff[c] + 1];
                    V[c] = {vr, i < c ? vi : -vi};
                }
            } else {
#pragma unroll
                for (int c = 0; c < N; ++c) {
                    V[c] = {i == c ? 1. : 0., 0.};
                    if (c < M) {                            // (uniform)
                        const double vr = vs[voff[c]], vi = vs[voff[c] + 1];
                        if (i < M) V[c] = {vr, i < c ? vi : -vi};
                    }
                }
            }
            dpp_fence();
            // Gauss-Jordan without pivot search (Hermitian positive definite; identity outside M x M), columns <= k skipped; the
            // pivot row by row_newbcast; the multiplier of step k stays in V[k] (0 on the pivot row itself)
            double dmine = 1.;
            auto step = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const double pk = bcm<k>(V[k].re);         // A[k][k] (real: Schur complements stay Hermitian)
                const double d = rcp_nr(pk);
                const bool rowk = i == k;
                dmine = rowk ? d : dmine;
                const double dm = rowk ? 0. : d;           // the pivot row's own multiplier is 0
                V[k] = {V[k].re * dm, V[k].im * dm};
                static_for<N - 1 - k>([&](auto jc) {
                    constexpr int c = k + 1 + decltype(jc)::value;
                    zsubmul_b<k>(V[c], V[k], V[c]);
                });
            };
#define OIVA_R16_STEP(c) \
    if (c < M) step(std::integral_constant<int, c>{});
            OIVA_R16_STEP(0) OIVA_R16_STEP(1) OIVA_R16_STEP(2) OIVA_R16_STEP(3) OIVA_R16_STEP(4) OIVA_R16_STEP(5)
            if (more) {                                     // (uniform) the next source's first two splits have arrived; its last two set out
                add_pair(0);
                if (nsplit > 2) load_pair(nxt, 2);
            }
            OIVA_R16_STEP(6)
            OIVA_R16_STEP(7) OIVA_R16_STEP(8) OIVA_R16_STEP(9) OIVA_R16_STEP(10) OIVA_R16_STEP(11) OIVA_R16_STEP(12)
            OIVA_R16_STEP(13) OIVA_R16_STEP(14) OIVA_R16_STEP(15)
#undef OIVA_R16_STEP
            R16_STAMP(src * 4 + 2);
            // hand-over: the buffer is free once wave 0 has taken source src - 3 (and, the last one, finished the inversion)
            if (src >= kFacBufs) spin_until(&s.consumed[fb], src - kFacBufs + 1);
            if (src == kFacBufs - 1) spin_until(&s.sq_done, 1);
#pragma unroll
            for (int k = 0; k < N; ++k) s.fac[fb][k][lane] = V[k];
            s.fac[fb][N][lane] = {dmine, 0.};
            lds_wait();
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) *const_cast<volatile int*>(&s.ready[fb]) = src + 1;
            R16_STAMP(src * 4 + 3);
            src = nxt;
        }
        return;
    }

#ifdef OIVA_R16_ONLY_B
    return;
#endif
    // =============== wave 0: C = (W_hat^H)^-1, then the chain through the sources ===============
    // (the chain is the critical path of the workgroup: where this wave shares its SIMD with an eliminating wave it goes first)
    __builtin_amdgcn_s_setprio(3);
    // ---- B = W_hat^H (identity outside M x M), rows scaled by 1 / wscale (overiva.py:163 / :167); lane (g, i): row i
    R16_STAMP(64);
    Z A[N];
#pragma unroll
    for (int c = 0; c < N; ++c) {
        A[c] = {i == c ? 1. : 0., 0.};
        if (i < M && c < M) {                              // (c < M is uniform)
            double vr, vi;
            load_what<double>(a, ((size_t)f * M + c) * M + i, vr, vi);
            A[c] = {vr, -vi};
        }
    }
    // (the loads above are requested, not yet back: the flag is set in program order behind their issue)
    __builtin_amdgcn_sched_barrier(0);
    if (lane == 0) *const_cast<volatile int*>(&s.w_requested) = 1;
    __builtin_amdgcn_sched_barrier(0);
    if (a.wscale != nullptr && i < M) {
        const double sc = 1.0 / (double)a.wscale[i];
#pragma unroll
        for (int c = 0; c < N; ++c) A[c] = {A[c].re * sc, A[c].im * sc};
    }
    // ---- C = B^-1: in-place Gauss-Jordan with partial pivoting, rows never move.  Step c: the pivot row p (largest |A[i][c]|
    //      among the rows not used yet) goes through LDS with 1 in slot c; A[i][j] -= (A[i][c] / A[p][c]) row[j] for every j with
    //      A[i][c] cleared first, which leaves -multiplier in slot c (column p of the would-be right-hand side) and 1 on the
    //      pivot row.  In the end row i, divided by its pivot, is row mycol of the inverse with its columns in the order "row that
    //      pivoted column j": inverse[r][j] = Zfinal[p_r][q_j], q_j = the column row j pivoted.
    Z C[N];
    {
        bool used = false;
        Z piv = {1., 0.};
        int mycol = i;
        // One step, on the column that sits in register U: the loop below runs four steps and then moves every register four places
        // down, so that the code of a step exists FOUR times, not sixteen -- the sixteen unrolled steps (190 instructions each,
        // executed once per workgroup, by all workgroups at the same moment) ran at 44 cycles per instruction: instruction-cache
        // misses, 24 us for the inversion; as a loop 10 us.
        auto step = [&](auto uc, int c) {
            constexpr int U = decltype(uc)::value;
            const Z aic = A[U];
            const float mag = (float)(aic.re * aic.re + aic.im * aic.im);
            unsigned key = used ? 0u : ((__float_as_uint(mag) & ~31u) | 16u | (unsigned)(15 - i));
            // the maximum over the 16 lanes of the row: row_ror:1, 2, 4, 8 as the DPP control of v_max_u32 itself (two wait states
            // between a vector write and a DPP read of the same register)
            asm volatile("s_nop 1\n\t"
                         "v_max_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\t"
                         "v_max_u32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\t"
                         "v_max_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\t"
                         "v_max_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf"
                         : "+v"(key));
            const int p = 15 - (int)(key & 15u);
            const bool mine = i == p;
            wave_lds_sync();
            if (mine) {
#pragma unroll
                for (int j = 0; j < N; ++j) s.prow[g][j] = j == U ? Z{1., 0.} : A[j];
                s.ppiv[g] = aic;
            }
            wave_lds_sync();
            // (every read of the pivot row requested before the first is waited for: hipcc's own schedule kept two in flight and paid
            //  nine LDS round trips per step)
            const Z apc = s.ppiv[g];
            Z row[N];
#pragma unroll
            for (int j = 0; j < N; ++j) row[j] = s.prow[g][j];
            __builtin_amdgcn_sched_barrier(0);
            used = used || mine;
            mycol = mine ? c : mycol;
            piv.re = mine ? apc.re : piv.re;
            piv.im = mine ? apc.im : piv.im;
            Z fct = zmul(aic, zinv_fast(apc));
            fct.re = mine ? 0. : fct.re;
            fct.im = mine ? 0. : fct.im;
            A[U] = {mine ? 1. : 0., 0.};
#pragma unroll
            for (int j = 0; j < N; ++j) zsubmul(A[j], fct, row[j]);
        };
        R16_STAMP(66);
#pragma unroll 1
        for (int c0 = 0; c0 < N; c0 += 4) {
            if (c0 + 0 < M) step(std::integral_constant<int, 0>{}, c0 + 0);
            if (c0 + 1 < M) step(std::integral_constant<int, 1>{}, c0 + 1);
            if (c0 + 2 < M) step(std::integral_constant<int, 2>{}, c0 + 2);
            if (c0 + 3 < M) step(std::integral_constant<int, 3>{}, c0 + 3);
            // registers four places down (column c0 + 4 to register 0, ...): after the fourth round they are back in place
            Z t0 = A[0], t1 = A[1], t2 = A[2], t3 = A[3];
#pragma unroll
            for (int j = 0; j + 4 < N; ++j) A[j] = A[j + 4];
            A[N - 4] = t0, A[N - 3] = t1, A[N - 2] = t2, A[N - 1] = t3;
        }
        R16_STAMP(72);
        wave_lds_sync();
        Z* sq = &s.fac[kFacBufs - 1][0][0];                 // (the last hand-over buffer, not yet in use)
        {
            const Z ip = zinv_fast(piv);
#pragma unroll
            for (int j = 0; j < N; ++j) sq[(g * N + mycol) * N + j] = zmul(A[j], ip);
        }
        wave_lds_sync();
        // lane j takes column j of the inverse: C[r][j] = sq[r][q_j]
#pragma unroll
        for (int r = 0; r < N; ++r) C[r] = sq[(g * N + r) * N + mycol];
        wave_lds_sync();
        if (lane == 0) *const_cast<volatile int*>(&s.sq_done) = 1;
    }


    R16_STAMP(65);
    for (int src = 0; src < M; ++src) {
        const int b = src % kFacBufs;
        R16_STAMP(src * 4 + 0);
        // u = C e_src: the 16 registers of lane src, to the rows through LDS
        wave_lds_sync();
        if (i == src) {
#pragma unroll
            for (int r = 0; r < N; ++r) s.ucol[g][r] = C[r];
        }
        wave_lds_sync();
        Z rhs = s.ucol[g][i];
        const Z ui = rhs;
        // the recorded elimination of V_src
        R16_STAMP(src * 4 + 1);
        spin_until(&s.ready[b], src + 1);
        R16_STAMP(src * 4 + 2);
        Z F[N];
#pragma unroll
        for (int k = 0; k < N; ++k) F[k] = s.fac[b][k][lane];
        const double dmine = s.fac[b][N][lane].re;
        lds_wait();
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) *const_cast<volatile int*>(&s.consumed[b]) = src + 1;
        // w = V^-1 u (not yet normalised): the row operations on u, then the division by the pivots
        dpp_fence();
        static_for<N>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if (k < M) zsubmul_b<k>(rhs, F[k], rhs);
        });
        const Z wi = {rhs.re * dmine, rhs.im * dmine};
        // y = w^H C: lane c forms column c, w_r by broadcast
        // (four running sums: a dependent v_fmac_f64 issues every 8 cycles, an independent one every 4)
        Z y = zero, y2 = zero;
        dpp_fence();
        static_for<N>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            fmacb<r>(y.re, wi.re, C[r].re);
            fmacb<r>(y2.re, wi.im, C[r].im);
            fmacb<r>(y.im, wi.re, C[r].im);
            fmacb_neg<r>(y2.im, wi.im, C[r].re);
        });
        y.re += y2.re;
        y.im += y2.im;
        // y_src = w^H u = w^H V w =: d (overiva.py:185; real for the exact w): the normalisation takes its real part, the
        // Sherman-Morrison step divides by the COMPLEX value the rounded w gives (see update_det_kernel)
        wave_lds_sync();
        if (i == src) s.ys[g] = y;
        wave_lds_sync();
        const Z ys = s.ys[g];
        const double d = ys.re;
        const double sc = rsq_nr(d);
        const Z gi = zmul(ui, zinv_fast(ys));
        const double sqd = d * sc;
        Z ye = y;
        if (i == src) ye.re -= sqd;
        // C' = C - (u / y_src) (y - sqrt(d) e_src^T): lane c updates column c, g_r by broadcast
        dpp_fence();
        static_for<N>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            // C[r] -= g_r ye: the broadcast operand is the multiplier here
            fmacb_neg<r>(C[r].re, gi.re, ye.re);
            fmacb<r>(C[r].re, gi.im, ye.im);
            fmacb_neg<r>(C[r].im, gi.re, ye.im);
            fmacb_neg<r>(C[r].im, gi.im, ye.re);
        });
        // row src of W_hat^H = (w / sqrt(d))^H, i.e. W_hat[f][i][src] = w_i / sqrt(d)
        if (live && i < M) store_what<double>(a, ((size_t)f * M + i) * M + src, wi.re * sc, wi.im * sc);
        R16_STAMP(src * 4 + 3);
    }
}

}  // namespace

// the determined float64 update of 9..16 channels with one matrix row per lane; false: not this kernel's case
bool update_det16r_applies(const UpdateArgs& a) {
    static const bool on = [] { const char* v = getenv("OIVA_DET16_ROWS"); return !(v && v[0] == '0'); }();
    return on && a.K == a.M && a.M > 8 && a.M <= 16 && a.use_double && !a.init_only && a.layout == 0 && a.nsplit <= kMaxSplitsR;
}

hipError_t launch_update_det16r(hipStream_t s, const UpdateArgs& a) {
    const dim3 grid((a.F + kBinsPerWaveR - 1) / kBinsPerWaveR);
    if (a.vpart_f64)
        hipLaunchKernelGGL(update_det16r_kernel<double>, grid, dim3(192), 0, s, a);
    else
        hipLaunchKernelGGL(update_det16r_kernel<float>, grid, dim3(192), 0, s, a);
    return hipGetLastError();
}

#ifdef OIVA_R16_TRACE
extern "C" int oiva_debug_r16_trace(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_r16_trace), sizeof(g_r16_trace));
}
extern "C" int oiva_debug_r16_hwid(unsigned* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_r16_hwid), sizeof(g_r16_hwid));
}
#endif

}  // namespace oiva
