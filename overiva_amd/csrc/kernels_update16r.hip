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
// The partial covariances of source s + 1 ([split][bin][source][M * M] float64, packed Hermitian) are moved into LDS by
// DMA (global_load_lds, no registers) while source s is solved.
#include "oiva_device.h"

#include <cstdint>
#include <cstdlib>
#include <type_traits>

namespace oiva {
namespace {

constexpr int N = 16;
constexpr int kBinsPerWaveR = 4;
constexpr int kMaxSplitsR = 4;      // frame splits staged in LDS (more: the one-matrix-per-wave kernel)

struct Z {
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
__device__ __forceinline__ Z zmul(Z a, Z b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ Z zinv_fast(Z a) {
    const double d = rcp_nr(a.re * a.re + a.im * a.im);
    return {a.re * d, -a.im * d};
}
__device__ __forceinline__ Z zinv(Z a) {
    const double d = 1.0 / (a.re * a.re + a.im * a.im);
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

template <typename VT>
struct LdsR {
    VT stage[kMaxSplitsR][kBinsPerWaveR][N * N];       // partial covariances of one source, as the covariance kernel stored them
    double vsum[kBinsPerWaveR * N * N + 2];             // their sum over the splits / T (packed Hermitian blocks of 256; + 2: the neighbour read of the last diagonal)
    Z sq[kBinsPerWaveR][N][N];                          // C-phase: the inverse's rows, filed under the column they pivoted
    Z prow[kBinsPerWaveR][N];                           // C-phase: pivot row
    Z ppiv[kBinsPerWaveR];                              //          pivot element
    Z ucol[kBinsPerWaveR][N];                           // u = C e_s by row index
    Z ys[kBinsPerWaveR];                                // y_s
};

template <typename VT>
__global__ __launch_bounds__(64) void update_det16r_kernel(UpdateArgs a) {
    __shared__ LdsR<VT> s;
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
    const int M = a.M, NA = M * M;
    const int fraw = blockIdx.x * kBinsPerWaveR + g;
    const bool live = fraw < a.F;                          // (the last wave of a bin count that is no multiple of 4)
    const int f = live ? fraw : a.F - 1;
    const Z zero = {0., 0.};

    // ---- DMA of the partials of one source into s.stage: blocks of NA values per (split, bin), 16 bytes per lane and request
    const VT* vbase = static_cast<const VT*>(a.Vpart);
    const size_t vstride = (size_t)a.F * M * NA;
    const int nsplit = a.nsplit;
    const unsigned blk_bytes = (unsigned)NA * (unsigned)sizeof(VT);
    const int npiece = (int)((blk_bytes + 1023u) / 1024u);                 // requests per block (<= 2)
    auto stage_source = [&](int src) {
        for (int sp = 0; sp < nsplit; ++sp) {
#pragma unroll
            for (int gg = 0; gg < kBinsPerWaveR; ++gg) {
                const int fr = blockIdx.x * kBinsPerWaveR + gg;
                const int fg = fr < a.F ? fr : a.F - 1;
                const char* blk = reinterpret_cast<const char*>(vbase + (size_t)sp * vstride + ((size_t)fg * M + src) * NA);
                for (int pc = 0; pc < npiece; ++pc) {
                    unsigned off = (unsigned)(pc * 64 + lane) * 16u;
                    off = off + 16u <= blk_bytes ? off : 0u;               // lanes past the block re-read its start (their LDS words are not used)
                    __builtin_amdgcn_global_load_lds((gvoid_t*)(blk + off), (lvoid_t*)(reinterpret_cast<char*>(&s.stage[sp][gg][0]) + pc * 1024), 16, 0, 0);
                }
            }
        }
    };
    stage_source(0);

    // ---- where row i of a packed Hermitian block lies: byte offset of (re, im) of entry (i, c), sign of the imaginary part
    int voff[N];
#pragma unroll
    for (int c = 0; c < N; ++c) {
        const int lo = i < c ? i : c, hi = i < c ? c : i;
        voff[c] = (i == c || hi >= M) ? (i < M ? i : 0) : herm_pair_index(M, lo, hi);
    }

    // ---- B = W_hat^H (identity outside M x M), rows scaled by 1 / wscale (overiva.py:163 / :167); lane (g, i): row i
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
        auto step = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const Z aic = A[c];
            const float mag = (float)(aic.re * aic.re + aic.im * aic.im);
            unsigned key = used ? 0u : ((__float_as_uint(mag) & ~31u) | 16u | (unsigned)(15 - i));
            unsigned o;
            o = dppu<0x121>(key); key = o > key ? o : key;     // row_ror:1, 2, 4, 8: the maximum over the 16 lanes of the row
            o = dppu<0x122>(key); key = o > key ? o : key;
            o = dppu<0x124>(key); key = o > key ? o : key;
            o = dppu<0x128>(key); key = o > key ? o : key;
            const int p = 15 - (int)(key & 15u);
            const bool mine = i == p;
            wave_lds_sync();
            if (mine) {
#pragma unroll
                for (int j = 0; j < N; ++j) s.prow[g][j] = j == c ? Z{1., 0.} : A[j];
                s.ppiv[g] = aic;
            }
            wave_lds_sync();
            const Z apc = s.ppiv[g];
            used = used || mine;
            mycol = mine ? c : mycol;
            piv.re = mine ? apc.re : piv.re;
            piv.im = mine ? apc.im : piv.im;
            Z fct = zmul(aic, zinv_fast(apc));
            fct.re = mine ? 0. : fct.re;
            fct.im = mine ? 0. : fct.im;
            A[c] = {mine ? 1. : 0., 0.};
#pragma unroll
            for (int j = 0; j < N; ++j) zsubmul(A[j], fct, s.prow[g][j]);
        };
#define OIVA_R16_STEP(c) \
    if (c < M) step(std::integral_constant<int, c>{});
        OIVA_R16_STEP(0) OIVA_R16_STEP(1) OIVA_R16_STEP(2) OIVA_R16_STEP(3) OIVA_R16_STEP(4) OIVA_R16_STEP(5) OIVA_R16_STEP(6)
        OIVA_R16_STEP(7) OIVA_R16_STEP(8) OIVA_R16_STEP(9) OIVA_R16_STEP(10) OIVA_R16_STEP(11) OIVA_R16_STEP(12)
        OIVA_R16_STEP(13) OIVA_R16_STEP(14) OIVA_R16_STEP(15)
#undef OIVA_R16_STEP
        wave_lds_sync();
        {
            const Z ip = zinv(piv);
#pragma unroll
            for (int j = 0; j < N; ++j) s.sq[g][mycol][j] = zmul(A[j], ip);
        }
        wave_lds_sync();
        // lane j takes column j of the inverse: C[r][j] = sq[r][q_j]
#pragma unroll
        for (int r = 0; r < N; ++r) C[r] = s.sq[g][r][mycol];
        wave_lds_sync();
    }

    // ---- A x = rhs for a Hermitian positive definite A (identity outside M x M), one row per lane: Gauss-Jordan without pivot
    //      search on [A | rhs], columns <= k skipped; the pivot row by row_newbcast.  Returns x_i.
    auto solve_hpd = [&](Z (&V)[N], Z rhs) -> Z {
        double dmine = 1.;
        auto step = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const double pk = bc<k>(V[k].re);              // A[k][k] (real: Schur complements stay Hermitian)
            const double d = rcp_nr(pk);
            const bool rowk = i == k;
            dmine = rowk ? d : dmine;
            const Z m = {rowk ? 0. : V[k].re * d, rowk ? 0. : V[k].im * d};      // the pivot row eliminates with factor 0
            static_for<N - 1 - k>([&](auto jc) {
                constexpr int c = k + 1 + decltype(jc)::value;
                zsubmul_b<k>(V[c], m, V[c]);
            });
            zsubmul_b<k>(rhs, m, rhs);
        };
#define OIVA_R16_STEP(c) \
    if (c < M) step(std::integral_constant<int, c>{});
        OIVA_R16_STEP(0) OIVA_R16_STEP(1) OIVA_R16_STEP(2) OIVA_R16_STEP(3) OIVA_R16_STEP(4) OIVA_R16_STEP(5) OIVA_R16_STEP(6)
        OIVA_R16_STEP(7) OIVA_R16_STEP(8) OIVA_R16_STEP(9) OIVA_R16_STEP(10) OIVA_R16_STEP(11) OIVA_R16_STEP(12)
        OIVA_R16_STEP(13) OIVA_R16_STEP(14) OIVA_R16_STEP(15)
#undef OIVA_R16_STEP
        return {rhs.re * dmine, rhs.im * dmine};
    };

    const double invT = 1.0 / (double)a.T;
    for (int src = 0; src < M; ++src) {
        // u = C e_src: the 16 registers of lane src, to the rows through LDS
        wave_lds_sync();
        if (i == src) {
#pragma unroll
            for (int r = 0; r < N; ++r) s.ucol[g][r] = C[r];
        }
        // the staged partials of this source have arrived (the only vector-memory traffic in flight besides stores).  Their sum
        // over the splits, in order, times 1 / T -- in the layout they arrived in: lane l adds the 16-byte pieces l + 64 j of the
        // four bins' blocks -- goes to s.vsum, and the DMA of the next source is requested at once.
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
        wave_lds_sync();
        const Z ui = s.ucol[g][i];
        {
            constexpr int kPieces = kBinsPerWaveR * N * N / 2 / 64;        // 16-byte pieces per lane and split (8)
            const double2* st = reinterpret_cast<const double2*>(&s.stage[0][0][0]);
            double2 acc[kPieces];
#pragma unroll
            for (int j = 0; j < kPieces; ++j) acc[j] = st[lane + 64 * j];
#pragma unroll
            for (int sp = 1; sp < kMaxSplitsR; ++sp)
                if (sp < nsplit) {                          // (uniform)
#pragma unroll
                    for (int j = 0; j < kPieces; ++j) {
                        const double2 v = st[sp * (kBinsPerWaveR * N * N / 2) + lane + 64 * j];
                        acc[j].x += v.x;
                        acc[j].y += v.y;
                    }
                }
            double2* vs = reinterpret_cast<double2*>(&s.vsum[0]);
#pragma unroll
            for (int j = 0; j < kPieces; ++j) vs[lane + 64 * j] = make_double2(acc[j].x * invT, acc[j].y * invT);
        }
        wave_lds_sync();
        if (src + 1 < M) stage_source(src + 1);
        // V_s, row i: entry (i, c) of the packed Hermitian block, the imaginary part negated below the diagonal
        Z V[N];
#pragma unroll
        for (int c = 0; c < N; ++c) {
            V[c] = {i == c ? 1. : 0., 0.};
            if (c < M) {                                    // (uniform)
                const double vr = s.vsum[g * (N * N) + voff[c]], vi = s.vsum[g * (N * N) + voff[c] + 1];
                if (i < M) {
                    V[c].re = vr;
                    V[c].im = i == c ? 0. : (i < c ? vi : -vi);
                }
            }
        }
        dpp_fence();
        // w = V^-1 u (not yet normalised)
        const Z wi = solve_hpd(V, ui);
        // y = w^H C: lane c forms column c, w_r by broadcast
        Z y = zero;
        dpp_fence();
        static_for<N>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            fmacb<r>(y.re, wi.re, C[r].re);
            fmacb<r>(y.re, wi.im, C[r].im);
            fmacb<r>(y.im, wi.re, C[r].im);
            fmacb_neg<r>(y.im, wi.im, C[r].re);
        });
        // y_src = w^H u = w^H V w =: d (overiva.py:185; real for the exact w): the normalisation takes its real part, the
        // Sherman-Morrison step divides by the COMPLEX value the rounded w gives (see update_det_kernel)
        wave_lds_sync();
        if (i == src) s.ys[g] = y;
        wave_lds_sync();
        const Z ys = s.ys[g];
        const double d = ys.re;
        const double sc = 1.0 / sqrt(d);
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
    }
}

}  // namespace

// the determined float64 update of 9..16 channels with one matrix row per lane; false: not this kernel's case
bool update_det16r_applies(const UpdateArgs& a) {
    static const bool on = [] { const char* v = getenv("OIVA_DET16_ROWS"); return !(v && v[0] == '0'); }();
    return on && a.K == a.M && a.M > 8 && a.M <= 16 && a.use_double && !a.init_only && a.layout == 0 && a.vpart_f64 && a.nsplit <= kMaxSplitsR;
}

hipError_t launch_update_det16r(hipStream_t s, const UpdateArgs& a) {
    const dim3 grid((a.F + kBinsPerWaveR - 1) / kBinsPerWaveR);
    if (!a.vpart_f64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(update_det16r_kernel<double>, grid, dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace oiva
