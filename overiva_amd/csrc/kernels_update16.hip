// Per-bin sequential algebra for 9..16 channels                            reference overiva.py:176-190
//
// Same mathematics as update_sq_kernel (kernels_update.hip) -- per source: A = W_hat^H V_s, Gauss-Jordan with
// partial pivoting for A w = e_s, w /= sqrt(w^H V w), J from the orthogonality constraint -- with ONE WAVEFRONT per
// bin and four matrix elements per lane (a 16 x 16 matrix has 256).
#include "oiva_device.h"

#include <cstdlib>
#include <type_traits>

namespace oiva {
namespace {

constexpr int N = 16;

template <typename R>
struct C2 {
    R re, im;
};
template <typename R>
__device__ __forceinline__ C2<R> cmul(C2<R> a, C2<R> b) {
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
template <typename R>
__device__ __forceinline__ C2<R> cconj(C2<R> a) {
    return {a.re, -a.im};
}
template <typename R>
__device__ __forceinline__ C2<R> cinv(C2<R> a) {
    const R d = R(1) / (a.re * a.re + a.im * a.im);
    return {a.re * d, -a.im * d};
}

__device__ __forceinline__ void herm_off(int M, int i, int j, int& off, float& sgn) {
    if (i == j) {
        off = i;
        sgn = 0.f;
    } else if (i < j) {
        off = herm_pair_index(M, i, j);
        sgn = 1.f;
    } else {
        off = herm_pair_index(M, j, i);
        sgn = -1.f;
    }
}

// Lane l = (i, q) = (l >> 2, l & 3) owns row i of the 16 x 16 matrices, columns q, q + 4, q + 8, q + 12 (element e is
// column 4e + q).  A row's 16 entries sit in one quad, so a column entry reaches its row by a quad-permute DPP; the
// pivot is an arg-max over a packed (magnitude, row) key by two row rotations and two lane exchanges; the pivot row is
// published through LDS (same-address reads broadcast).  Nothing crosses a wavefront: no barrier waits on another wave,
// and Gauss-Jordan skips the columns that are already eliminated (known at compile time in the unrolled loop).
// (Round 1 used a workgroup of 256 lanes per bin, one element per lane, with rows / columns / pivots through LDS and two
// barriers per elimination step: 516 us at 2048 x 16 / 16 against 142 us here, 366 -> 113 us at 8 sources, 97 -> 39 us at 2.)

// entry of quad lane QL to the whole quad
template <int QL, typename R>
__device__ __forceinline__ C2<R> quad_bcast(C2<R> v) {
    return {dpp<QL * 0x55>(v.re), dpp<QL * 0x55>(v.im)};
}
// sum over the four lanes of a quad, on every lane
template <typename R>
__device__ __forceinline__ R quad_sum(R v) {
    v += dpp<kDppXor1>(v);
    v += dpp<kDppXor2>(v);
    return v;
}
template <typename R>
__device__ __forceinline__ R wave_sum16(R v) {           // v is already equal within quads: sum over the 16 rows
    v += dpp<kDppRor4>(v);
    v += dpp<kDppRor8>(v);
    return swapsum32(swapsum16(v));
}

// 1 / a from the hardware reciprocal: 1 ulp in single precision (the elimination factors of the float32 mode do not
// need the correctly rounded quotient); in double the ~24-bit seed takes two Newton steps to the last bits.  A correctly
// rounded division is a dozen (float) to thirty (double) instructions on the critical path of every pivot.
__device__ __forceinline__ C2<float> cinv_fast(C2<float> a) {
    const float d = __builtin_amdgcn_rcpf(a.re * a.re + a.im * a.im);
    return {a.re * d, -a.im * d};
}
__device__ __forceinline__ C2<double> cinv_fast(C2<double> a) {
    const double n = a.re * a.re + a.im * a.im;
    double d = __builtin_amdgcn_rcp(n);
    d = fma(fma(-n, d, 1.0), d, d);
    d = fma(fma(-n, d, 1.0), d, d);
    return {a.re * d, -a.im * d};
}

template <typename R>
struct LdsDet {
    C2<R> V[N][N];        // V_s, row m, columns permuted so that a lane's four are contiguous: slot q * 4 + e
    C2<R> prow[4][4];     // pivot row, [q][e]
    C2<R> ppiv, prhs;     // pivot element and right-hand side of the pivot row
    C2<R> w[N];           // solution by column index
    R pk[N * N];          // split-summed packed Hermitian block of the current source
};

// the workgroup is one wavefront and LDS operations of a wave complete in order: waiting for the wave's own LDS
// traffic and pinning the instruction order is all the synchronisation a write -> read exchange needs
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
}

template <typename R, typename VT, bool OVER>   // OVER: K < M, background rows and the orthogonality constraint
__global__ __launch_bounds__(64) void update_wave16_kernel(UpdateArgs a) {
    __shared__ LdsDet<R> s;
    const int lane = threadIdx.x, i = lane >> 2, q = lane & 3;
    const int f = blockIdx.x, M = a.M, K = a.K, NA = M * M;
    const C2<R> zero = {R(0), R(0)};
    // (a run-time test in the K < M instantiation: with the branch known taken the compiler's schedule needs 290-450
    // registers and spills; kept uniform and 'unknown' it needs 173-268)
    const bool over = OVER && K < M;
    // B = W_hat^H, identity outside M x M; C = Cx (only needed for the orthogonality constraint, K < M)
    C2<R> B[4], C[4];
    int off[4];
    float sgn[4];
    bool in[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * e + q;
        in[e] = i < M && c < M;
        B[e] = {R(i == c ? 1 : 0), R(0)};
        C[e] = zero;
        off[e] = 0;
        sgn[e] = 0.f;
        if (in[e]) {
            R vr, vi;
            load_what<R>(a, ((size_t)f * M + c) * M + i, vr, vi);
            B[e] = {vr, -vi};
            herm_off(M, i, c, off[e], sgn[e]);
            if (over) {
                const double* pc = a.Cx + (size_t)f * NA + off[e];
                C[e].re = R(pc[0]);
                if (sgn[e] != 0.f) C[e].im = R(sgn[e] * pc[1]);
            }
        }
    }
    if (a.wscale != nullptr && i < K) {                   // overiva.py:163 / :167
        const R sc = R(1) / R(a.wscale[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) B[e] = {B[e].re * sc, B[e].im * sc};
    }
    // out[i][4e + q] = sum_m B[i][m] S[m][4e + q] with S in LDS (s.V, a lane's four columns contiguous)
    auto times_lds = [&](C2<R> (&out)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = zero;
#pragma unroll
        for (int me = 0; me < 4; ++me) {
            auto term = [&](C2<R> b, int m) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const C2<R> v = s.V[m][q * 4 + e];
                    out[e].re += b.re * v.re - b.im * v.im;
                    out[e].im += b.re * v.im + b.im * v.re;
                }
            };
            term(quad_bcast<0>(B[me]), 4 * me + 0);
            term(quad_bcast<1>(B[me]), 4 * me + 1);
            term(quad_bcast<2>(B[me]), 4 * me + 2);
            term(quad_bcast<3>(B[me]), 4 * me + 3);
        }
    };
    auto to_lds = [&](const C2<R> (&m)[4]) {
        wave_lds_sync();
#pragma unroll
        for (int e = 0; e < 4; ++e) s.V[i][q * 4 + e] = m[e];
        wave_lds_sync();
    };
    C2<R> Tm[4] = {zero, zero, zero, zero};               // rows < K: W^H Cx
    if (over) {
        to_lds(C);
        times_lds(Tm);
    }
    // one Gauss-Jordan step with partial pivoting on column c of [A | rhs]; rows with `used` are never chosen
    auto step = [&](auto cc, C2<R> (&A)[4], C2<R>& rhs, bool& used, C2<R>& piv, int& mycol) {
        constexpr int c = decltype(cc)::value;
        constexpr int ce = c >> 2, cq = c & 3;
        constexpr int e0 = (c + 1) >> 2;                   // elements below e0 hold only eliminated columns
        const C2<R> aic = quad_bcast<cq>(A[ce]);
        const float mag = (float)(aic.re * aic.re + aic.im * aic.im);
        unsigned key = used ? 0u : ((__float_as_uint(mag) & ~31u) | 16u | (unsigned)(15 - i));
        unsigned o;
        o = dpp<kDppRor4>(key); key = o > key ? o : key;
        o = dpp<kDppRor8>(key); key = o > key ? o : key;
        key = swapmax32(swapmax16(key));
        const int p = 15 - (int)(__builtin_amdgcn_readfirstlane((int)key) & 15);
        wave_lds_sync();
        if (i == p) {
#pragma unroll
            for (int e = e0; e < 4; ++e) s.prow[q][e] = A[e];
            if (q == 0) {
                s.ppiv = aic;
                s.prhs = rhs;
            }
        }
        wave_lds_sync();
        const C2<R> apc = s.ppiv, bp = s.prhs;
        const bool mine = i == p;
        used = used || mine;
        mycol = mine ? c : mycol;
        piv.re = mine ? apc.re : piv.re;                   // (component-wise: a select of the pair goes through scratch)
        piv.im = mine ? apc.im : piv.im;
        C2<R> fct = cmul(aic, cinv_fast(apc));
        fct.re = mine ? R(0) : fct.re;                     // the pivot row eliminates with factor 0
        fct.im = mine ? R(0) : fct.im;
#pragma unroll
        for (int e = e0; e < 4; ++e) {
            const C2<R> r = s.prow[q][e];
            A[e].re -= fct.re * r.re - fct.im * r.im;
            A[e].im -= fct.re * r.im + fct.im * r.re;
        }
        rhs.re -= fct.re * bp.re - fct.im * bp.im;
        rhs.im -= fct.re * bp.im + fct.im * bp.re;
    };
    // columns 0 .. npiv-1 (npiv uniform)
    auto gauss_jordan = [&](int npiv, C2<R> (&A)[4], C2<R>& rhs, bool& used, C2<R>& piv, int& mycol) {
#define OIVA_GJ_STEP(c) \
    if (c < npiv) step(std::integral_constant<int, c>{}, A, rhs, used, piv, mycol);
        OIVA_GJ_STEP(0) OIVA_GJ_STEP(1) OIVA_GJ_STEP(2) OIVA_GJ_STEP(3) OIVA_GJ_STEP(4) OIVA_GJ_STEP(5) OIVA_GJ_STEP(6)
        OIVA_GJ_STEP(7) OIVA_GJ_STEP(8) OIVA_GJ_STEP(9) OIVA_GJ_STEP(10) OIVA_GJ_STEP(11) OIVA_GJ_STEP(12)
        OIVA_GJ_STEP(13) OIVA_GJ_STEP(14) OIVA_GJ_STEP(15)
#undef OIVA_GJ_STEP
    };
    const R invT = R(1) / R(a.T);
    // V_s arrives as the packed Hermitian block of every frame split ([split][bin][source][M * M]): lane l fetches
    // values 4l .. 4l + 3 of each block (16-byte loads when M is even), the splits are added in order in float64, the
    // sum goes to LDS and every lane gathers its four entries.  The loads of source s + 1 are in flight while source s
    // is solved.
    // splits held in registers across a solve.  (float64 partials + float64 algebra + background rows: 8 of them made 255
    // registers and ONE wave per SIMD -- two rounds of bins at 2048 bins; with 4 the kernel fits two waves per SIMD)
    constexpr int kAhead = (OVER && sizeof(VT) == 8 && sizeof(R) == 8) ? 4 : 8;
    const VT* vbase = static_cast<const VT*>(a.Vpart);
    const size_t vstride = (size_t)a.F * K * NA;
    const int nahead = a.nsplit < kAhead ? a.nsplit : kAhead;
    auto fetch = [&](int src, int sp, VT (&raw)[4]) {
        const VT* p = vbase + (size_t)sp * vstride + ((size_t)f * K + src) * NA + lane * 4;
        if ((M & 1) == 0) {                                   // block start and length are multiples of 4 values
            if (lane * 4 < NA) {
                if constexpr (sizeof(VT) == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(p);
                    raw[0] = v.x, raw[1] = v.y, raw[2] = v.z, raw[3] = v.w;
                } else {
                    const double2 v0 = reinterpret_cast<const double2*>(p)[0], v1 = reinterpret_cast<const double2*>(p)[1];
                    raw[0] = v0.x, raw[1] = v0.y, raw[2] = v1.x, raw[3] = v1.y;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (lane * 4 + r < NA) raw[r] = p[r];
        }
    };
    VT raw[kAhead][4] = {};
    auto fetch_ahead = [&](int src) {
#pragma unroll
        for (int sp = 0; sp < kAhead; ++sp)
            if (sp < nahead) fetch(src, sp, raw[sp]);
    };
    const int nsrc = a.init_only ? 0 : K;
    if (nsrc > 0) fetch_ahead(0);
    for (int src = 0; src <= nsrc; ++src) {
        const bool solve = src < nsrc;
        if (!solve && !a.init_only) break;
        C2<R> wi = zero;
        if (solve) {
            double acc[4] = {0., 0., 0., 0.};
#pragma unroll
            for (int sp = 0; sp < kAhead; ++sp)
                if (sp < nahead) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] += (double)raw[sp][r];
                }
            for (int sp = kAhead; sp < a.nsplit; ++sp) {       // very long frame axes only
                VT more[4] = {VT(0), VT(0), VT(0), VT(0)};
                fetch(src, sp, more);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] += (double)more[r];
            }
            if (src + 1 < nsrc) fetch_ahead(src + 1);
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < 4; ++r) s.pk[lane * 4 + r] = R(acc[r]);
            wave_lds_sync();
            C2<R> V[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                V[e] = zero;
                if (in[e]) {
                    V[e].re = s.pk[off[e]] * invT;
                    if (sgn[e] != 0.f) V[e].im = s.pk[off[e] + 1] * R(sgn[e]) * invT;
                }
            }
            to_lds(V);
            C2<R> A[4];
            times_lds(A);                                      // A = W_hat^H V
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (!in[e]) A[e] = {R(i == 4 * e + q ? 1 : 0), R(0)};
            // Gauss-Jordan with partial pivoting on [A | e_src]
            C2<R> rhs = {R(i == src ? 1 : 0), R(0)};
            bool used = false;
            C2<R> piv = {R(1), R(0)};
            int mycol = i;
            gauss_jordan(N, A, rhs, used, piv, mycol);
            // w[c] = rhs / pivot on the row that pivoted column c
            wave_lds_sync();
            if (q == 0) s.w[mycol] = cmul(rhs, cinv(piv));
            wave_lds_sync();
            wi = s.w[i];
            C2<R> wc[4];
            // d = w^H V w (real, positive): row i of V w from this lane's four columns, then over the quad and the rows
            R tr = R(0), ti = R(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                wc[e] = s.w[4 * e + q];
                tr += V[e].re * wc[e].re - V[e].im * wc[e].im;
                ti += V[e].re * wc[e].im + V[e].im * wc[e].re;
            }
            tr = quad_sum(tr);
            ti = quad_sum(ti);
            const R d = wave_sum16(wi.re * tr + wi.im * ti);
            const R sc = R(1) / sqrt(d);
            wi = {wi.re * sc, wi.im * sc};
            if (i == src) {
#pragma unroll
                for (int e = 0; e < 4; ++e) B[e] = {wc[e].re * sc, -wc[e].im * sc};
            }
        }
        if (over) {                                            // orthogonality constraint, overiva.py:189-190 -> :96-98
            if (solve) {
                // row src of W^H Cx = sum_m conj(w_m) Cx[m][:]: column sums over the 16 rows (lanes of equal q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const C2<R> x = cmul(cconj(wi), C[e]);
                    const R tre = wave_sum16(x.re), tim = wave_sum16(x.im);
                    if (i == src) Tm[e] = {tre, tim};
                }
            }
            // J = (W^H Cx)[:, :K]^-1 (W^H Cx)[:, K:]: eliminate the first K columns on the first K rows
            C2<R> G[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) G[e] = i < K ? Tm[e] : C2<R>{R(i == 4 * e + q ? 1 : 0), R(0)};
            C2<R> dummy = zero;
            bool used = i >= K;
            C2<R> piv = {R(1), R(0)};
            int mycol = i;
            gauss_jordan(K, G, dummy, used, piv, mycol);
            // J[m][j] = G[row that pivoted column m][j] / pivot: the pivot rows publish their normalised rows at index m
            wave_lds_sync();
            if (i < K) {
                const C2<R> ip = cinv(piv);
#pragma unroll
                for (int e = 0; e < 4; ++e) s.V[mycol][q * 4 + e] = cmul(G[e], ip);
            }
            wave_lds_sync();
            // W_hat[m][i] = J[m][i - K]  ->  (W_hat^H)[i][m] = conj, for rows i >= K and columns m < K
            if (i >= K && i < M) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = 4 * e + q;
                    if (m < K) B[e] = cconj(s.V[m][(i & 3) * 4 + (i >> 2)]);
                }
            }
            wave_lds_sync();
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (in[e]) store_what<R>(a, ((size_t)f * M + 4 * e + q) * M + i, B[e].re, -B[e].im);
}

// ---------------------------------------------------------------------------------------------------------------------
// The determined case (K = M, AuxIVA; BASELINE configs[4] is 16 x 16), float64: A w = e_s with A = W_hat^H V_s as
//   w = V_s^-1 u,  u = column s of C = (W_hat^H)^-1
// (the scheme of update_det_kernel, kernels_update.hip, in this kernel's lane layout): C from ONE pivoted elimination on
// [W_hat^H | I] per bin and iteration, updated after every source by the rank-one formula
//   C' = C - (u / d) (y - sqrt(d) e_s^T),   y = w^H C,   d = y_s = w^H u = w^H V_s w   (overiva.py:185)
// and V_s^-1 by an elimination without pivot search (Hermitian positive definite) and without the product W_hat^H V_s.
// ---------------------------------------------------------------------------------------------------------------------
template <typename VT, bool INVERSE>   // INVERSE: w = V^-1 u through the explicit inverse (round 4; $OIVA_DET16_INVERSE=1), else one elimination on [V | u]
__global__ __launch_bounds__(64) void update_det16_kernel(UpdateArgs a) {
    using R = double;
    __shared__ LdsDet<R> s;
    const int lane = threadIdx.x, i = lane >> 2, q = lane & 3;
    const int f = blockIdx.x, M = a.M, NA = M * M;
    C2<R> B[4];
    int off[4];
    float sgn[4];
    bool in[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * e + q;
        in[e] = i < M && c < M;
        B[e] = {R(i == c ? 1 : 0), R(0)};
        off[e] = 0;
        sgn[e] = 0.f;
        if (in[e]) {
            R vr, vi;
            load_what<R>(a, ((size_t)f * M + c) * M + i, vr, vi);
            B[e] = {vr, -vi};
            herm_off(M, i, c, off[e], sgn[e]);
        }
    }
    if (a.wscale != nullptr && i < M) {                   // overiva.py:163 / :167
        const R sc = R(1) / R(a.wscale[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) B[e] = {B[e].re * sc, B[e].im * sc};
    }
    // the partials of source s + 1 are in flight while source s is worked on (as update_wave16_kernel)
    constexpr int kAhead = 8;
    const VT* vbase = static_cast<const VT*>(a.Vpart);
    const size_t vstride = (size_t)a.F * M * NA;
    const int nahead = a.nsplit < kAhead ? a.nsplit : kAhead;
    auto fetch = [&](int src, int sp, VT (&raw)[4]) {
        const VT* p = vbase + (size_t)sp * vstride + ((size_t)f * M + src) * NA + lane * 4;
        if ((M & 1) == 0) {
            if (lane * 4 < NA) {
                if constexpr (sizeof(VT) == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(p);
                    raw[0] = v.x, raw[1] = v.y, raw[2] = v.z, raw[3] = v.w;
                } else {
                    const double2 v0 = reinterpret_cast<const double2*>(p)[0], v1 = reinterpret_cast<const double2*>(p)[1];
                    raw[0] = v0.x, raw[1] = v0.y, raw[2] = v1.x, raw[3] = v1.y;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (lane * 4 + r < NA) raw[r] = p[r];
        }
    };
    VT raw[kAhead][4] = {};
    auto fetch_ahead = [&](int src) {
#pragma unroll
        for (int sp = 0; sp < kAhead; ++sp)
            if (sp < nahead) fetch(src, sp, raw[sp]);
    };
    fetch_ahead(0);

    // ---- C = (W_hat^H)^-1: Gauss-Jordan with partial pivoting on [A | Rm], rows never move; row c of the inverse is the row
    //      that pivoted column c, divided by its pivot
    C2<R> C[4];
    {
        C2<R> A[4], Rm[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            A[e] = B[e];
            Rm[e] = {R(i == 4 * e + q ? 1 : 0), R(0)};
        }
        bool used = false;
        C2<R> piv = {R(1), R(0)};
        int mycol = i;
        auto step = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int ce = c >> 2, cq = c & 3;
            constexpr int e0 = (c + 1) >> 2;               // elements of A below e0 hold only eliminated columns
            const C2<R> aic = quad_bcast<cq>(A[ce]);
            const float mag = (float)(aic.re * aic.re + aic.im * aic.im);
            unsigned key = used ? 0u : ((__float_as_uint(mag) & ~31u) | 16u | (unsigned)(15 - i));
            unsigned o;
            o = dpp<kDppRor4>(key); key = o > key ? o : key;
            o = dpp<kDppRor8>(key); key = o > key ? o : key;
            key = swapmax32(swapmax16(key));
            const int p = 15 - (int)(__builtin_amdgcn_readfirstlane((int)key) & 15);
            wave_lds_sync();
            if (i == p) {
#pragma unroll
                for (int e = e0; e < 4; ++e) s.prow[q][e] = A[e];
#pragma unroll
                for (int e = 0; e < 4; ++e) s.V[0][q * 4 + e] = Rm[e];
                if (q == 0) s.ppiv = aic;
            }
            wave_lds_sync();
            const C2<R> apc = s.ppiv;
            const bool mine = i == p;
            used = used || mine;
            mycol = mine ? c : mycol;
            piv.re = mine ? apc.re : piv.re;
            piv.im = mine ? apc.im : piv.im;
            C2<R> fct = cmul(aic, cinv_fast(apc));
            fct.re = mine ? R(0) : fct.re;
            fct.im = mine ? R(0) : fct.im;
#pragma unroll
            for (int e = e0; e < 4; ++e) {
                const C2<R> r = s.prow[q][e];
                A[e].re -= fct.re * r.re - fct.im * r.im;
                A[e].im -= fct.re * r.im + fct.im * r.re;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const C2<R> r = s.V[0][q * 4 + e];
                Rm[e].re -= fct.re * r.re - fct.im * r.im;
                Rm[e].im -= fct.re * r.im + fct.im * r.re;
            }
        };
#define OIVA_D16_STEP(c) \
    if (c < M) step(std::integral_constant<int, c>{});
        OIVA_D16_STEP(0) OIVA_D16_STEP(1) OIVA_D16_STEP(2) OIVA_D16_STEP(3) OIVA_D16_STEP(4) OIVA_D16_STEP(5) OIVA_D16_STEP(6)
        OIVA_D16_STEP(7) OIVA_D16_STEP(8) OIVA_D16_STEP(9) OIVA_D16_STEP(10) OIVA_D16_STEP(11) OIVA_D16_STEP(12)
        OIVA_D16_STEP(13) OIVA_D16_STEP(14) OIVA_D16_STEP(15)
#undef OIVA_D16_STEP
        wave_lds_sync();
        {
            const C2<R> ip = cinv(piv);
#pragma unroll
            for (int e = 0; e < 4; ++e) s.V[mycol][q * 4 + e] = cmul(Rm[e], ip);
        }
        wave_lds_sync();
#pragma unroll
        for (int e = 0; e < 4; ++e) C[e] = s.V[i][q * 4 + e];
        wave_lds_sync();
    }

    // ---- in-place inverse of a Hermitian positive definite matrix (identity outside M x M): no pivot search
    auto herm_inverse = [&](C2<R> (&A)[4]) {
        auto step = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int ke = k >> 2, kq = k & 3;
            const C2<R> aik = quad_bcast<kq>(A[ke]);       // A[i][k]
            wave_lds_sync();
            if (i == k) {
#pragma unroll
                for (int e = 0; e < 4; ++e) s.prow[q][e] = A[e];
                if (q == 0) s.ppiv = aik;                    // A[k][k]
            }
            wave_lds_sync();
            // (the pivots of a Hermitian positive definite elimination are real: Schur complements stay Hermitian)
            const R pr = s.ppiv.re;
            R d = __builtin_amdgcn_rcp(pr);
            d = fma(fma(-pr, d, 1.0), d, d);
            d = fma(fma(-pr, d, 1.0), d, d);
            const C2<R> ad = {aik.re * d, aik.im * d};
            const bool rowk = i == k;
            // every row: A[i][c] -= f A[k][c] with f = A[i][k] d -- and f = 1 - d on row k itself, which leaves A[k][c] d there;
            // column k (where that gives 0 and 1) is then set to -A[i][k] d, and to d on the diagonal
            const C2<R> fct = {rowk ? R(1) - d : ad.re, rowk ? R(0) : ad.im};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const C2<R> r = s.prow[q][e];
                A[e].re -= fct.re * r.re - fct.im * r.im;
                A[e].im -= fct.re * r.im + fct.im * r.re;
            }
            if (q == kq) {
                A[ke].re = rowk ? d : -ad.re;
                A[ke].im = rowk ? R(0) : -ad.im;
            }
        };
#define OIVA_H16_STEP(c) \
    if (c < M) step(std::integral_constant<int, c>{});
        OIVA_H16_STEP(0) OIVA_H16_STEP(1) OIVA_H16_STEP(2) OIVA_H16_STEP(3) OIVA_H16_STEP(4) OIVA_H16_STEP(5) OIVA_H16_STEP(6)
        OIVA_H16_STEP(7) OIVA_H16_STEP(8) OIVA_H16_STEP(9) OIVA_H16_STEP(10) OIVA_H16_STEP(11) OIVA_H16_STEP(12)
        OIVA_H16_STEP(13) OIVA_H16_STEP(14) OIVA_H16_STEP(15)
#undef OIVA_H16_STEP
    };

    // ---- A x = rhs for a Hermitian positive definite A (identity outside M x M): Gauss-Jordan without pivot search on [A | rhs],
    //      columns that are already eliminated skipped (known at compile time in the unrolled loop) -- 36 element updates per lane
    //      instead of the 64 of the in-place inverse, and no product with the inverse afterwards.  rhs: one value per row (equal
    //      within the quad); returns x the same way.
    auto solve_hpd = [&](C2<R> (&A)[4], C2<R> rhs) -> C2<R> {
        R piv = R(1);
        auto step = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int ke = k >> 2, kq = k & 3;
            constexpr int e0 = (k + 1) >> 2;               // elements below e0 hold only eliminated columns
            const C2<R> aik = quad_bcast<kq>(A[ke]);       // A[i][k]
            wave_lds_sync();
            if (i == k) {
#pragma unroll
                for (int e = e0; e < 4; ++e) s.prow[q][e] = A[e];
                if (q == 0) {
                    s.ppiv = aik;                          // A[k][k] (real: Schur complements stay Hermitian)
                    s.prhs = rhs;
                }
            }
            wave_lds_sync();
            const R pr = s.ppiv.re;
            R d = __builtin_amdgcn_rcp(pr);
            d = fma(fma(-pr, d, 1.0), d, d);
            d = fma(fma(-pr, d, 1.0), d, d);
            const bool rowk = i == k;
            piv = rowk ? pr : piv;
            const C2<R> fct = {rowk ? R(0) : aik.re * d, rowk ? R(0) : aik.im * d};      // the pivot row eliminates with factor 0
#pragma unroll
            for (int e = e0; e < 4; ++e) {
                const C2<R> r = s.prow[q][e];
                A[e].re -= fct.re * r.re - fct.im * r.im;
                A[e].im -= fct.re * r.im + fct.im * r.re;
            }
            const C2<R> bp = s.prhs;
            rhs.re -= fct.re * bp.re - fct.im * bp.im;
            rhs.im -= fct.re * bp.im + fct.im * bp.re;
        };
#define OIVA_S16_STEP(c) \
    if (c < M) step(std::integral_constant<int, c>{});
        OIVA_S16_STEP(0) OIVA_S16_STEP(1) OIVA_S16_STEP(2) OIVA_S16_STEP(3) OIVA_S16_STEP(4) OIVA_S16_STEP(5) OIVA_S16_STEP(6)
        OIVA_S16_STEP(7) OIVA_S16_STEP(8) OIVA_S16_STEP(9) OIVA_S16_STEP(10) OIVA_S16_STEP(11) OIVA_S16_STEP(12)
        OIVA_S16_STEP(13) OIVA_S16_STEP(14) OIVA_S16_STEP(15)
#undef OIVA_S16_STEP
        // x_i = rhs_i / (the pivot of row i, kept when the row pivoted: entries of eliminated columns are not zeroed, so the
        // diagonal itself is stale by now)
        R d = __builtin_amdgcn_rcp(piv);
        d = fma(fma(-piv, d, 1.0), d, d);
        d = fma(fma(-piv, d, 1.0), d, d);
        return {rhs.re * d, rhs.im * d};
    };

    const R invT = R(1) / R(a.T);
    for (int src = 0; src < M; ++src) {
        double acc[4] = {0., 0., 0., 0.};
#pragma unroll
        for (int sp = 0; sp < kAhead; ++sp)
            if (sp < nahead) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] += (double)raw[sp][r];
            }
        for (int sp = kAhead; sp < a.nsplit; ++sp) {           // very long frame axes only
            VT more[4] = {VT(0), VT(0), VT(0), VT(0)};
            fetch(src, sp, more);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += (double)more[r];
        }
        if (src + 1 < M) fetch_ahead(src + 1);
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) s.pk[lane * 4 + r] = R(acc[r]);
        wave_lds_sync();
        C2<R> Vi[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Vi[e] = {R(i == 4 * e + q ? 1 : 0), R(0)};
            if (in[e]) {
                Vi[e] = {s.pk[off[e]] * invT, R(0)};
                if (sgn[e] != 0.f) Vi[e].im = s.pk[off[e] + 1] * R(sgn[e]) * invT;
            }
        }
        // u = column src of C: u_i for the row (through LDS)
        const int se = src >> 2, sq = src & 3;
        wave_lds_sync();
        if (q == sq) {
            C2<R> v = C[0];
            v.re = se == 1 ? C[1].re : (se == 2 ? C[2].re : (se == 3 ? C[3].re : v.re));
            v.im = se == 1 ? C[1].im : (se == 2 ? C[2].im : (se == 3 ? C[3].im : v.im));
            s.w[i] = v;
        }
        wave_lds_sync();
        const C2<R> ui = s.w[i];
        // w = V^-1 u (not yet normalised)
        C2<R> wi;
        if constexpr (INVERSE) {
            herm_inverse(Vi);
            R wr = R(0), wim = R(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const C2<R> uc = s.w[4 * e + q];
                wr += Vi[e].re * uc.re - Vi[e].im * uc.im;
                wim += Vi[e].re * uc.im + Vi[e].im * uc.re;
            }
            wi = {quad_sum(wr), quad_sum(wim)};
        } else {
            wi = solve_hpd(Vi, ui);
        }
        wave_lds_sync();
        if (q == 0) s.w[i] = wi;
        wave_lds_sync();
        // y = w^H C: column sums over the 16 rows (lanes of equal q)
        C2<R> y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const C2<R> x = cmul(cconj(wi), C[e]);
            y[e] = {wave_sum16(x.re), wave_sum16(x.im)};
        }
        // y_src = w^H u = w^H V w =: d (overiva.py:185; real for the exact w): the normalisation takes its real part, the
        // Sherman-Morrison step divides by the COMPLEX value the rounded w gives (see update_det_kernel); held by the lanes
        // q == sq in element se
        wave_lds_sync();
        if (lane == sq) {
            // (component-wise: a select of an (re, im) pair is compiled into a stack array indexed by the lane)
            const R vre = se == 0 ? y[0].re : (se == 1 ? y[1].re : (se == 2 ? y[2].re : y[3].re));
            const R vim = se == 0 ? y[0].im : (se == 1 ? y[1].im : (se == 2 ? y[2].im : y[3].im));
            s.ppiv = {vre, vim};
        }
        wave_lds_sync();
        const C2<R> ys = s.ppiv;
        const R d = ys.re;
        const R sc = R(1) / sqrt(d);
        // Sherman-Morrison for the new row src of W_hat^H (exact for any w): C' = C - (u / y_src) (y - sqrt(d) e_src^T)
        const C2<R> g = cmul(ui, cinv_fast(ys));
        const R sqd = d * sc;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            C2<R> ye = y[e];
            if (4 * e + q == src) ye.re -= sqd;
            C[e].re -= g.re * ye.re - g.im * ye.im;
            C[e].im -= g.re * ye.im + g.im * ye.re;
        }
        // row src of W_hat^H = (w / sqrt(d))^H
        if (i == src) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const C2<R> wc = s.w[4 * e + q];
                B[e] = {wc.re * sc, -wc.im * sc};
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (in[e]) store_what<R>(a, ((size_t)f * M + 4 * e + q) * M + i, B[e].re, -B[e].im);
}

}  // namespace

hipError_t launch_update_wave16(hipStream_t s, const UpdateArgs& a) {
    if (update_det16r_applies(a)) return launch_update_det16r(s, a);       // (round 6; $OIVA_DET16_ROWS=0: the kernels below)
    dim3 grid(a.F);
    auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, dim3(64), 0, s, a); };
    const bool over = a.K < a.M;
    static const bool det = [] { const char* v = getenv("OIVA_UPDATE_DET"); return !(v && v[0] == '0'); }();
    if (det && !over && a.use_double && !a.init_only) {
        static const bool inv = [] { const char* v = getenv("OIVA_DET16_INVERSE"); return v && v[0] == '1'; }();
        if (a.vpart_f64) {
            if (inv) go(update_det16_kernel<double, true>); else go(update_det16_kernel<double, false>);
        } else {
            if (inv) go(update_det16_kernel<float, true>); else go(update_det16_kernel<float, false>);
        }
        return hipGetLastError();
    }
#define OIVA_GO(RR, VV)                                         \
    if (over) go(update_wave16_kernel<RR, VV, true>);           \
    else go(update_wave16_kernel<RR, VV, false>);
    if (a.use_double) {
        if (a.vpart_f64) { OIVA_GO(double, double) } else { OIVA_GO(double, float) }
    } else {
        if (a.vpart_f64) { OIVA_GO(float, double) } else { OIVA_GO(float, float) }
    }
#undef OIVA_GO
    return hipGetLastError();
}

}  // namespace oiva
