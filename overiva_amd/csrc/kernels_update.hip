// Per-bin sequential algebra of one iteration                       reference overiva.py:176-190
//
//   for s in 0..K-1:
//       V_s   = (1/T) * sum of the covariance partials                       (:179, reduction tail)
//       A     = W_hat^H V_s ;  w_s = A^{-1} e_s                              (:181-182, IP1)
//       w_s  /= sqrt(w_s^H V_s w_s)                                          (:185-186)
//       J     = (W^H Cx)[:, :K]^{-1} (W^H Cx)[:, K:]   when K < M            (:189-190 -> :96-98)
//   plus, on entry, the pending scale normalisation W[:, k] /= wscale[k]     (:163 / :167)
//   and, with init_only, just the J initialisation of the prologue           (:120-123)
//
// One bin is handled by a group of SG lanes (SG = next power of two >= M); lane i holds ROW i of the
// matrices: row i of W_hat^H (i.e. conj of column i of W_hat), of Cx, of V_s, of A.  Rows are
// exchanged with sub-group shuffles; Gauss-Jordan elimination with partial pivoting never moves a
// row (the pivot lane just broadcasts).  Every register array is indexed with compile-time
// constants only (loops over columns are fully unrolled; run-time M and K enter as predicates).
#include <cstdlib>

#include "oiva_device.h"
#include "update_chain.h"

namespace oiva {
namespace {

// Gauss-Jordan elimination on rows held one per lane.  Pivots columns 0..npiv-1 (npiv uniform);
// rows with used == true are never chosen.  On return: perm[c] = lane that pivoted column c,
// piv = the pivot element of this lane's own row (if it pivoted), every non-pivot row has a zero in
// each pivoted column, rhs transformed alongside.
template <int SG, typename R>
__device__ __forceinline__ void gauss_jordan(Cx<R> (&A)[SG], Cx<R>& rhs, int npiv, bool used, int (&perm)[SG],
                                             Cx<R>& piv, int i) {
#pragma unroll
    for (int c = 0; c < SG; ++c) {
        perm[c] = c;
        if (c < npiv) {
            R mag = used ? R(-1) : (A[c].re * A[c].re + A[c].im * A[c].im);
            int bl = i;
#pragma unroll
            for (int off = SG / 2; off > 0; off >>= 1) {
                const R om = __shfl_xor(mag, off, SG);
                const int ol = __shfl_xor(bl, off, SG);
                const bool take = (om > mag) || (om == mag && ol < bl);
                mag = take ? om : mag;
                bl = take ? ol : bl;
            }
            const int p = bl;
            perm[c] = p;
            const bool isp = (i == p);
            const Cx<R> pc = gshfl<SG>(A[c], p);
            const Cx<R> fct = cmul(A[c], cinv(pc));
            if (isp) {
                used = true;
                piv = A[c];
            }
            const Cx<R> pb = gshfl<SG>(rhs, p);
            if (!isp) cfms(rhs, fct, pb);
#pragma unroll
            for (int j = c + 1; j < SG; ++j) {
                const Cx<R> pj = gshfl<SG>(A[j], p);
                if (!isp) cfms(A[j], fct, pj);
            }
            if (!isp) A[c] = {R(0), R(0)};
        }
    }
}

template <int SG, typename R, int MT, int KT>
__global__ __launch_bounds__(kBlock) void update_kernel(UpdateArgs a) {
    const int tid = threadIdx.x;
    const int i = tid % SG;
    const int grp = tid / SG;
    const int f_raw = blockIdx.x * (kBlock / SG) + grp;
    const bool fvalid = f_raw < a.F;
    const int f = fvalid ? f_raw : a.F - 1;
    const int M = MT ? MT : a.M, K = KT ? KT : a.K;   // compile-time constants fold the run-time predicates
    const int NA = M * M;
    const bool row_ok = i < M;
    const Cx<R> zero = {R(0), R(0)};

    // B = row i of W_hat^H: B[m] = conj(W_hat[f][m][i]); padded rows/cols are identity
    Cx<R> B[SG], C[SG], Tm[SG];
#pragma unroll
    for (int m = 0; m < SG; ++m) {
        B[m] = {R(m == i ? 1 : 0), R(0)};
        if (m < M && row_ok) {
            R vr, vi;
            load_what<R>(a, ((size_t)f * M + m) * M + i, vr, vi);
            B[m] = {vr, -vi};
        }
    }
    if (a.wscale != nullptr && i < K) {  // overiva.py:163 / :167
        const R s = R(1) / R(a.wscale[i]);
#pragma unroll
        for (int m = 0; m < SG; ++m) {
            B[m].re *= s;
            B[m].im *= s;
        }
    }
    // C = row i of Cx
#pragma unroll
    for (int j = 0; j < SG; ++j) {
        C[j] = zero;
        if (j < M && row_ok) {
            int off;
            float sgn;
            herm_offsets(M, i, j, off, sgn);
            const double* p = a.Cx + (size_t)f * NA + off;
            C[j].re = R(p[0]);
            if (sgn != 0.f) C[j].im = R(sgn * p[1]);
        }
    }
    // Tm = row i of W^H Cx for i < K (kept across the source loop; only row s changes per source)
#pragma unroll
    for (int j = 0; j < SG; ++j) Tm[j] = zero;
    if (K < M) {
#pragma unroll
        for (int m = 0; m < SG; ++m) {
            if (m < M) {
#pragma unroll
                for (int j = 0; j < SG; ++j) {
                    const Cx<R> cj = gshfl<SG>(C[j], m);
                    cfma(Tm[j], B[m], cj);
                }
            }
        }
    }

    const int nsrc = a.init_only ? 0 : K;
    const R invT = R(1) / R(a.T);
    for (int s = 0; s <= nsrc; ++s) {
        const bool solve = s < nsrc;       // the last trip (s == nsrc) exists only for init_only's J update
        if (!solve && !a.init_only) break;
        Cx<R> w[SG];
        Cx<R> own = zero;
        if (solve) {
            // V row i: fixed-order fp64 sum of the frame-split partials
            Cx<R> Vr[SG];
#pragma unroll
            for (int j = 0; j < SG; ++j) {
                Vr[j] = zero;
                if (j < M && row_ok) {
                    int off;
                    float sgn;
                    herm_offsets(M, i, j, off, sgn);
                    double sr, si;
                    sum_vpart(a.Vpart, a.vpart_f64, ((size_t)f * K + s) * NA + off, (size_t)a.F * K * NA, a.nsplit, sgn != 0.f,
                              sr, si);
                    Vr[j].re = R(sr) * invT;
                    Vr[j].im = R(si) * R(sgn) * invT;
                }
            }
            // A = W_hat^H V
            Cx<R> A[SG];
#pragma unroll
            for (int j = 0; j < SG; ++j) A[j] = zero;
#pragma unroll
            for (int m = 0; m < SG; ++m) {
                if (m < M) {
#pragma unroll
                    for (int j = 0; j < SG; ++j) {
                        const Cx<R> vj = gshfl<SG>(Vr[j], m);
                        cfma(A[j], B[m], vj);
                    }
                }
            }
            if (!row_ok) {
#pragma unroll
                for (int j = 0; j < SG; ++j) A[j] = {R(j == i ? 1 : 0), R(0)};
            }
            Cx<R> rhs = {R(i == s ? 1 : 0), R(0)};
            int perm[SG];
            Cx<R> piv = {R(1), R(0)};
            gauss_jordan<SG, R>(A, rhs, SG, false, perm, piv, i);
            // w[c] lives on lane perm[c] as rhs / pivot; gather the whole vector on every lane
            const Cx<R> q = cmul(rhs, cinv(piv));
#pragma unroll
            for (int c = 0; c < SG; ++c) {
                w[c] = gshfl<SG>(q, perm[c]);
                if (c == i) own = w[c];
            }
            // normalise by sqrt(w^H V w) (real and positive for Hermitian PSD V)
            Cx<R> u = zero;
#pragma unroll
            for (int j = 0; j < SG; ++j) cfma(u, Vr[j], w[j]);
            const R d = gsum<SG, R>(own.re * u.re + own.im * u.im);
            const R sc = R(1) / sqrt(d);
#pragma unroll
            for (int c = 0; c < SG; ++c) {
                w[c].re *= sc;
                w[c].im *= sc;
            }
            own.re *= sc;
            own.im *= sc;
            if (i == s) {
#pragma unroll
                for (int m = 0; m < SG; ++m) B[m] = {w[m].re, -w[m].im};
            }
        }
        if (K < M) {
            if (solve) {
                // row s of W^H Cx: sum over lanes m of conj(w_m) * Cx[m][:]
                Cx<R> t[SG];
                const Cx<R> oc = {own.re, -own.im};
#pragma unroll
                for (int j = 0; j < SG; ++j) {
                    t[j] = cmul(oc, C[j]);
                    t[j].re = gsum<SG, R>(t[j].re);
                    t[j].im = gsum<SG, R>(t[j].im);
                }
                if (i == s) {
#pragma unroll
                    for (int j = 0; j < SG; ++j) Tm[j] = t[j];
                }
            }
            // J: eliminate columns 0..K-1 among rows 0..K-1 of [Tm]; J[c][j-K] = G[j]/pivot on lane perm[c]
            Cx<R> G[SG];
#pragma unroll
            for (int j = 0; j < SG; ++j) G[j] = (i < K) ? Tm[j] : Cx<R>{R(j == i ? 1 : 0), R(0)};
            Cx<R> dummy = zero;
            int perm[SG];
            Cx<R> piv = {R(1), R(0)};
            gauss_jordan<SG, R>(G, dummy, K, i >= K, perm, piv, i);
            const Cx<R> ip = cinv(piv);
#pragma unroll
            for (int m = 0; m < SG; ++m) {
                if (m < K) {
#pragma unroll
                    for (int j = 0; j < SG; ++j) {
                        if (j >= K && j < M) {
                            const Cx<R> val = gshfl<SG>(cmul(G[j], ip), perm[m]);
                            // W_hat[m][j] = J[m][j-K]  ->  row j of W_hat^H, entry m = conj
                            if (i == j) B[m] = {val.re, -val.im};
                        }
                    }
                }
            }
        }
    }

    if (fvalid && row_ok) {
#pragma unroll
        for (int m = 0; m < SG; ++m) {
            if (m < M) store_what<R>(a, ((size_t)f * M + m) * M + i, B[m].re, -B[m].im);
        }
    }
}

// MT / KT: channel and source counts as compile-time constants (0 = run-time value from the arguments).
// With MT == MP and a fixed KT every "m < M" / "c < npiv" predicate folds away, which removes about a third
// of the issued instructions of this latency-bound kernel.
template <int MP, typename R, int MT, int KT>
__global__ __launch_bounds__(kBlock) void update_sq_kernel(UpdateArgs a) {
    constexpr int G = MP * MP;
    const int tid = threadIdx.x;
    const Sq<MP, R> sq(tid % G);
    const int i = sq.i, j = sq.j;
    const int f_raw = blockIdx.x * (kBlock / G) + tid / G;
    const bool fvalid = f_raw < a.F;
    const int f = fvalid ? f_raw : a.F - 1;
    const int M = MT ? MT : a.M, K = KT ? KT : a.K;
    const int NA = M * M;
    const bool in = i < M && j < M;
    const Cx<R> zero = {R(0), R(0)};
    const Cx<R> eye = {R(i == j ? 1 : 0), R(0)};

    // B[i][j] = (W_hat^H)[i][j] = conj(W_hat[j][i]); identity outside M x M
    Cx<R> B = eye;
    if (in) {
        R vr, vi;
        load_what<R>(a, ((size_t)f * M + j) * M + i, vr, vi);
        B = {vr, -vi};
    }
    if (a.wscale != nullptr && i < K) {  // overiva.py:163 / :167
        const R s = R(1) / R(a.wscale[i]);
        B.re *= s;
        B.im *= s;
    }
    int off = 0;
    float sgn = 0.f;
    if (in) herm_offsets(M, i, j, off, sgn);
    Cx<R> C = zero;
    if (in) {
        const double* p = a.Cx + (size_t)f * NA + off;
        C.re = R(p[0]);
        if (sgn != 0.f) C.im = R(sgn * p[1]);
    }
    // Tm = rows of W^H Cx (rows >= K unused)
    Cx<R> Tm = zero;
    if (K < M) Tm = sq.matmul(B, C, M);

    const int nsrc = a.init_only ? 0 : K;
    const R invT = R(1) / R(a.T);
    // V_s[i][j] = (1/T) * fixed-order fp64 sum of the frame-split partials (reduction tail of overiva.py:179)
    auto load_v = [&](int s) {
        Cx<R> V = zero;
        if (in) {
            double sr, si;
            sum_vpart(a.Vpart, a.vpart_f64, ((size_t)f * K + s) * NA + off, (size_t)a.F * K * NA, a.nsplit, sgn != 0.f, sr, si);
            V.re = R(sr) * invT;
            V.im = R(si) * R(sgn) * invT;
        }
        return V;
    };
    Cx<R> Vnext = zero;
    if (nsrc > 0) Vnext = load_v(0);
    for (int s = 0; s <= nsrc; ++s) {
        const bool solve = s < nsrc;
        if (!solve && !a.init_only) break;
        Cx<R> wi = zero, wj = zero;
        if (solve) {
            // V_s was fetched while the previous source was being solved; start the fetch of V_{s+1}
            const Cx<R> V = Vnext;
            if (s + 1 < nsrc) Vnext = load_v(s + 1);
            Cx<R> A = sq.matmul(B, V, M);  // W_hat^H V
            if (!in) A = eye;
            Cx<R> rhs = {R(i == s ? 1 : 0), R(0)};
            int perm[MP];
            Cx<R> piv = {R(1), R(0)};
            sq.gauss_jordan(A, rhs, MP, false, perm, piv);
            const Cx<R> q = cmul(rhs, cinv(piv));  // = w[c] on the row that pivoted column c
#pragma unroll
            for (int c = 0; c < MP; ++c) {
                const Cx<R> wc = sq.colb(q, perm[c]);
                if (i == c) wi = wc;
                if (j == c) wj = wc;
            }
            // d = w^H V w
            const Cx<R> vw = cmul(V, wj);
            const R d = sq.allsum(wi.re * vw.re + wi.im * vw.im);
            const R sc = fast_rsqrt(d);
            wi.re *= sc;
            wi.im *= sc;
            wj.re *= sc;
            wj.im *= sc;
            if (i == s) B = {wj.re, -wj.im};
        }
        if (K < M) {
            if (solve) {
                // row s of W^H Cx = sum_m conj(w_m) Cx[m][:]
                Cx<R> t = cmul(Cx<R>{wi.re, -wi.im}, C);
                t.re = sq.colsum(t.re);
                t.im = sq.colsum(t.im);
                if (i == s) Tm = t;
            }
            Cx<R> Gm = (i < K) ? Tm : eye;
            Cx<R> dummy = zero;
            int perm[MP];
            Cx<R> piv = {R(1), R(0)};
            sq.gauss_jordan(Gm, dummy, K, i >= K, perm, piv);
            const Cx<R> Jn = cmul(Gm, cinv(piv));  // on row perm[m]: J[m][j-K] for j >= K
#pragma unroll
            for (int m = 0; m < MP; ++m) {
                if (m < K) {
                    const Cx<R> row = sq.colb(Jn, perm[m]);   // lane (i, j): J[m][j-K]
                    const Cx<R> tr = sq.transp(row);          // lane (i, j): J[m][i-K]
                    // W_hat[m][i] = J[m][i-K]  ->  (W_hat^H)[i][m] = conj
                    if (j == m && i >= K && i < M) B = {tr.re, -tr.im};
                }
            }
        }
    }
    if (fvalid && in) store_what<R>(a, ((size_t)f * M + j) * M + i, B.re, -B.im);
}

// ---------------------------------------------------------------------------------------------
// Structured form of the same update for 1 or 2 sources with background channels (K < M): the chain that
// depends on W is a K x K solve and a few matrix-vector products instead of an M x M elimination with pivoting.
//   W_hat^H = [[W^H], [J^H | -I]],  A = W_hat^H V,  A w = e_s   <=>   W_hat^H u = e_s,  u = V w:
//     rows >= K :  u_bot = J^H u_top
//     rows <  K :  Q u_top = e_s   with   Q = B_tt + B_tb B_bt   (B = W_hat^H in K | M-K blocks)      K x K
//     w = V^-1 u                   V^-1 of the Hermitian positive definite V_s needs no pivoting and does not
//                                  depend on W: all K inverses are formed before the chain starts
//   w /= sqrt(w^H V w)  (overiva.py:185-186);  J from (W^H Cx)[:, :K]^-1 (W^H Cx)[:, K:] in closed form (:96-98).
// Same lane layout as update_sq_kernel (lane (i, j) = element [i][j], one wavefront per bin at 8 channels).
// Requires the -I block of W_hat (overiva.py:122-123), which every path of the library maintains.
// ---------------------------------------------------------------------------------------------
template <int MP, typename R, int MT, int KT>
__global__ __launch_bounds__(kBlock) void update_bg_kernel(UpdateArgs a) {
    static_assert(KT == 1 || KT == 2, "closed-form K x K solves");
    constexpr int G = MP * MP;
    constexpr int K = KT;
    const int tid = threadIdx.x;
    const Sq<MP, R> sq(tid % G);
    const int i = sq.i, j = sq.j;
    const int f_raw = blockIdx.x * (kBlock / G) + tid / G;
    const bool fvalid = f_raw < a.F;
    const int f = fvalid ? f_raw : a.F - 1;
    const int M = MT ? MT : a.M;
    const int NA = M * M;
    const bool in = i < M && j < M;
    const Cx<R> zero = {R(0), R(0)};
    const Cx<R> eye = {R(i == j ? 1 : 0), R(0)};

    Cx<R> B = eye;                        // B[i][j] = (W_hat^H)[i][j] = conj(W_hat[j][i]); identity outside M x M
    if (in) {
        R vr, vi;
        load_what<R>(a, ((size_t)f * M + j) * M + i, vr, vi);
        B = {vr, -vi};
    }
    if (a.wscale != nullptr && i < K) {  // overiva.py:163 / :167
        const R sc = R(1) / R(a.wscale[i]);
        B.re *= sc;
        B.im *= sc;
    }
    int off = 0;
    float sgn = 0.f;
    if (in) herm_offsets(M, i, j, off, sgn);
    Cx<R> C = zero;
    if (in) {
        const double* p = a.Cx + (size_t)f * NA + off;
        C.re = R(p[0]);
        if (sgn != 0.f) C.im = R(sgn * p[1]);
    }
    // V_s and V_s^-1 for all sources (off the chain); V_s[i][j] = (1/T) * fixed-order fp64 sum of the partials
    const R invT = R(1) / R(a.T);
    Cx<R> V[K];
#pragma unroll
    for (int s = 0; s < K; ++s) {
        V[s] = eye;
        if (in) {
            double sr, si;
            sum_vpart(a.Vpart, a.vpart_f64, ((size_t)f * K + s) * NA + off, (size_t)a.F * K * NA, a.nsplit, sgn != 0.f, sr, si);
            V[s] = {R(sr) * invT, R(si) * R(sgn) * invT};
        }
    }
    bg_chain<MP, R, K>(sq, B, C, V, M);
    if (fvalid && in) store_what<R>(a, ((size_t)f * M + j) * M + i, B.re, -B.im);
}

// ---------------------------------------------------------------------------------------------
// The determined case (K = M, AuxIVA): A w = e_s with A = W_hat^H V_s  <=>  w = V_s^-1 u,  u = column s of C = (W_hat^H)^-1.
//   * V_s^-1: Hermitian positive definite, no pivot search, independent of W;
//   * C: ONE pivoted elimination per bin and iteration; when source s replaces row s of W_hat^H by w'^H the inverse follows by
//     the rank-one formula  C' = C - u (y - e_s^T) / y_s,  y = w'^H C  (Sherman-Morrison; y_s = sqrt(w^H V w) > 0);
//   * w^H V_s w = u^H V_s^-1 u = u^H w: the normalisation (overiva.py:185-186) costs a dot product.
// Per source two matrix-vector products, a dot product and a rank-one update instead of an M x M elimination with pivoting
// (2049 x 235, float64, update stage: 8 / 8 55.5 -> 37.5 us, 7 / 7 47.8 -> 31.1, 6 / 6 40.6 -> 25.2, 5 / 5 33.8 -> 19.9, 4 / 4 10.5 -> 8.9,
// 3 / 3 8.7 -> 6.9).  Same lane layout as update_sq_kernel.
// ---------------------------------------------------------------------------------------------
template <int MP, typename R, int MT>
__global__ __launch_bounds__(kBlock) void update_det_kernel(UpdateArgs a) {
    constexpr int G = MP * MP;
    const int tid = threadIdx.x;
    const Sq<MP, R> sq(tid % G);
    const int i = sq.i, j = sq.j;
    const int f_raw = blockIdx.x * (kBlock / G) + tid / G;
    const bool fvalid = f_raw < a.F;
    const int f = fvalid ? f_raw : a.F - 1;
    const int M = MT ? MT : a.M;
    const int NA = M * M;
    const bool in = i < M && j < M;
    const Cx<R> eye = {R(i == j ? 1 : 0), R(0)};

    Cx<R> B = eye;                        // B[i][j] = (W_hat^H)[i][j] = conj(W_hat[j][i]); identity outside M x M
    if (in) {
        R vr, vi;
        load_what<R>(a, ((size_t)f * M + j) * M + i, vr, vi);
        B = {vr, -vi};
    }
    if (a.wscale != nullptr && i < M) {  // overiva.py:163 / :167
        const R sc = R(1) / R(a.wscale[i]);
        B.re *= sc;
        B.im *= sc;
    }
    int off = 0;
    float sgn = 0.f;
    if (in) herm_offsets(M, i, j, off, sgn);
    const R invT = R(1) / R(a.T);
    auto load_v = [&](int s) {
        Cx<R> V = eye;
        if (in) {
            double sr, si;
            sum_vpart(a.Vpart, a.vpart_f64, ((size_t)f * M + s) * NA + off, (size_t)a.F * M * NA, a.nsplit, sgn != 0.f, sr, si);
            V = {R(sr) * invT, R(si) * R(sgn) * invT};
        }
        return V;
    };
    // (all M inverses V_s^-1 first, their eliminations interleaved step by step, measured no faster -- 8 / 8 float64 40.8 against
    //  38.9 us, float32 33.6 against 29.8 -- the kernel is bound by the instructions it issues, two waves per SIMD, not by the
    //  latency of one chain)
    Cx<R> Vnext = load_v(0);
    Cx<R> C = sq.inverse_pivoted(B, M);
    static_for<MP>([&](auto sc_) {
        constexpr int s = decltype(sc_)::value;
        if (s < M) {
            const Cx<R> V = Vnext;
            if (s + 1 < M) Vnext = load_v(s + 1);
            const Cx<R> Vinv = sq.herm_inverse(V, M);
            const Cx<R> ui = sq.template rowb_c<s>(C);                 // u_i = C[i][s]
            const Cx<R> uj = sq.at(C, j, s);                           // u_j
            const Cx<R> wi = sq.rowsum(cmul(Vinv, uj));                // w = V^-1 u (not yet normalised), one entry per row
            // y = w^H C, one entry per column;  y_s = w^H u = w^H V w =: d  (overiva.py:185) -- real for the exact w.  The
            // normalisation takes its real part; the Sherman-Morrison step below must divide by the COMPLEX y_s the rounded
            // w really gives: round 4 divided by Re(y_s) there, and on a W_hat not yet adapted to V_s (first iterations) with
            // cond(V) = 1e10 that put the result 2e-4 from the reference's, whose own sensitivity there is 8e-7; with the
            // complex denominator the form stays at that sensitivity up to cond 1e12 (tests/test_update_forms.py; d from
            // V itself on top of it, or a step of iterative refinement, changed nothing and is not done)
            const Cx<R> t = cmul(Cx<R>{wi.re, -wi.im}, C);
            Cx<R> y = {sq.colsum(t.re), sq.colsum(t.im)};
            const Cx<R> ys = sq.template rowb_c<s>(y);
            const R d = ys.re;
            const R sc = fast_rsqrt(d);
            // row s of W_hat^H becomes w'^H, w' = w / sqrt(d): by Sherman-Morrison (exact for ANY w)
            //   C' = C - u (w'^H C - e_s^T) / (w'^H u) = C - (u / y_s) (y - sqrt(d) e_s^T)
            const Cx<R> g = cmul(ui, cinv(ys));
            if (j == s) y.re -= d * sc;
            cfms(C, g, y);
            // (off the chain) row s of W_hat^H = w'^H
            const Cx<R> wj = sq.transp(wi);
            if (i == s && j < M) B = {wj.re * sc, -wj.im * sc};
        }
    });
    if (fvalid && in) store_what<R>(a, ((size_t)f * M + j) * M + i, B.re, -B.im);
}

// ---------------------------------------------------------------------------------------------
// Sources with background channels (K < M), any K: the column the IP1 solve needs has a closed form that never touches J.
// With W_hat^H = [[W^H], [J^H | -I]] and J from the orthogonality constraint (overiva.py:96-98, J = (W^H Cx)[:, :K]^-1
// (W^H Cx)[:, K:]):
//     column s of (W_hat^H)^-1  =  P G^-1 e_s,     P = Cx W  (M x K),   G = W^H Cx W = W^H P  (K x K, Hermitian positive definite)
// (block inverse: the first K columns are [Q^-1; J^H Q^-1] with Q = W^H [I; J^H] = G (W^H Cx)[:, :K]^-H), so per source
//     c = P G^-1 e_s,   w = V_s^-1 c,   w /= sqrt(c^H w)   (c^H w = w^H V_s w, overiva.py:185),   column s of P and row / column s of G
// follow w -- two eliminations WITHOUT pivot search (M x M and K x K), three matrix-vector products, one product with w^H --
// and J is formed ONCE, after the last source, from the final W (the reference forms it after every source, :189-190, but uses
// only the last).  Float64.  Same lane layout as update_sq_kernel.
// ---------------------------------------------------------------------------------------------
template <int MP, int MT, int KT>
__global__ __launch_bounds__(kBlock) void update_gram_kernel(UpdateArgs a) {
    using R = double;
    constexpr int G_ = MP * MP;
    const int tid = threadIdx.x;
    const Sq<MP, R> sq(tid % G_);
    const int i = sq.i, j = sq.j;
    const int f_raw = blockIdx.x * (kBlock / G_) + tid / G_;
    const bool fvalid = f_raw < a.F;
    const int f = fvalid ? f_raw : a.F - 1;
    const int M = MT ? MT : a.M, K = KT ? KT : a.K;
    const int NA = M * M;
    const bool in = i < M && j < M;
    const Cx<R> zero = {R(0), R(0)};
    const Cx<R> eye = {R(i == j ? 1 : 0), R(0)};

    Cx<R> B = eye;                        // B[i][j] = (W_hat^H)[i][j] = conj(W_hat[j][i]); identity outside M x M
    if (in) {
        R vr, vi;
        load_what<R>(a, ((size_t)f * M + j) * M + i, vr, vi);
        B = {vr, -vi};
    }
    if (a.wscale != nullptr && i < K) {  // overiva.py:163 / :167
        const R sc = R(1) / R(a.wscale[i]);
        B.re *= sc;
        B.im *= sc;
    }
    int off = 0;
    float sgn = 0.f;
    if (in) herm_offsets(M, i, j, off, sgn);
    Cx<R> C = zero;                       // Cx
    if (in) {
        const double* p = a.Cx + (size_t)f * NA + off;
        C.re = R(p[0]);
        if (sgn != 0.f) C.im = R(sgn * p[1]);
    }
    const R invT = R(1) / R(a.T);
    auto load_v = [&](int s) {
        Cx<R> V = eye;
        if (in) {
            double sr, si;
            sum_vpart(a.Vpart, a.vpart_f64, ((size_t)f * K + s) * NA + off, (size_t)a.F * K * NA, a.nsplit, sgn != 0.f, sr, si);
            V = {R(sr) * invT, R(si) * R(sgn) * invT};
        }
        return V;
    };
    Cx<R> Vnext = load_v(0);
    // Tm = W^H Cx on rows < K (Tm[k][j]);  P = Tm^H (P[i][k] on columns < K);  G = W^H P on (k, l) < K, identity outside
    Cx<R> Tm = sq.matmul(B, C, M);
    if (i >= K) Tm = zero;
    Cx<R> P = sq.transp(Tm);
    P = {P.re, -P.im};
    // G[k][l] = sum_i conj(W[i][k]) P[i][l] = sum_m B[k][m] P[m][l]
    Cx<R> Gm = sq.matmul(B, P, M);
    if (i >= K || j >= K) Gm = eye;
    static_for<MP>([&](auto sc_) {
        constexpr int s = decltype(sc_)::value;
        if (s < K) {
            const Cx<R> V = Vnext;
            if (s + 1 < K) Vnext = load_v(s + 1);
            const Cx<R> Vinv = sq.herm_inverse(V, M);
            const Cx<R> Ginv = sq.herm_inverse(Gm, K);
            const Cx<R> gj = sq.at(Ginv, j, s);                        // g_j = (G^-1)[j][s]   (0 for j >= K: identity there)
            const Cx<R> ci = sq.rowsum(cmul(P, j < K ? gj : zero));    // c = P g, one entry per row
            const Cx<R> cj = sq.transp(ci);
            Cx<R> wi = sq.rowsum(cmul(Vinv, j < M ? cj : zero));       // w = V^-1 c
            const R d = sq.allsum(j == 0 && i < M ? wi.re * ci.re + wi.im * ci.im : R(0));     // c^H w = w^H V w
            const R sc = fast_rsqrt(d);
            wi = {i < M ? wi.re * sc : R(0), i < M ? wi.im * sc : R(0)};
            const Cx<R> wj = sq.transp(wi);
            if (i == s && j < M) B = {wj.re, -wj.im};
            // column s of P = Cx w;  row s of G = w^H P, column s its conjugate
            const Cx<R> pi = sq.rowsum(cmul(C, wj));
            if (j == s) P = pi;
            const Cx<R> t = cmul(Cx<R>{wi.re, -wi.im}, P);
            const Cx<R> y = {sq.colsum(t.re), sq.colsum(t.im)};        // y_l = w^H P[:, l]
            const Cx<R> yt = sq.transp(y);                             // lane (k, s): y_k
            if (i == s && j < K) Gm = (j == s) ? Cx<R>{y.re, R(0)} : y;
            if (j == s && i < K && i != s) Gm = {yt.re, -yt.im};
        }
    });
    // J = (W^H Cx)[:, :K]^-1 (W^H Cx)[:, K:] from the final W (overiva.py:189-190 -> :96-98): Tm = P^H on rows < K
    {
        const Cx<R> Pt = sq.transp(P);
        Cx<R> Gj = (i < K) ? Cx<R>{Pt.re, -Pt.im} : eye;
        Cx<R> dummy = zero;
        int perm[MP];
        Cx<R> piv = {R(1), R(0)};
        sq.gauss_jordan(Gj, dummy, K, i >= K, perm, piv);
        const Cx<R> Jn = cmul(Gj, cinv(piv));  // on row perm[m]: J[m][j-K] for j >= K
#pragma unroll
        for (int m = 0; m < MP; ++m) {
            if (m < K) {
                const Cx<R> row = sq.colb(Jn, perm[m]);   // lane (i, j): J[m][j-K]
                const Cx<R> tr = sq.transp(row);          // lane (i, j): J[m][i-K]
                if (j == m && i >= K && i < M) B = {tr.re, -tr.im};
            }
        }
    }
    if (fvalid && in) store_what<R>(a, ((size_t)f * M + j) * M + i, B.re, -B.im);
}

template <int MP, int MT, int KT>
hipError_t launch_gram_one(hipStream_t s, const UpdateArgs& a) {
    const int bins_per_block = kBlock / (MP * MP);
    dim3 grid((a.F + bins_per_block - 1) / bins_per_block);
    hipLaunchKernelGGL((update_gram_kernel<MP, MT, KT>), grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

template <int MP, int MT>
hipError_t launch_det_one(hipStream_t s, const UpdateArgs& a) {
    const int bins_per_block = kBlock / (MP * MP);
    dim3 grid((a.F + bins_per_block - 1) / bins_per_block);
    hipLaunchKernelGGL((update_det_kernel<MP, double, MT>), grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

template <int MP, int MT, int KT>
hipError_t launch_sq_one(hipStream_t s, const UpdateArgs& a) {
    const int bins_per_block = kBlock / (MP * MP);
    dim3 grid((a.F + bins_per_block - 1) / bins_per_block);
    if (a.use_double)
        hipLaunchKernelGGL((update_sq_kernel<MP, double, MT, KT>), grid, dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL((update_sq_kernel<MP, float, MT, KT>), grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

// specialised instantiations for the common shapes (full power-of-two channel count with 2 sources or
// determined), generic otherwise
template <int MP, int MT, int KT>
hipError_t launch_bg_one(hipStream_t s, const UpdateArgs& a) {
    const int bins_per_block = kBlock / (MP * MP);
    dim3 grid((a.F + bins_per_block - 1) / bins_per_block);
    if (a.use_double)
        hipLaunchKernelGGL((update_bg_kernel<MP, double, MT, KT>), grid, dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL((update_bg_kernel<MP, float, MT, KT>), grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

// channel count MT (MP / 2 < MT <= MP) as a compile-time constant, and with it the source counts the reference's own
// sweeps use (overiva_sim_config.json: 2..8 microphones, 1..4 targets, determined AuxIVA): 1, 2, 3, 4, MT
template <int MP, int MT>
hipError_t launch_sq_m(hipStream_t s, const UpdateArgs& a) {
    // 1 or 2 sources with background channels: the structured chain with closed-form K x K solves; 3 and more (float64): the
    // Gram form (update_gram_kernel).  2049 x 235, float64, update stage, generic kernel -> this dispatch: 8 / 3 23.0 -> 21.9 us,
    // 6 / 3 20.2 -> 16.6, 5 / 3 18.7 -> 15.4, 8 / 4 39.5 -> 28.4, 6 / 4 37.8 -> 22.2, 8 / 6 64.2 -> 37.3; with one or two sources the
    // Gram form is on a par with the structured chain (8 / 2 15.4 against 13.7 us, 6 / 2 11.2 / 11.9, 8 / 1 8.6 / 7.7), which
    // the X-resident kernel shares.  (The J initialisation of the prologue keeps the generic kernel.)
    if constexpr (MP >= 4) {
        if (!a.init_only && a.K < MT) {
            static const int gram = [] { const char* v = getenv("OIVA_UPDATE_GRAM"); return v ? atoi(v) : 1; }();
            if (gram == 2 && a.use_double) {      // (measurement only: every K through the Gram form)
                if (a.K == 1) return launch_gram_one<MP, MT, 1>(s, a);
                if (a.K == 2) return launch_gram_one<MP, MT, 2>(s, a);
            }
            if (a.K == 2) return launch_bg_one<MP, MT, 2>(s, a);
            if (a.K == 1) return launch_bg_one<MP, MT, 1>(s, a);
            if (gram && a.use_double) {
                if constexpr (MT > 3) if (a.K == 3) return launch_gram_one<MP, MT, 3>(s, a);
                if constexpr (MT > 4) if (a.K == 4) return launch_gram_one<MP, MT, 4>(s, a);
                if constexpr (MT > 5) return launch_gram_one<MP, MT, 0>(s, a);
            }
        }
    }
    if (a.K == MT) {
        static const bool det = [] { const char* v = getenv("OIVA_UPDATE_DET"); return !(v && v[0] == '0'); }();
        // (float64 only: in float32 the explicit inverses cost accuracy -- 8 / 8 mixture, 20 iterations: 6.5 reference floors
        //  against 2.8 with the elimination per source -- and `fast` keeps the latter)
        if (det && a.use_double && !a.init_only && MT >= 2) return launch_det_one<MP, MT>(s, a);
        return launch_sq_one<MP, MT, MT>(s, a);
    }
    if constexpr (MT > 1) if (a.K == 1) return launch_sq_one<MP, MT, 1>(s, a);
    if constexpr (MT > 2) if (a.K == 2) return launch_sq_one<MP, MT, 2>(s, a);
    if constexpr (MT > 3) if (a.K == 3) return launch_sq_one<MP, MT, 3>(s, a);
    if constexpr (MT > 4) if (a.K == 4) return launch_sq_one<MP, MT, 4>(s, a);
    return launch_sq_one<MP, MT, 0>(s, a);
}

template <int MP>
hipError_t launch_sq(hipStream_t s, const UpdateArgs& a) {
    if (a.M == MP) return launch_sq_m<MP, MP>(s, a);
    if constexpr (MP >= 4) if (a.M == MP - 1) return launch_sq_m<MP, MP - 1>(s, a);
    if constexpr (MP >= 8) {
        if (a.M == MP - 2) return launch_sq_m<MP, MP - 2>(s, a);
        if (a.M == MP - 3) return launch_sq_m<MP, MP - 3>(s, a);
    }
    return launch_sq_one<MP, 0, 0>(s, a);
}

template <int SG, int MT, int KT>
hipError_t launch_sg_one(hipStream_t s, const UpdateArgs& a) {
    const int bins_per_block = kBlock / SG;
    dim3 grid((a.F + bins_per_block - 1) / bins_per_block);
    if (a.use_double)
        hipLaunchKernelGGL((update_kernel<SG, double, MT, KT>), grid, dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL((update_kernel<SG, float, MT, KT>), grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

template <int SG>
hipError_t launch_sg(hipStream_t s, const UpdateArgs& a) {
    // run-time M and K only: folding them as constants made this variant slower (the fully unrolled source
    // loop spills: 16 channels / 16 sources 0.89 -> 1.67 ms)
    return launch_sg_one<SG, 0, 0>(s, a);
}

}  // namespace

hipError_t launch_update(hipStream_t s, const UpdateArgs& a) {
    if (a.layout == 0) {  // square layout: one lane per matrix element
        if (a.M <= 2) return launch_sq<2>(s, a);
        if (a.M <= 4) return launch_sq<4>(s, a);
        if (a.M <= 8) return launch_sq<8>(s, a);
    }
    if (a.layout == 0 && a.M > 8 && a.M <= 16) return launch_update_wave16(s, a);   // one wavefront per bin
    if (a.M <= 2) return launch_sg<2>(s, a);
    if (a.M <= 4) return launch_sg<4>(s, a);
    if (a.M <= 8) return launch_sg<8>(s, a);
    if (a.M <= 16) return launch_sg<16>(s, a);
    return hipErrorInvalidValue;
}

}  // namespace oiva
