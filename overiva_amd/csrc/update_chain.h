// Per-bin algebra shared by the stand-alone update kernels (kernels_update.hip) and the X-resident iteration kernel
// (kernels_resident.hip): complex helpers, the square lane layout (one lane per matrix element) and the structured
// IP1 + orthogonal-constraint chain for 1 or 2 sources with background channels (reference overiva.py:181-190).
#pragma once
#include "oiva_device.h"

namespace oiva {

template <typename R>
struct Cx {
    R re, im;
};
template <typename R>
__device__ __forceinline__ Cx<R> cmul(Cx<R> a, Cx<R> b) {
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
template <typename R>
__device__ __forceinline__ void cfma(Cx<R>& acc, Cx<R> a, Cx<R> b) {  // acc += a*b
    acc.re += a.re * b.re - a.im * b.im;
    acc.im += a.re * b.im + a.im * b.re;
}
template <typename R>
__device__ __forceinline__ void cfms(Cx<R>& acc, Cx<R> a, Cx<R> b) {  // acc -= a*b
    acc.re -= a.re * b.re - a.im * b.im;
    acc.im -= a.re * b.im + a.im * b.re;
}
// reciprocal / reciprocal square root for float: hardware approximation (1 ulp) plus one Newton step
// (~0.5 ulp).  Each sits on the critical path of an elimination step, where the IEEE-exact division
// sequence costs ~12 dependent instructions against 3 here.
__device__ __forceinline__ float fast_rcp(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(r, fmaf(-x, r, 1.f), r);
}
// double: the hardware seed (v_rcp_f64 / v_rsq_f64, > 26 bits) and two Newton steps -- the last bits of a double (the
// IEEE division costs about thirty dependent instructions, 1 / sqrt about sixty, on the critical path of every step)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ float fast_rsqrt(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    return fmaf(0.5f * r, fmaf(-x * r, r, 1.f), r);
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = fma(0.5 * r, fma(-x * r, r, 1.0), r);
    return fma(0.5 * r, fma(-x * r, r, 1.0), r);
}

template <typename R>
__device__ __forceinline__ Cx<R> cinv(Cx<R> a) {
    const R d = fast_rcp(a.re * a.re + a.im * a.im);
    return {a.re * d, -a.im * d};
}
template <int SG, typename R>
__device__ __forceinline__ R gshfl(R v, int src) {
    return __shfl(v, src, SG);
}
template <int SG, typename R>
__device__ __forceinline__ Cx<R> gshfl(Cx<R> v, int src) {
    return {__shfl(v.re, src, SG), __shfl(v.im, src, SG)};
}
template <int SG, typename R>
__device__ __forceinline__ R gsum(R v) {
#pragma unroll
    for (int off = SG / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, SG);
    return v;
}

// entry (i, j) of a packed Hermitian matrix (see herm_pair_index), as a complex number
__device__ __forceinline__ void herm_offsets(int M, int i, int j, int& off, float& sgn) {
    if (i == j) {
        off = i;
        sgn = 0.f;  // no imaginary part
    } else if (i < j) {
        off = herm_pair_index(M, i, j);
        sgn = 1.f;
    } else {
        off = herm_pair_index(M, j, i);
        sgn = -1.f;
    }
}

// ---------------------------------------------------------------------------------------------
// Square layout for M <= 8: one bin = MP x MP lanes (MP = next power of two >= M), lane (i, j) holds
// element [i][j] of every matrix.  For M = 8 that is exactly one wavefront per bin, so 2048 bins give
// 2048 independent waves and each elimination step is O(1) arithmetic per lane: the sequential chain
// per source is ~MP pivots x (a handful of cross-lane moves + one complex divide + one complex FMA).
// ---------------------------------------------------------------------------------------------
template <int MP, typename R>
struct Sq {
    static constexpr int G = MP * MP;
    int i, j, gl;
    __device__ __forceinline__ Sq(int lane_in_group) : i(lane_in_group / MP), j(lane_in_group % MP), gl(lane_in_group) {}
    // value held by lane (i, c): same row, column c
    __device__ __forceinline__ R rowb(R v, int c) const { return __shfl(v, i * MP + c, G); }
    __device__ __forceinline__ Cx<R> rowb(Cx<R> v, int c) const { return {rowb(v.re, c), rowb(v.im, c)}; }
    // the same with the column known at compile time: register-level (DPP) instead of the LDS crossbar -- a quad-permute
    // broadcast inside every quad, and for 8 columns a half-row mirror that carries the right quad's value into the
    // other one (written only to those banks).  2 VALU moves against a ds_bpermute round trip on a latency-bound chain.
    template <int C>
    __device__ __forceinline__ R rowb_c(R v) const {
        if constexpr (MP == 8) {
            const R t = dpp<(C & 3) * 0x55>(v);
            return dpp_banks<kDppHalfMirror, (C < 4 ? 0xA : 0x5)>(t, t);
        } else if constexpr (MP == 4) {
            return dpp<(C & 3) * 0x55>(v);
        } else if constexpr (MP == 2) {
            return dpp<(C | (C << 2) | ((2 + C) << 4) | ((2 + C) << 6))>(v);     // rows are lane pairs of a quad
        } else {
            return v;
        }
    }
    template <int C>
    __device__ __forceinline__ Cx<R> rowb_c(Cx<R> v) const { return {rowb_c<C>(v.re), rowb_c<C>(v.im)}; }
    // value held by lane (r, j): same column, row r
    __device__ __forceinline__ R colb(R v, int r) const { return __shfl(v, r * MP + j, G); }
    __device__ __forceinline__ Cx<R> colb(Cx<R> v, int r) const { return {colb(v.re, r), colb(v.im, r)}; }
    // value held by the transposed lane (j, i)
    __device__ __forceinline__ Cx<R> transp(Cx<R> v) const {
        return {__shfl(v.re, j * MP + i, G), __shfl(v.im, j * MP + i, G)};
    }
    // sum over the rows of a column (every lane of the column gets it).  Lane = i*MP + j, so the rows
    // of a column are MP lanes apart: rotations inside a 16-lane DPP row, then row swaps.
    __device__ __forceinline__ R colsum(R v) const {
        if constexpr (MP == 2) {
            v += dpp<kDppXor2>(v);
        } else if constexpr (MP == 4) {
            v += dpp<kDppRor4>(v);
            v += dpp<kDppRor8>(v);
        } else {
            v += dpp<kDppRor8>(v);
            v = swapsum16(v);
            v = swapsum32(v);
        }
        return v;
    }
    __device__ __forceinline__ unsigned colmax(unsigned v) const {
        unsigned o;
        if constexpr (MP == 2) {
            o = dpp<kDppXor2>(v); v = o > v ? o : v;
        } else if constexpr (MP == 4) {
            o = dpp<kDppRor4>(v); v = o > v ? o : v;
            o = dpp<kDppRor8>(v); v = o > v ? o : v;
        } else {
            o = dpp<kDppRor8>(v); v = o > v ? o : v;
            v = swapmax16(v);
            v = swapmax32(v);
        }
        return v;
    }
    // sum over the columns of a row (every lane of the row gets it): the MP lanes of a row are consecutive
    __device__ __forceinline__ R rowsum(R v) const {
        v += dpp<kDppXor1>(v);
        if constexpr (MP >= 4) v += dpp<kDppXor2>(v);
        if constexpr (MP == 8) v += dpp<kDppHalfMirror>(v);   // quads are uniform: the mirror pairs quad 0 with quad 1
        return v;
    }
    __device__ __forceinline__ Cx<R> rowsum(Cx<R> v) const { return {rowsum(v.re), rowsum(v.im)}; }
    // value held by lane (r, c) of this lane's group
    __device__ __forceinline__ Cx<R> at(Cx<R> v, int r, int c) const {
        return {__shfl(v.re, r * MP + c, G), __shfl(v.im, r * MP + c, G)};
    }
    // the same for compile-time (r, c): with one bin per wavefront it is a v_readlane (scalar broadcast)
    template <int RR, int CC>
    __device__ __forceinline__ Cx<R> at_c(Cx<R> v) const {
        if constexpr (G == 64) {
            auto rl = [](R x) -> R {
                if constexpr (sizeof(R) == 4) {
                    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), RR * MP + CC));
                } else {
                    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), RR * MP + CC),
                                            __builtin_amdgcn_readlane(__double2loint(x), RR * MP + CC));
                }
            };
            return {rl(v.re), rl(v.im)};
        } else {
            return at(v, RR, CC);
        }
    }
    // inverse of a Hermitian positive definite matrix (identity outside M x M): in-place Gauss-Jordan, no pivot
    // search (every pivot of an HPD elimination is a positive Schur complement)
    __device__ __forceinline__ Cx<R> herm_inverse(Cx<R> A, int M) const {
        static_for<MP>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if (k < M) {
                const Cx<R> rk = colb(A, k);                 // A[k][j]
                const Cx<R> ck = rowb_c<k>(A);               // A[i][k]
                const R d = fast_rcp(at_c<k, k>(A).re);      // 1 / A[k][k]: the pivots of this elimination are real (Schur
                                                             // complements of a Hermitian matrix stay Hermitian)
                const Cx<R> rkd = {rk.re * d, rk.im * d};
                if (i == k)
                    A = (j == k) ? Cx<R>{d, R(0)} : rkd;
                else if (j == k)
                    A = Cx<R>{-(ck.re * d), -(ck.im * d)};
                else
                    cfms(A, ck, rkd);
            }
        });
        return A;
    }
    // sum over all MP*MP lanes of the group
    __device__ __forceinline__ R allsum(R v) const {
        v += dpp<kDppXor1>(v);
        v += dpp<kDppXor2>(v);
        if constexpr (MP >= 4) {
            v += dpp<kDppRor4>(v);   // quads are uniform now: two rotations add the four quads of a row
            v += dpp<kDppRor8>(v);
        }
        if constexpr (MP == 8) {
            v = swapsum16(v);
            v = swapsum32(v);
        }
        return v;
    }
    // C = A * B for matrices distributed one element per lane; only the first M rows/cols of the
    // inner index contribute
    __device__ __forceinline__ Cx<R> matmul(Cx<R> A, Cx<R> B, int M) const {
        Cx<R> acc = {R(0), R(0)};
        static_for<MP>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            if (m < M) cfma(acc, rowb_c<m>(A), colb(B, m));
        });
        return acc;
    }
    // Inverse of a general matrix (identity outside M x M) by Gauss-Jordan elimination with partial pivoting on [A | I]: rows
    // never move (the pivot row broadcasts), so row c of the inverse is the row that pivoted column c, divided by its pivot.
    __device__ __forceinline__ Cx<R> inverse_pivoted(Cx<R> A, int M) const {
        Cx<R> Rm = {R(i == j ? 1 : 0), R(0)};
        Cx<R> piv = {R(1), R(0)};
        bool used = false;
        int src = i;
        static_for<MP>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if (c < M) {
                const Cx<R> aic = rowb_c<c>(A);
                float mag = used ? 0.f : (float)(aic.re * aic.re + aic.im * aic.im);
                unsigned key = (__float_as_uint(mag) & ~(unsigned)(MP - 1)) | (unsigned)(MP - 1 - i);
                key = used ? 0u : key;
                key = colmax(key);
                const int p = MP - 1 - (int)(key & (unsigned)(MP - 1));
                if (i == c) src = p;
                const bool isp = (i == p);
                const Cx<R> apc = colb(aic, p);
                const Cx<R> apj = colb(A, p);
                const Cx<R> rpj = colb(Rm, p);
                const Cx<R> fct = cmul(aic, cinv(apc));
                if (isp) {
                    used = true;
                    piv = apc;
                } else {
                    cfms(A, fct, apj);
                    cfms(Rm, fct, rpj);
                    if (j == c) A = {R(0), R(0)};
                }
            }
        });
        return colb(cmul(Rm, cinv(piv)), src);
    }
    // Gauss-Jordan with partial pivoting over columns 0..npiv-1.  rhs is a per-row scalar replicated
    // along the row.  Returns, per lane: perm[c] = row that pivoted column c, piv = pivot element of
    // the lane's own row.
    __device__ __forceinline__ void gauss_jordan(Cx<R>& A, Cx<R>& rhs, int npiv, bool used, int (&perm)[MP],
                                                 Cx<R>& piv) const {
        static_for<MP>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            perm[c] = c;
            if (c < npiv) {
                const Cx<R> aic = rowb_c<c>(A);
                // arg max over rows of |A[i][c]|^2, row index packed into the low mantissa bits
                float mag = used ? 0.f : (float)(aic.re * aic.re + aic.im * aic.im);
                unsigned key = (__float_as_uint(mag) & ~(unsigned)(MP - 1)) | (unsigned)(MP - 1 - i);
                key = used ? 0u : key;
                key = colmax(key);
                const int p = MP - 1 - (int)(key & (unsigned)(MP - 1));
                perm[c] = p;
                const bool isp = (i == p);
                const Cx<R> apc = colb(aic, p);
                const Cx<R> apj = colb(A, p);
                const Cx<R> bp = colb(rhs, p);
                const Cx<R> fct = cmul(aic, cinv(apc));
                if (isp) {
                    used = true;
                    piv = apc;
                } else {
                    cfms(A, fct, apj);
                    cfms(rhs, fct, bp);
                    if (j == c) A = {R(0), R(0)};
                }
            }
        });
    }
};


// ---------------------------------------------------------------------------------------------
// The update for 1 or 2 sources with background channels (K < M), given V_s for every source: see the comment at
// update_bg_kernel (kernels_update.hip).  Lane (i, j) of the bin's MP x MP group holds element [i][j] of
// B = W_hat^H (in/out), C = Cx, V[s] = V_s (identity outside M x M).
// ---------------------------------------------------------------------------------------------
// The part of the chain that needs W_hat^H and Cx only (not the covariances): T = W^H Cx on rows < K, and the transpose of B.
template <int MP, typename R>
__device__ __forceinline__ void bg_pre(const Sq<MP, R>& sq, const Cx<R>& B, const Cx<R>& C, int M, Cx<R>& Tm, Cx<R>& Bt) {
    Tm = sq.matmul(B, C, M);        // rows < K: W^H Cx (rows >= K unused)
    Bt = sq.transp(B);              // lane (i, j): B[j][i]
}
// One source of the chain proper (S = its index), given V_S, its inverse and the running T = W^H Cx and B^T: the IP1 solve +
// normalisation of w_S (overiva.py:181-186) and the update of J from the orthogonality constraint (:189-190).
template <int MP, typename R, int K, int S>
__device__ __forceinline__ void bg_source(const Sq<MP, R>& sq, Cx<R>& B, const Cx<R>& C, const Cx<R>& Vs, const Cx<R>& Vinvs, Cx<R>& Tm, Cx<R>& Bt,
                                          int M) {
    static_assert(K == 1 || K == 2, "closed-form K x K solves");
    constexpr int s = S;
    const int i = sq.i, j = sq.j;
    const Cx<R> zero = {R(0), R(0)};
    // Q = B_tt + B_tb B_bt on lanes i, j < K
    Cx<R> Q = B;
    static_for<MP>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        if (m >= K && m < M) cfma(Q, sq.template rowb_c<m>(B), sq.colb(B, m));
    });
    // u_top = Q^-1 e_s
    Cx<R> u0, u1 = zero;
    if constexpr (K == 1) {
        u0 = cinv(sq.template at_c<0, 0>(Q));
    } else {
        const Cx<R> q00 = sq.template at_c<0, 0>(Q), q01 = sq.template at_c<0, 1>(Q), q10 = sq.template at_c<1, 0>(Q),
                    q11 = sq.template at_c<1, 1>(Q);
        Cx<R> det = cmul(q00, q11);
        cfms(det, q01, q10);
        const Cx<R> idet = cinv(det);
        u0 = s == 0 ? cmul(q11, idet) : cmul(Cx<R>{-q01.re, -q01.im}, idet);
        u1 = s == 0 ? cmul(Cx<R>{-q10.re, -q10.im}, idet) : cmul(q00, idet);
    }
    // u, one entry per column: u_j = u_top[j] (j < K) | sum_m B[j][m] u_top[m] (K <= j < M)
    Cx<R> ub = cmul(sq.colb(Bt, 0), u0);
    if constexpr (K == 2) cfma(ub, sq.colb(Bt, 1), u1);
    // (component-wise selects: a select of an (re, im) pair may be compiled into a two-slot stack array indexed by the lane)
    const bool j0 = j == 0, j1 = K == 2 && j == 1;
    Cx<R> uj = {j0 ? u0.re : (j1 ? u1.re : ub.re), j0 ? u0.im : (j1 ? u1.im : ub.im)};
    if (j >= M) uj = zero;
    // w = V^-1 u (one entry per row), then its copy per column
    Cx<R> wi = sq.rowsum(cmul(Vinvs, uj));
    Cx<R> wj = sq.transp(wi);
    // d = w^H V w, overiva.py:185
    const Cx<R> vw = sq.rowsum(cmul(Vs, wj));
    const R d = sq.allsum(j == 0 && i < M ? wi.re * vw.re + wi.im * vw.im : R(0));
    const R sc = fast_rsqrt(d);
    wi.re *= sc;
    wi.im *= sc;
    wj.re *= sc;
    wj.im *= sc;
    if (i == s && j < M) B = {wj.re, -wj.im};
    // J from the orthogonality constraint, overiva.py:189-190 -> :96-98; row s of W^H Cx = sum_i conj(w_i) Cx[i][:]
    Cx<R> t = cmul(Cx<R>{wi.re, -wi.im}, C);
    t.re = sq.colsum(t.re);
    t.im = sq.colsum(t.im);
    if (i == s) Tm = t;
    Cx<R> Jn;                          // lanes i < K, j >= K: J[i][j - K]
    if constexpr (K == 1) {
        Jn = cmul(sq.colb(Tm, 0), cinv(sq.template at_c<0, 0>(Tm)));
    } else {
        const Cx<R> t00 = sq.template at_c<0, 0>(Tm), t01 = sq.template at_c<0, 1>(Tm), t10 = sq.template at_c<1, 0>(Tm),
                    t11 = sq.template at_c<1, 1>(Tm);
        const Cx<R> r0 = sq.colb(Tm, 0), r1 = sq.colb(Tm, 1);        // Tm[0][j], Tm[1][j]
        Cx<R> det = cmul(t00, t11);
        cfms(det, t01, t10);
        const Cx<R> idet = cinv(det);
        Cx<R> n0 = cmul(t11, r0), n1 = cmul(t00, r1);
        cfms(n0, t01, r1);
        cfms(n1, t10, r0);
        const bool i0 = i == 0;
        Jn = cmul(Cx<R>{i0 ? n0.re : n1.re, i0 ? n0.im : n1.im}, idet);
    }
    // W_hat[m][i] = J[m][i - K]  ->  (W_hat^H)[i][m] = conj, for K <= i < M, m = j < K
    const Cx<R> Jt = sq.transp(Jn);
    if (j < K && i >= K && i < M) B = {Jt.re, -Jt.im};
    Bt = sq.transp(B);
}

// The chain proper: the sources one after the other.
template <int MP, typename R, int K>
__device__ __forceinline__ void bg_core(const Sq<MP, R>& sq, Cx<R>& B, const Cx<R>& C, const Cx<R> (&V)[K], const Cx<R> (&Vinv)[K], Cx<R> Tm,
                                        Cx<R> Bt, int M) {
    bg_source<MP, R, K, 0>(sq, B, C, V[0], Vinv[0], Tm, Bt, M);
    if constexpr (K == 2) bg_source<MP, R, K, 1>(sq, B, C, V[1], Vinv[1], Tm, Bt, M);
}

// everything in one call (the stand-alone update kernels)
template <int MP, typename R, int K>
__device__ __forceinline__ void bg_chain(const Sq<MP, R>& sq, Cx<R>& B, const Cx<R>& C, const Cx<R> (&V)[K], int M) {
    Cx<R> Vinv[K];
#pragma unroll
    for (int s = 0; s < K; ++s) Vinv[s] = sq.herm_inverse(V[s], M);
    Cx<R> Tm, Bt;
    bg_pre<MP, R>(sq, B, C, M, Tm, Bt);
    bg_core<MP, R, K>(sq, B, C, V, Vinv, Tm, Bt, M);
}

}  // namespace oiva
