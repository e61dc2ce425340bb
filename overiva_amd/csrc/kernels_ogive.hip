// Per-bin part of OGIVE (orthogonally constrained independent vector extraction of one source by gradient steps)
//                                                                                       reference ive.py:96-246
// The streaming work of an epoch is done by the kernels the AuxIVA path already has: the demix + power pass
// (ive.py:196, the norm of :210/:213), the activation (r = ||y|| / sqrt(F) | ||y||^2 / F, floor, :209-217) and the
// weighted covariance V = (1/T) sum_t r_inv x x^H with K = 1, because
//     X^T psi = sum_t r_inv x conj(y) = T V w     and     zeta = sum_t r_inv |y|^2 = T w^H V w      (ive.py:221-227)
// so x_psi = V w / (w^H V w).  What is left per bin is a handful of M-vector operations, float64 throughout:
//     demixing step   delta = a - x_psi;  w += mu delta;  a = Cx w / Re(w^H Cx w)                    (:231-232, :136-139)
//     mixing step     delta = w - lambda_a Cx^-1 x_psi;  a += mu delta;  w = lambda_a Cx^-1 a        (:236-237, :141-144)
//     lambda_a = 1 / Re(a^H Cx^-1 a) is refreshed for EVERY bin in every epoch                       (:143)
//     switching criterion every 10 epochs                                                             (:146-166)
// and the stopping rule max_f ||delta_f|| < tol (:243-246): the step kernel raises a device flag after which it leaves
// the state untouched, so the host may run epochs in chunks without a round trip per epoch.
#include "oiva_internal.h"

namespace oiva {
namespace {

constexpr int MX = OIVA_MAX_CHANNELS;

struct Z {
    double re, im;
};
__device__ __forceinline__ Z zmul(Z a, Z b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ Z zconj(Z a) { return {a.re, -a.im}; }
__device__ __forceinline__ Z zinv(Z a) {
    const double d = 1.0 / (a.re * a.re + a.im * a.im);
    return {a.re * d, -a.im * d};
}
__device__ __forceinline__ void zacc(Z& s, Z a, Z b) {   // s += a * b
    s.re += a.re * b.re - a.im * b.im;
    s.im += a.re * b.im + a.im * b.re;
}

// entry (i, j) of a packed Hermitian matrix held as M diagonals then (re, im) of every i < j pair
__device__ __forceinline__ Z herm_at(const double* __restrict__ p, int M, int i, int j) {
    if (i == j) return {p[i], 0.};
    if (i < j) {
        const int o = herm_pair_index(M, i, j);
        return {p[o], p[o + 1]};
    }
    const int o = herm_pair_index(M, j, i);
    return {p[o], -p[o + 1]};
}

// v = Cx^-1 u with Cx^-1 stored as full (M, M) complex
__device__ __forceinline__ void apply_inv(const double2* __restrict__ Ci, int M, const Z (&u)[MX], Z (&v)[MX]) {
    for (int i = 0; i < M; ++i) {
        Z s = {0., 0.};
        for (int j = 0; j < M; ++j) zacc(s, Z{Ci[i * M + j].x, Ci[i * M + j].y}, u[j]);
        v[i] = s;
    }
}

__device__ __forceinline__ void load_w(const OgiveState& st, int f, int M, Z (&w)[MX]) {
    for (int m = 0; m < M; ++m) {
        const double2 v = st.What64[((size_t)f * M + m) * M];          // column 0 of W_hat
        w[m] = {v.x, v.y};
    }
}
__device__ __forceinline__ void store_w(const OgiveState& st, int f, int M, const Z (&w)[MX]) {
    for (int m = 0; m < M; ++m) {
        st.What64[((size_t)f * M + m) * M] = make_double2(w[m].re, w[m].im);
        st.What[((size_t)f * M + m) * M] = make_float2((float)w[m].re, (float)w[m].im);
    }
}
// a = Cx w / Re(w^H Cx w)      ive.py:136-139
__device__ __forceinline__ void a_from_w(const double* __restrict__ cx, int M, const Z (&w)[MX], Z (&a)[MX]) {
    double den = 0.;
    for (int i = 0; i < M; ++i) {
        Z s = {0., 0.};
        for (int j = 0; j < M; ++j) zacc(s, herm_at(cx, M, i, j), w[j]);
        a[i] = s;
        den += w[i].re * s.re + w[i].im * s.im;                          // Re(conj(w_i) s)
    }
    const double l = 1.0 / den;
    for (int i = 0; i < M; ++i) a[i] = {a[i].re * l, a[i].im * l};
}

// prologue ive.py:100-102,136-139,173-180: Cx^-1 (Gauss-Jordan on the Hermitian positive definite Cx), ||Cx||_F,
// a from the initial w, delta = 0, lambda_a = 0, the initial step selection
__global__ __launch_bounds__(64) void ogive_init_kernel(OgiveState st, int F, int M, int mode) {
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f >= F) return;
    const double* cx = st.Cx + (size_t)f * M * M;
    double2* Ci = st.CxInv + (size_t)f * M * M;
    double nrm = 0.;
    // [A | I] -> [I | A^-1] in place on Ci (A read from the packed Cx), partial pivoting on the real magnitude
    Z A[MX][MX], B[MX][MX];
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) {
            A[i][j] = herm_at(cx, M, i, j);
            B[i][j] = {i == j ? 1. : 0., 0.};
            nrm += A[i][j].re * A[i][j].re + A[i][j].im * A[i][j].im;
        }
    for (int c = 0; c < M; ++c) {
        int p = c;
        double best = A[c][c].re * A[c][c].re + A[c][c].im * A[c][c].im;
        for (int r = c + 1; r < M; ++r) {
            const double m = A[r][c].re * A[r][c].re + A[r][c].im * A[r][c].im;
            if (m > best) {
                best = m;
                p = r;
            }
        }
        for (int j = 0; j < M; ++j) {
            const Z ta = A[c][j], tb = B[c][j];
            A[c][j] = A[p][j];
            A[p][j] = ta;
            B[c][j] = B[p][j];
            B[p][j] = tb;
        }
        const Z d = zinv(A[c][c]);
        for (int j = 0; j < M; ++j) {
            A[c][j] = zmul(A[c][j], d);
            B[c][j] = zmul(B[c][j], d);
        }
        for (int r = 0; r < M; ++r) {
            if (r == c) continue;
            const Z fct = A[r][c];
            for (int j = 0; j < M; ++j) {
                const Z pa = zmul(fct, A[c][j]), pb = zmul(fct, B[c][j]);
                A[r][j] = {A[r][j].re - pa.re, A[r][j].im - pa.im};
                B[r][j] = {B[r][j].re - pb.re, B[r][j].im - pb.im};
            }
        }
    }
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) Ci[i * M + j] = make_double2(B[i][j].re, B[i][j].im);
    st.CxNorm[f] = sqrt(nrm);
    Z w[MX], a[MX];
    load_w(st, f, M, w);
    a_from_w(cx, M, w, a);
    for (int m = 0; m < M; ++m) {
        st.A[(size_t)f * M + m] = make_double2(a[m].re, a[m].im);
        st.Delta[(size_t)f * M + m] = make_double2(0., 0.);
    }
    st.Lambda[f] = 0.;
    st.DoA[f] = mode == OIVA_OGIVE_MIX ? 1 : 0;         // ive.py:175-180 (switching decides at epoch 0, :192-193)
    st.DoW[f] = mode == OIVA_OGIVE_MIX ? 0 : 1;
    st.Dnorm[f] = 0.;
    if (f == 0) {
        st.ctrl[0] = 0;      // done
        st.ctrl[1] = 0;      // epochs run
        st.ctrl[2] = 0;      // workgroups of the step kernel that have finished the current epoch
        st.maxdelta[0] = 0.;
        st.maxdelta[1] = 0.; // running max of ||delta_f|| (bit pattern) of the current epoch
    }
}

// switching criterion, ive.py:146-166
__global__ __launch_bounds__(64) void ogive_switch_kernel(OgiveState st, int F, int M) {
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f >= F || st.ctrl[0]) return;
    const double* cx = st.Cx + (size_t)f * M * M;
    Z an[MX], bn[MX];
    const Z a0 = {st.A[(size_t)f * M].x, st.A[(size_t)f * M].y};
    const Z ia0 = zinv(a0);
    for (int m = 0; m < M; ++m) an[m] = zmul(Z{st.A[(size_t)f * M + m].x, st.A[(size_t)f * M + m].y}, ia0);
    for (int i = 0; i < M; ++i) {
        Z s = {0., 0.};
        for (int j = 0; j < M; ++j) zacc(s, herm_at(cx, M, i, j), an[j]);
        bn[i] = s;
    }
    const Z lmb = bn[0];
    const Z il = zinv(lmb);
    double p1 = 0., nb = 0.;
    for (int m = 0; m < M; ++m) {
        bn[m] = zmul(bn[m], il);
        const double dr = an[m].re - bn[m].re, di = an[m].im - bn[m].im;
        p1 += dr * dr + di * di;
        nb += bn[m].re * bn[m].re + bn[m].im * bn[m].im;
    }
    p1 = sqrt(p1) / st.CxNorm[f];
    double p2 = 0.;
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) {
            Z cbb = zmul(lmb, zmul(bn[i], zconj(bn[j])));          // lmb * b b^H / ||b||^2
            cbb = {cbb.re / nb, cbb.im / nb};
            const Z c = herm_at(cx, M, i, j);
            const double dr = c.re - cbb.re, di = c.im - cbb.im;
            p2 += dr * dr + di * di;
        }
    const double kappa = p1 * sqrt(p2) / sqrt((double)M);
    st.DoA[f] = kappa >= 0.1 ? 1 : 0;                                // ive.py:163-166 (NaN -> demixing step off too)
    st.DoW[f] = kappa < 0.1 ? 1 : 0;
}

// one epoch of the per-bin part, ive.py:221-241, and the stopping rule max_f ||delta_f|| < tol (ive.py:243-246).
// MP lanes per bin (MP = channels rounded up to 4 | 8 | 16): lane i owns row i of every matrix-vector product and
// entry i of w, a, x_psi, delta; vectors and the terms of the scalar products are exchanged through LDS and summed by
// every lane in index order, so the arithmetic is the sequential one of the restatement.  The last workgroup to finish
// (ticket counter) folds the per-workgroup maxima into the stop flag -- no separate reduction launch.
template <int MP>
__global__ __launch_bounds__(64) void ogive_step_kernel(OgiveState st, const void* __restrict__ Vpart, int vpart_f64, int nsplit,
                                                        int T, int F, int M, double mu, double tol) {
    constexpr int kBins = 64 / MP;
    __shared__ Z sv[kBins][MP];
    __shared__ double sr[kBins][MP];
    __shared__ double sdn[kBins];
    if (st.ctrl[0]) return;                                              // grid-uniform: the rule was met in an earlier epoch
    const int g = threadIdx.x / MP, i = threadIdx.x % MP;
    const int fraw = blockIdx.x * kBins + g;
    const bool live = fraw < F && i < M;
    const int f = fraw < F ? fraw : F - 1, ic = i < M ? i : M - 1;      // padding lanes shadow a real one and store nothing
    const int NA = M * M;
    auto sum_seq = [&](double term) {                                    // sum_j term_j, j ascending, on every lane of the bin
        __syncthreads();
        sr[g][i] = term;
        __syncthreads();
        double s = 0.;
#pragma unroll
        for (int j = 0; j < MP; ++j)
            if (j < M) s += sr[g][j];
        return s;
    };
    auto share = [&](Z v) {
        __syncthreads();
        sv[g][i] = v;
        __syncthreads();
    };
    // row ic of a packed Hermitian matrix times the shared vector
    auto herm_row_times = [&](auto&& entry) {
        Z s = {0., 0.};
#pragma unroll
        for (int j = 0; j < MP; ++j)
            if (j < M) zacc(s, entry(j), sv[g][j]);
        return s;
    };
    auto packed_entry = [&](auto&& load, int j) -> Z {                   // entry (ic, j); load(e) reads packed slot e
        if (j == ic) return {load(ic), 0.};
        const int lo = j < ic ? j : ic, hi = j < ic ? ic : j;
        const int o = herm_pair_index(M, lo, hi);
        const double re = load(o), im = load(o + 1);
        return {re, j < ic ? -im : im};
    };
    const double2 w0 = st.What64[((size_t)f * M + ic) * M], a0 = st.A[(size_t)f * M + ic];
    Z w = {w0.x, w0.y}, a = {a0.x, a0.y};
    const bool do_a = st.DoA[f] != 0, do_w = st.DoW[f] != 0;
    // x_psi = V w / (w^H V w); V: fixed-order sum of the frame-split partials, / T
    share(w);
    Z xpsi = herm_row_times([&](int j) {
        return packed_entry(
            [&](int e) {
                double s = 0.;
                for (int sp = 0; sp < nsplit; ++sp) s += load_vpart(Vpart, vpart_f64, ((size_t)sp * F + f) * NA + e);
                return s / (double)T;
            },
            j);
    });
    const double den = sum_seq(w.re * xpsi.re + w.im * xpsi.im);        // w^H V w is real
    xpsi = {xpsi.re / den, xpsi.im / den};
    const double* cx = st.Cx + (size_t)f * NA;
    const double2* Ci = st.CxInv + ((size_t)f * M + ic) * M;
    auto inv_row_times = [&]() {                                         // row ic of Cx^-1 times the shared vector
        Z s = {0., 0.};
#pragma unroll
        for (int j = 0; j < MP; ++j)
            if (j < M) zacc(s, Z{Ci[j].x, Ci[j].y}, sv[g][j]);
        return s;
    };
    Z dl = {0., 0.};
    if (do_w) {                                                          // ive.py:231-232
        dl = {a.re - xpsi.re, a.im - xpsi.im};
        w = {w.re + mu * dl.re, w.im + mu * dl.im};
    }
    share(do_w ? w : xpsi);
    Z t = {0., 0.};
    if (do_w) {                                                          // ive.py:240 -> :136-139: a = Cx w / Re(w^H Cx w)
        t = herm_row_times([&](int j) { return packed_entry([&](int e) { return cx[e]; }, j); });
    } else if (do_a) {                                                   // ive.py:236-237
        t = inv_row_times();
        const double la = st.Lambda[f];
        dl = {w.re - t.re * la, w.im - t.im * la};
        a = {a.re + mu * dl.re, a.im + mu * dl.im};
    }
    const double wcw = sum_seq(do_w ? w.re * t.re + w.im * t.im : 0.);
    if (do_w) {
        const double l = 1.0 / wcw;
        a = {t.re * l, t.im * l};
    }
    const double dsq = sum_seq(dl.re * dl.re + dl.im * dl.im);
    const bool stepped = do_w || do_a;
    // lambda_a = 1 / Re(a^H Cx^-1 a) for every bin, w = lambda_a Cx^-1 a where the mixing step ran (ive.py:141-144)
    share(a);
    t = inv_row_times();
    const double la = 1.0 / sum_seq(a.re * t.re + a.im * t.im);
    if (do_a) w = {t.re * la, t.im * la};
    double dn = stepped ? sqrt(dsq) : st.Dnorm[f];                       // bins without a step keep their last delta
    if (live) {
        st.A[(size_t)f * M + i] = make_double2(a.re, a.im);
        st.What64[((size_t)f * M + i) * M] = make_double2(w.re, w.im);
        st.What[((size_t)f * M + i) * M] = make_float2((float)w.re, (float)w.im);
        if (stepped) st.Delta[(size_t)f * M + i] = make_double2(dl.re, dl.im);
        if (i == 0) {
            st.Lambda[f] = la;
            if (stepped) st.Dnorm[f] = dn;
        }
    }
    // stopping rule: a non-negative double orders like its bit pattern and every NaN sorts above +inf, which is numpy's
    // rule too (the max of an array holding NaN is NaN, NaN < tol is False)
    if (i == 0) sdn[g] = fraw < F ? dn : 0.;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long top = 0;
        for (int k = 0; k < kBins; ++k) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(sdn[k]) & 0x7fffffffffffffffull;
            top = bits > top ? bits : top;
        }
        unsigned long long* slot = reinterpret_cast<unsigned long long*>(st.maxdelta + 1);
        atomicMax(slot, top);
        __threadfence();
        if (atomicAdd(reinterpret_cast<unsigned int*>(st.ctrl + 2), 1u) == gridDim.x - 1) {
            __threadfence();
            const double mx = __longlong_as_double((long long)atomicExch(slot, 0ull));
            st.ctrl[2] = 0;
            st.ctrl[1] += 1;
            st.maxdelta[0] = mx;
            if (mx < tol) st.ctrl[0] = 1;
        }
    }
}

}  // namespace

hipError_t launch_ogive_init(hipStream_t s, const OgiveState& st, int F, int M, int mode) {
    hipLaunchKernelGGL(ogive_init_kernel, dim3((F + 63) / 64), dim3(64), 0, s, st, F, M, mode);
    return hipGetLastError();
}
hipError_t launch_ogive_switch(hipStream_t s, const OgiveState& st, int F, int M) {
    hipLaunchKernelGGL(ogive_switch_kernel, dim3((F + 63) / 64), dim3(64), 0, s, st, F, M);
    return hipGetLastError();
}
hipError_t launch_ogive_step(hipStream_t s, const OgiveState& st, const void* Vpart, bool vpart_f64, int nsplit, int T, int F,
                             int M, double mu, double tol) {
    const int mp = M <= 4 ? 4 : (M <= 8 ? 8 : 16);
    const dim3 grid((F + 64 / mp - 1) / (64 / mp));
    auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, dim3(64), 0, s, st, Vpart, vpart_f64 ? 1 : 0, nsplit, T, F, M, mu, tol); };
    if (mp == 4) go(ogive_step_kernel<4>);
    else if (mp == 8) go(ogive_step_kernel<8>);
    else go(ogive_step_kernel<16>);
    return hipGetLastError();
}

}  // namespace oiva
