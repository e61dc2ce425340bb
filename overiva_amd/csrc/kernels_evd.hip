// Principal subspace of the input covariance on the device                      reference auxiva_pca.py:71-81
//
// auxiva_pca reduces the M channels to the K principal components of Cx[f] before a determined AuxIVA:
//     v, w = np.linalg.eigh(covmat);  new_X = X conj(w[:, :, -n_src:])          (auxiva_pca.py:75-81)
// i.e. the demixing matrix of the projection is P = w[:, :, -K:], the eigenvectors of the K largest eigenvalues in
// ascending order of the eigenvalue.  One wavefront per bin runs a cyclic complex Jacobi iteration in float64 on the
// (M x M) Hermitian matrix held in LDS: the round-robin ordering gives M/2 disjoint pairs per step, whose rotations
// are computed by M/2 lanes and applied by all 64 (columns, then rows; the eigenvector matrix gets the column
// rotations).  It converges quadratically -- 3 to 6 sweeps at M = 4..16 -- and stops on off(A)^2 < 1e-30 diag(A)^2.
// The phase of each eigenvector is whatever the rotations leave (LAPACK's is a convention too): the result of
// auxiva_pca does not depend on it, because a phase of a principal component turns into a phase of the demixed
// sources, which the projection back onto the original channel 0 (auxiva_pca.py:89-90) removes.
#include "oiva_device.h"

namespace oiva {
namespace {

constexpr int NMAX = OIVA_MAX_CHANNELS;

struct Zd {
    double re, im;
};
__device__ __forceinline__ Zd zmul(Zd a, Zd b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ Zd zconj(Zd a) { return {a.re, -a.im}; }
__device__ __forceinline__ Zd zsub(Zd a, Zd b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ Zd zadd(Zd a, Zd b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ Zd zscale(Zd a, double c) { return {a.re * c, a.im * c}; }

// pair k of step r of the round-robin tournament on n (even) players
__device__ __forceinline__ void pair_of(int n, int r, int k, int& p, int& q) {
    if (k == 0) {
        p = n - 1;
        q = r;
    } else {
        p = (r + k) % (n - 1);
        q = (r - k + (n - 1)) % (n - 1);
    }
}

// lapack_phase: the convention of LAPACK's zgeev, which numpy.linalg.eig -- the call of the reference's init_eig,
// overiva.py:106-109 -- inherits: every eigenvector is scaled by a phase that makes its largest component real (and
// positive); W = conj(vecs) as the reference stores it.  Without it the vectors keep the Jacobi rotations' phases.
__global__ __launch_bounds__(64) void pca_subspace_kernel(const double* __restrict__ Cx, float2* __restrict__ What,
                                                          double2* __restrict__ What64, double* __restrict__ evals, int F, int M,
                                                          int K, int lapack_phase) {
    __shared__ Zd A[NMAX][NMAX + 1], V[NMAX][NMAX + 1], B[NMAX][NMAX + 1];
    __shared__ double rc[NMAX / 2];
    __shared__ Zd rs[NMAX / 2];
    __shared__ double red[64];
    __shared__ int rank[NMAX];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = (M + 1) & ~1;                          // an odd size plays with one idle (zero) index
    const double* cx = Cx + (size_t)f * M * M;
    for (int e = tid; e < n * n; e += 64) {
        const int i = e / n, j = e % n;
        Zd a = {0., 0.};
        if (i < M && j < M) {
            if (i == j) {
                a = {cx[i], 0.};
            } else {
                const int lo = i < j ? i : j, hi = i < j ? j : i;
                const int o = herm_pair_index(M, lo, hi);
                a = {cx[o], i < j ? cx[o + 1] : -cx[o + 1]};
            }
        }
        A[i][j] = a;
        V[i][j] = {i == j ? 1. : 0., 0.};
    }
    __syncthreads();
    constexpr int kMaxSweeps = 30;
    for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
        // off(A)^2 against diag(A)^2
        double off = 0., dg = 0.;
        for (int e = tid; e < n * n; e += 64) {
            const int i = e / n, j = e % n;
            const double m = A[i][j].re * A[i][j].re + A[i][j].im * A[i][j].im;
            if (i == j) dg += m;
            else off += m;
        }
        red[tid] = off;
        __syncthreads();
        for (int s = 32; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        off = red[0];
        __syncthreads();
        red[tid] = dg;
        __syncthreads();
        for (int s = 32; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        dg = red[0];
        __syncthreads();
        if (!(off > 1e-30 * dg)) break;                  // uniform
        for (int r = 0; r < n - 1; ++r) {
            if (tid < n / 2) {
                int p, q;
                pair_of(n, r, tid, p, q);
                const Zd apq = A[p][q];
                const double m = sqrt(apq.re * apq.re + apq.im * apq.im);
                double c = 1.;
                Zd se = {0., 0.};
                if (m > 1e-300) {
                    const double tau = (A[q][q].re - A[p][p].re) / (2. * m);
                    const double t = (tau >= 0. ? 1. : -1.) / (fabs(tau) + sqrt(1. + tau * tau));
                    c = 1. / sqrt(1. + t * t);
                    const double s = t * c;
                    se = {s * apq.re / m, s * apq.im / m};       // s e^{i phi}
                }
                rc[tid] = c;
                rs[tid] = se;
            }
            __syncthreads();
            // columns: (A J)[:, p] = A[:, p] c - A[:, q] conj(se);  (A J)[:, q] = A[:, p] se + A[:, q] c   (and V)
            for (int e = tid; e < n * (n / 2); e += 64) {
                const int i = e / (n / 2), k = e % (n / 2);
                int p, q;
                pair_of(n, r, k, p, q);
                const double c = rc[k];
                const Zd se = rs[k];
                const Zd ap = A[i][p], aq = A[i][q];
                B[i][p] = zsub(zscale(ap, c), zmul(aq, zconj(se)));
                B[i][q] = zadd(zmul(ap, se), zscale(aq, c));
                const Zd vp = V[i][p], vq = V[i][q];
                V[i][p] = zsub(zscale(vp, c), zmul(vq, zconj(se)));
                V[i][q] = zadd(zmul(vp, se), zscale(vq, c));
            }
            __syncthreads();
            // rows: (J^H B)[p, :] = c B[p, :] - se B[q, :];  (J^H B)[q, :] = conj(se) B[p, :] + c B[q, :]
            for (int e = tid; e < n * (n / 2); e += 64) {
                const int j = e / (n / 2), k = e % (n / 2);
                int p, q;
                pair_of(n, r, k, p, q);
                const double c = rc[k];
                const Zd se = rs[k];
                const Zd bp = B[p][j], bq = B[q][j];
                A[p][j] = zsub(zscale(bp, c), zmul(se, bq));
                A[q][j] = zadd(zmul(zconj(se), bp), zscale(bq, c));
            }
            __syncthreads();
        }
    }
    // ascending rank of every eigenvalue (ties by index), the K largest go to columns rank - (M - K)
    if (tid < M) {
        const double l = A[tid][tid].re;
        int rk = 0;
        for (int i = 0; i < M; ++i) {
            const double li = A[i][i].re;
            rk += (li < l || (li == l && i < tid)) ? 1 : 0;
        }
        rank[tid] = rk;
        if (evals) evals[(size_t)f * M + rk] = l;
    }
    __syncthreads();
    // W_hat = [P | [0; -I]]; J is filled by the orthogonality constraint afterwards (overiva.py:120-123)
    for (int e = tid; e < M * M; e += 64) {
        const int r = e / M, col = e % M;
        Zd v = {0., 0.};
        if (col >= K && r == col) v = {-1., 0.};
        const size_t o = (size_t)f * M * M + e;
        What64[o] = make_double2(v.re, v.im);
        What[o] = make_float2((float)v.re, (float)v.im);
    }
    __syncthreads();
    if (lapack_phase) {
        if (tid < M) {                                   // column tid: phase of its largest component (first on ties)
            int kmax = 0;
            double best = -1.;
            for (int r = 0; r < M; ++r) {
                const double m = V[r][tid].re * V[r][tid].re + V[r][tid].im * V[r][tid].im;
                if (m > best) {
                    best = m;
                    kmax = r;
                }
            }
            const double a = sqrt(best);
            B[0][tid] = a > 0. ? Zd{V[kmax][tid].re / a, -V[kmax][tid].im / a} : Zd{1., 0.};   // conj(v_k) / |v_k|
        }
        __syncthreads();
    }
    for (int e = tid; e < M * M; e += 64) {
        const int r = e / M, j = e % M;
        const int col = rank[j] - (M - K);
        if (col >= 0) {
            Zd v = V[r][j];
            if (lapack_phase) v = zconj(zmul(v, B[0][j]));   // largest component real, then W = conj(vecs)
            const size_t o = ((size_t)f * M + r) * M + col;
            What64[o] = make_double2(v.re, v.im);
            What[o] = make_float2((float)v.re, (float)v.im);
        }
    }
}

}  // namespace

hipError_t launch_pca_subspace(hipStream_t s, const double* Cx, float2* What, double2* What64, double* evals, int F, int M, int K,
                               bool lapack_phase) {
    hipLaunchKernelGGL(pca_subspace_kernel, dim3(F), dim3(64), 0, s, Cx, What, What64, evals, F, M, K, lapack_phase ? 1 : 0);
    return hipGetLastError();
}

}  // namespace oiva
