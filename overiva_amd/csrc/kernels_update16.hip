// Per-bin sequential algebra for 9..16 channels                            reference overiva.py:176-190
//
// Same mathematics as update_sq_kernel (kernels_update.hip) -- per source: A = W_hat^H V_s, Gauss-Jordan with
// partial pivoting for A w = e_s, w /= sqrt(w^H V w), J from the orthogonality constraint -- with ONE WORKGROUP
// per bin: 256 lanes = the 16 x 16 matrix, lane (i, j) = tid / 16, tid % 16 holds element [i][j].  A 16 x 16
// matrix spans four wavefronts, so rows / columns / pivots travel through LDS (two barriers per elimination
// step) instead of lane permutes.  Every lane does O(1) arithmetic per step; 2048 bins = 2048 workgroups run
// concurrently.  (The row-per-lane variant needs ~1 ms for 16 sources x 16 channels; this one a few tens of us.)
#include "oiva_device.h"

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

template <typename R>
struct Lds16 {
    C2<R> mA[N][N + 1];   // left operand of a product (W_hat^H)
    C2<R> mB[N][N + 1];   // right operand (V or Cx)
    C2<R> row[N];         // pivot row
    C2<R> col[N];         // pivot column
    C2<R> q[N];           // solution / per-column scratch
    C2<R> rhs_p;          // right-hand side of the pivot row
    float mag[N];
    int pivrow[N];        // pivrow[c] = row that pivoted column c
    double red[kWaves];
};

// out[i][j] = sum_m L[i][m] * Rm[m][j] over m < M, operands distributed one element per lane
template <typename R>
__device__ __forceinline__ C2<R> matmul(Lds16<R>& s, C2<R> L, C2<R> Rm, int i, int j, int M) {
    __syncthreads();
    s.mA[i][j] = L;
    s.mB[i][j] = Rm;
    __syncthreads();
    C2<R> acc = {R(0), R(0)};
    for (int m = 0; m < M; ++m) {
        const C2<R> a = s.mA[i][m], b = s.mB[m][j];
        acc.re += a.re * b.re - a.im * b.im;
        acc.im += a.re * b.im + a.im * b.re;
    }
    return acc;
}

// Gauss-Jordan with partial pivoting over columns 0..npiv-1.  rhs is a per-row scalar replicated along the
// row; rows with used == true are never chosen.  Afterwards s.pivrow[c] = pivot row of column c; for a pivot
// row, mycol = the column it pivoted and piv its pivot element.
template <typename R>
__device__ __forceinline__ void gauss_jordan(Lds16<R>& s, C2<R>& A, C2<R>& rhs, int npiv, bool used, int& mycol,
                                             C2<R>& piv, int i, int j) {
    for (int c = 0; c < npiv; ++c) {
        if (j == c) s.mag[i] = used ? -1.f : (float)(A.re * A.re + A.im * A.im);
        __syncthreads();
        int p = 0;
        float best = s.mag[0];
#pragma unroll
        for (int r = 1; r < N; ++r) {
            const float v = s.mag[r];
            if (v > best) {
                best = v;
                p = r;
            }
        }
        if (i == p) s.row[j] = A;
        if (j == c) s.col[i] = A;
        if (i == p && j == 0) {
            s.rhs_p = rhs;
            s.pivrow[c] = p;
        }
        __syncthreads();
        const C2<R> apc = s.col[p], aic = s.col[i], apj = s.row[j], bp = s.rhs_p;
        if (i == p) {
            used = true;
            mycol = c;
            piv = apc;
        } else {
            const C2<R> fct = cmul(aic, cinv(apc));
            const C2<R> d1 = cmul(fct, apj), d2 = cmul(fct, bp);
            A.re -= d1.re;
            A.im -= d1.im;
            rhs.re -= d2.re;
            rhs.im -= d2.im;
            if (j == c) A = {R(0), R(0)};
        }
    }
}

template <typename R>
__device__ __forceinline__ R block_sum16(Lds16<R>& s, R v) {
    return (R)block_sum((double)v, s.red);
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

template <typename R>
__global__ __launch_bounds__(kBlock) void update_lds16_kernel(UpdateArgs a) {
    __shared__ Lds16<R> s;
    const int tid = threadIdx.x;
    const int i = tid >> 4, j = tid & 15;
    const int f = blockIdx.x;
    const int M = a.M, K = a.K;
    const int NA = M * M;
    const bool in = i < M && j < M;
    const C2<R> zero = {R(0), R(0)};
    const C2<R> eye = {R(i == j ? 1 : 0), R(0)};

    // B[i][j] = (W_hat^H)[i][j] = conj(W_hat[j][i]); identity outside M x M
    C2<R> B = eye;
    if (in) {
        R vr, vi;
        load_what<R>(a, ((size_t)f * M + j) * M + i, vr, vi);
        B = {vr, -vi};
    }
    if (a.wscale != nullptr && i < K) {   // overiva.py:163 / :167
        const R sc = R(1) / R(a.wscale[i]);
        B.re *= sc;
        B.im *= sc;
    }
    int off = 0;
    float sgn = 0.f;
    if (in) herm_off(M, i, j, off, sgn);
    C2<R> C = zero;
    if (in) {
        const double* p = a.Cx + (size_t)f * NA + off;
        C.re = R(p[0]);
        if (sgn != 0.f) C.im = R(sgn * p[1]);
    }
    C2<R> Tm = zero;                      // rows < K: W^H Cx
    if (K < M) Tm = matmul(s, B, C, i, j, M);

    const int nsrc = a.init_only ? 0 : K;
    const R invT = R(1) / R(a.T);
    // V_s[i][j]: fixed-order fp64 sum of the frame-split partials; the loads of source s+1 are issued before
    // source s is solved
    auto load_v = [&](int src) {
        C2<R> V = zero;
        if (in) {
            double sr, si;
            sum_vpart(a.Vpart, a.vpart_f64, ((size_t)f * K + src) * NA + off, (size_t)a.F * K * NA, a.nsplit, sgn != 0.f, sr, si);
            V.re = R(sr) * invT;
            V.im = R(si) * R(sgn) * invT;
        }
        return V;
    };
    C2<R> Vnext = zero;
    if (nsrc > 0) Vnext = load_v(0);
    for (int src = 0; src <= nsrc; ++src) {
        const bool solve = src < nsrc;
        if (!solve && !a.init_only) break;
        C2<R> wi = zero, wj = zero;
        if (solve) {
            const C2<R> V = Vnext;
            if (src + 1 < nsrc) Vnext = load_v(src + 1);
            C2<R> A = matmul(s, B, V, i, j, M);   // W_hat^H V
            if (!in) A = eye;
            C2<R> rhs = {R(i == src ? 1 : 0), R(0)};
            int mycol = i;
            C2<R> piv = {R(1), R(0)};
            gauss_jordan(s, A, rhs, N, false, mycol, piv, i, j);
            // w[c] = rhs / pivot on the row that pivoted column c
            __syncthreads();
            if (j == 0) s.q[mycol] = cmul(rhs, cinv(piv));
            __syncthreads();
            wi = s.q[i];
            wj = s.q[j];
            // d = w^H V w  (real, positive)
            const C2<R> vw = cmul(V, wj);
            const R d = block_sum16(s, wi.re * vw.re + wi.im * vw.im);
            const R sc = R(1) / sqrt(d);
            wi.re *= sc;
            wi.im *= sc;
            wj.re *= sc;
            wj.im *= sc;
            if (i == src) B = cconj(wj);
        }
        if (K < M) {
            if (solve) {
                // row src of W^H Cx = sum_m conj(w_m) Cx[m][:]  (column sums over i)
                __syncthreads();
                s.mA[i][j] = cmul(cconj(wi), C);
                __syncthreads();
                C2<R> t = zero;
                for (int m = 0; m < M; ++m) {
                    t.re += s.mA[m][j].re;
                    t.im += s.mA[m][j].im;
                }
                if (i == src) Tm = t;
            }
            C2<R> G = (i < K) ? Tm : eye;
            C2<R> dummy = zero;
            int mycol = i;
            C2<R> piv = {R(1), R(0)};
            gauss_jordan(s, G, dummy, K, i >= K, mycol, piv, i, j);
            // J[m][j-K] = G[pivrow[m]][j] / pivot: the pivot rows publish their normalised rows at index m
            __syncthreads();
            if (i < K) s.mB[mycol][j] = cmul(G, cinv(piv));
            __syncthreads();
            // W_hat[m][i] = J[m][i-K]  ->  (W_hat^H)[i][m] = conj, for i >= K, m = j < K
            if (i >= K && i < M && j < K) B = cconj(s.mB[j][i]);
        }
    }
    if (in) store_what<R>(a, ((size_t)f * M + j) * M + i, B.re, -B.im);
}

}  // namespace

hipError_t launch_update_lds16(hipStream_t s, const UpdateArgs& a) {
    dim3 grid(a.F);
    if (a.use_double)
        hipLaunchKernelGGL((update_lds16_kernel<double>), grid, dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL((update_lds16_kernel<float>), grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

}  // namespace oiva
