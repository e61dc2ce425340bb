// Per-bin sequential algebra for 9..16 channels                            reference overiva.py:176-190
//
// Same mathematics as update_sq_kernel (kernels_update.hip) -- per source: A = W_hat^H V_s, Gauss-Jordan with
// partial pivoting for A w = e_s, w /= sqrt(w^H V w), J from the orthogonality constraint -- with ONE WORKGROUP
// per bin: 256 lanes = the 16 x 16 matrix, lane (i, j) = tid / 16, tid % 16 holds element [i][j].  A 16 x 16
// matrix spans four wavefronts, so rows / columns / pivots travel through LDS (two barriers per elimination
// step) instead of lane permutes.  Every lane does O(1) arithmetic per step; 2048 bins = 2048 workgroups run
// concurrently.  (The row-per-lane variant needs ~1 ms for 16 sources x 16 channels; this one a few tens of us.)
#include "oiva_device.h"

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


// ---------------------------------------------------------------------------------------------------------------
// Determined case K == M (AuxIVA, overiva.py:176-181 only): ONE WAVEFRONT per bin.
// Lane l = (i, q) = (l >> 2, l & 3) owns row i of the 16 x 16 matrices, columns q, q + 4, q + 8, q + 12 (element e is
// column 4e + q).  A row's 16 entries sit in one quad, so a column entry reaches its row by a quad-permute DPP; the
// pivot is an arg-max over a packed (magnitude, row) key by two row rotations and two lane exchanges; the pivot row is
// published through LDS (same-address reads broadcast).  Nothing crosses a wavefront: no barrier waits on another wave,
// and Gauss-Jordan skips the columns that are already eliminated (known at compile time in the unrolled loop).
// Instruction count per bin is about a fifth of the workgroup-per-bin form above.
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

template <typename R, typename VT>
__global__ __launch_bounds__(64) void update_det16_kernel(UpdateArgs a) {
    __shared__ LdsDet<R> s;
    const int lane = threadIdx.x, i = lane >> 2, q = lane & 3;
    const int f = blockIdx.x, M = a.M, NA = M * M;
    const C2<R> zero = {R(0), R(0)};
    // B = W_hat^H, identity outside M x M
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
    const R invT = R(1) / R(a.T);
    // V_s arrives as the packed Hermitian block of every frame split ([split][bin][source][M * M]): lane l fetches
    // values 4l .. 4l + 3 of each block (16-byte loads when M is even), the splits are added in order in float64, the
    // sum goes to LDS and every lane gathers its four entries.  The loads of source s + 1 are in flight while source s
    // is solved.
    constexpr int kAhead = 8;                                 // splits held in registers across a solve
    const VT* vbase = static_cast<const VT*>(a.Vpart);
    const size_t vstride = (size_t)a.F * M * NA;
    const int nahead = a.nsplit < kAhead ? a.nsplit : kAhead;
    auto fetch = [&](int src, int sp, VT (&raw)[4]) {
        const VT* p = vbase + (size_t)sp * vstride + ((size_t)f * M + src) * NA + lane * 4;
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
    fetch_ahead(0);
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
        C2<R> V[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            V[e] = zero;
            if (in[e]) {
                V[e].re = s.pk[off[e]] * invT;
                if (sgn[e] != 0.f) V[e].im = s.pk[off[e] + 1] * R(sgn[e]) * invT;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) s.V[i][q * 4 + e] = V[e];
        wave_lds_sync();
        // A = W_hat^H V: A[i][4e + q] = sum_m B[i][m] V[m][4e + q]
        C2<R> A[4] = {zero, zero, zero, zero};
#pragma unroll
        for (int me = 0; me < 4; ++me) {
            auto term = [&](C2<R> b, int m) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const C2<R> v = s.V[m][q * 4 + e];
                    A[e].re += b.re * v.re - b.im * v.im;
                    A[e].im += b.re * v.im + b.im * v.re;
                }
            };
            term(quad_bcast<0>(B[me]), 4 * me + 0);
            term(quad_bcast<1>(B[me]), 4 * me + 1);
            term(quad_bcast<2>(B[me]), 4 * me + 2);
            term(quad_bcast<3>(B[me]), 4 * me + 3);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (!in[e]) A[e] = {R(i == 4 * e + q ? 1 : 0), R(0)};
        // Gauss-Jordan with partial pivoting on [A | e_src]
        C2<R> rhs = {R(i == src ? 1 : 0), R(0)};
        bool used = false;
        C2<R> piv = {R(1), R(0)};
        int mycol = i;
        auto step = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int ce = c >> 2, cq = c & 3;
            constexpr int e0 = (c + 1) >> 2;               // elements below e0 hold only eliminated columns
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
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{});
        step(std::integral_constant<int, 7>{});
        step(std::integral_constant<int, 8>{});
        step(std::integral_constant<int, 9>{});
        step(std::integral_constant<int, 10>{});
        step(std::integral_constant<int, 11>{});
        step(std::integral_constant<int, 12>{});
        step(std::integral_constant<int, 13>{});
        step(std::integral_constant<int, 14>{});
        step(std::integral_constant<int, 15>{});
        // w[c] = rhs / pivot on the row that pivoted column c
        wave_lds_sync();
        if (q == 0) s.w[mycol] = cmul(rhs, cinv(piv));
        wave_lds_sync();
        C2<R> wi = s.w[i], wc[4];
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
        if (i == src) {
#pragma unroll
            for (int e = 0; e < 4; ++e) B[e] = {wc[e].re * sc, -wc[e].im * sc};
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (in[e]) store_what<R>(a, ((size_t)f * M + 4 * e + q) * M + i, B[e].re, -B[e].im);
}

}  // namespace

hipError_t launch_update_lds16(hipStream_t s, const UpdateArgs& a) {
    dim3 grid(a.F);
    if (a.K == a.M && !a.init_only) {                     // determined: one wavefront per bin
        auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, dim3(64), 0, s, a); };
        if (a.use_double) {
            if (a.vpart_f64) go(update_det16_kernel<double, double>);
            else go(update_det16_kernel<double, float>);
        } else {
            if (a.vpart_f64) go(update_det16_kernel<float, double>);
            else go(update_det16_kernel<float, float>);
        }
        return hipGetLastError();
    }
    if (a.use_double)
        hipLaunchKernelGGL((update_lds16_kernel<double>), grid, dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL((update_lds16_kernel<float>), grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

}  // namespace oiva
