// Per-lane arithmetic of the weighted covariance pass (reference overiva.py:179), shared by the streaming kernels
// (kernels_cov.hip) and the X-resident iteration kernel (kernels_resident.hip).
#pragma once
#include "oiva_device.h"

namespace oiva {

// acc[kk][*] += w[kk] * pack(x x^H); ACC = float, or double for the float64 accumulation mode (products of
// float32 data are then exact, as in the reference's complex128 product at overiva.py:179)
template <int M, int KC, bool UNIT, typename ACC>
__device__ __forceinline__ void accumulate(ACC (&acc)[KC][M * M], const ACC (&xr)[M], const ACC (&xi)[M],
                                           const ACC (&w)[KC]) {
    if constexpr (KC == 1) {
        // one source: scale x once, then 4 FMAs per complex pair
        ACC sr[M], si[M];
#pragma unroll
        for (int c = 0; c < M; ++c) {
            sr[c] = xr[c] * w[0];   // UNIT: w is 1 (live frame) or 0 (clamped tail frame)
            si[c] = xi[c] * w[0];
        }
#pragma unroll
        for (int c = 0; c < M; ++c) acc[0][c] = fma(sr[c], xr[c], fma(si[c], xi[c], acc[0][c]));
        int a = M;
#pragma unroll
        for (int c = 0; c < M; ++c) {
#pragma unroll
            for (int d = c + 1; d < M; ++d) {
                acc[0][a] = fma(sr[c], xr[d], fma(si[c], xi[d], acc[0][a]));           // Re x_c conj(x_d)
                acc[0][a + 1] = fma(si[c], xr[d], fma(-sr[c], xi[d], acc[0][a + 1]));  // Im x_c conj(x_d)
                a += 2;
            }
        }
    } else {
        // several sources: form each product once, one FMA per source
#pragma unroll
        for (int c = 0; c < M; ++c) {
            const ACC p = fma(xr[c], xr[c], xi[c] * xi[c]);
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) acc[kk][c] = fma(w[kk], p, acc[kk][c]);
        }
        int a = M;
#pragma unroll
        for (int c = 0; c < M; ++c) {
#pragma unroll
            for (int d = c + 1; d < M; ++d) {
                const ACC pre = fma(xr[c], xr[d], xi[c] * xi[d]);
                const ACC pim = fma(xi[c], xr[d], -(xr[c] * xi[d]));
#pragma unroll
                for (int kk = 0; kk < KC; ++kk) {
                    acc[kk][a] = fma(w[kk], pre, acc[kk][a]);
                    acc[kk][a + 1] = fma(w[kk], pim, acc[kk][a + 1]);
                }
                a += 2;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Packed-fp32 arithmetic for two sources on natural (re, im) register pairs.  hipcc's own code for the generic
// accumulate() spends a third of its VALU instructions on moves that assemble operand pairs (measured in the
// ISA: 102 v_mov next to 146 v_pk_* per frame); VOP3P's op_sel / op_sel_hi / neg_hi modifiers make them
// unnecessary: every operand is a pair exactly as ds_read_b128 delivered it.
//   x_c conj(x_d) = (xr_c xr_d + xi_c xi_d,  xi_c xr_d - xr_c xi_d)
//     p  = (xr_c * xr_d, -(xr_c * xi_d))            v_pk_mul  src0 lo broadcast, neg_hi on src1
//     p += (xi_c * xi_d,   xi_c * xr_d)             v_pk_fma  src0 hi broadcast, src1 halves swapped
//     V_k += w_k * p                                 v_pk_fma  w = (w_0, w_1) pair, lo | hi broadcast
// ---------------------------------------------------------------------------------------------
using v2f = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ v2f pk_mul_lo_negim(v2f a, v2f b) {          // (a.x * b.x, -(a.x * b.y))
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_fma_hi_swap(v2f a, v2f b, v2f c) {    // (c.x + a.y * b.y, c.y + a.y * b.x)
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ v2f pk_fma_w0(v2f w, v2f p, v2f c) {         // c + w.x * p
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(w), "v"(p), "v"(c));
    return r;
}
__device__ __forceinline__ v2f pk_fma_wy_rot(v2f w, v2f x, v2f c) {     // (c.x - w.y * x.y, c.y + w.y * x.x): c + i w.y x
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(w), "v"(x), "v"(c));
    return r;
}
__device__ __forceinline__ v2f pk_fma_w1(v2f w, v2f p, v2f c) {         // c + w.y * p
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(w), "v"(p), "v"(c));
    return r;
}

// The same packed forms as VOLATILE asm with the accumulator tied to the result: the compiler keeps volatile asm in source
// order (it reorders plain asm statements freely, which in the kernels that carry 128 accumulators next to 28 operand
// registers stretched the live ranges of the products past what 256 registers hold).
__device__ __forceinline__ v2f qk_mul_lo_negim(v2f a, v2f b) {          // (a.x * b.x, -(a.x * b.y))
    v2f r;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void qk_fma_hi_swap(v2f a, v2f b, v2f& c) {   // c += (a.y * b.y, a.y * b.x)
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void qk_fma_w0(v2f w, v2f p, v2f& c) {        // c += w.x * p
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(c) : "s"(w), "v"(p));
}
__device__ __forceinline__ void qk_fma_w1(v2f w, v2f p, v2f& c) {        // c += w.y * p
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(c) : "s"(w), "v"(p));
}

// 8 x 8 Hermitian half split over the two lanes of a pair (kernels_cov_pair32.hip, kernels_cov_pair64.hip): channel groups
// A = 0..3, B = 4..7; lane j holds the diagonal block of group j (sums 0..3: the real diagonals; then (re, im) of its 6
// entries r < c) and rows 2j, 2j+1 of A x B (8 entries) = 32 real sums.  Sum a of lane j -> position in the packed layout.
__device__ __forceinline__ int pair_position(int j, int a) {
    if (a < 4) return 4 * j + a;
    const int p = (a - 4) >> 1, im = (a - 4) & 1;
    int c, d;
    if (p < 6) {
        const int r = p < 3 ? 0 : (p < 5 ? 1 : 2);
        const int cc = p < 3 ? p + 1 : (p < 5 ? p - 1 : 3);
        c = 4 * j + r;
        d = 4 * j + cc;
    } else {
        c = 2 * j + ((p - 6) >> 2);
        d = 4 + ((p - 6) & 3);
    }
    return herm_pair_index(8, c, d) + im;
}

// accumulators of TWO sources in that layout: off-diagonal entries as (re, im) pairs, diagonals as scalars
template <int M>
struct PkAcc2 {
    static constexpr int NP = M * (M - 1) / 2;
    v2f pair[2][NP];
    float diag[2][M];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int i = 0; i < NP; ++i) pair[k][i] = v2f{0.f, 0.f};
#pragma unroll
            for (int c = 0; c < M; ++c) diag[k][c] = 0.f;
        }
    }
    // x[c] = (re, im) of channel c; w = (w_0, w_1)
    __device__ __forceinline__ void add(const v2f (&x)[M], v2f w) {
#pragma unroll
        for (int c = 0; c < M; ++c) {
            const v2f sq = x[c] * x[c];
            const float p = sq.x + sq.y;
            diag[0][c] = fmaf(w.x, p, diag[0][c]);
            diag[1][c] = fmaf(w.y, p, diag[1][c]);
        }
        // The compiler keeps inline asm in source order and knows no latencies, so the order written here is the issue order:
        // independent instructions back to back, a dependent one as far behind its producer as the registers allow.  Rows c
        // and M-2-c of the upper triangle together hold M pairs (M-1-c and c+1), so taking the rows in such couples gives
        // groups of M independent instructions (the middle row, M/2, stands alone) where row by row the groups shrink to 1 --
        // a wave that is alone on its SIMD (the X-resident kernel) stalls on every short group, and hipcc pads each
        // back-to-back dependent pair of asm statements with an s_nop.  Per accumulator the arithmetic is unchanged.
        constexpr int NG = (M - 1 + 1) / 2;                      // couples of rows (+ the middle row when M - 1 is odd)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int c1 = M - 2 - g;                            // partner row; == g for the middle row
            const int n0 = M - 1 - g;                            // pairs of row g
            const int n = c1 > g ? n0 + (g + 1) : n0;            // members of the group
            v2f p[M];
            // member e of the group -> (row c, column d, index of the pair)
#define OIVA_MEMBER(e)                                                                     \
            const int c = (e) < n0 ? g : c1;                                               \
            const int d = (e) < n0 ? g + 1 + (e) : c1 + 1 + ((e) - n0);                    \
            const int ip = c * (M - 1) - (c * (c - 1)) / 2 + (d - c - 1);
#pragma unroll
            for (int e = 0; e < M; ++e) {
                if (e < n) {
                    OIVA_MEMBER(e)
                    (void)ip;
                    p[e] = pk_mul_lo_negim(x[c], x[d]);
                }
            }
#pragma unroll
            for (int e = 0; e < M; ++e) {
                if (e < n) {
                    OIVA_MEMBER(e)
                    (void)ip;
                    p[e] = pk_fma_hi_swap(x[c], x[d], p[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < M; ++e) {
                if (e < n) {
                    OIVA_MEMBER(e)
                    (void)d;
                    pair[0][ip] = pk_fma_w0(w, p[e], pair[0][ip]);
                }
            }
#pragma unroll
            for (int e = 0; e < M; ++e) {
                if (e < n) {
                    OIVA_MEMBER(e)
                    (void)d;
                    pair[1][ip] = pk_fma_w1(w, p[e], pair[1][ip]);
                }
            }
#undef OIVA_MEMBER
        }
    }
    // packed Hermitian layout (herm_pair_index): accumulator e = k * M*M + a
    __device__ __forceinline__ float at(int e) const {
        const int k = e / (M * M), a = e % (M * M);
        if (a < M) return diag[k][a];
        return ((a - M) & 1) ? pair[k][(a - M) >> 1].y : pair[k][(a - M) >> 1].x;
    }
};

}  // namespace oiva
