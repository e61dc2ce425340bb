// Weighted covariance + per-bin update in ONE kernel per bin batch          reference overiva.py:158-190
//
//   V_k[f] = sum_t w_k[t] x_{t,f} x_{t,f}^H  (:179, both sources in one pass over X, gamma normalisation and floor :158-173 on
//   the fly), then for the same bins: W /= gamma (:163 / :167), w_k <- (W_hat^H V_k)^-1 e_k, w_k /= sqrt(w_k^H V_k w_k) (:181-186),
//   J from the orthogonality constraint (:189-190)
//
// -- the fusion BASELINE.json's north_star names, at the headline shape (8 channels, 2 sources, 2048 bins x 4000 frames), where the
// four-launch iteration runs cov_dma_kernel and update_bg_kernel as two launches with 8 MB of float64 partials between them.
// MEASURED (round 5, 2048 x 4000 x 8 / 2, `mixed`): this kernel 113-114 us per launch, 537 MB of HBM traffic (1.00 x); the two
// launches 97.6-98.6 + 12.8-14.9 us: the iteration 204.7 against 200.7 us on one box, 205.8 against 208.3 on another -- no
// gain: the accumulation runs ~6 % slower here (a fifth request and LDS read per step for the activations, which
// cov_dma_kernel takes from scalar loads), and the reduction and the 6 us chain of the update end every workgroup at the same
// moment, as the update kernel's did.  It is therefore NOT the default (oiva_plan_set_fuse_cov_update(p, 1) or
// $OIVA_COV_UPDATE=1 switch it on; same bits either way), and the covariance pass that bench.py's roofline is about stays
// cov_dma_kernel.
//
// Geometry.  cov_dma_kernel's workgroup is 16 bins x one of FOUR frame splits, so a bin's covariance is spread over four
// workgroups and the update has to wait for all of them (another launch, or a wait between workgroups).  Here a workgroup is
// 4 bins x ALL frames: wave c takes the frames of split c, its 64 lanes are 4 bins x 16 frame phases, and lane (bin, q) of wave
// c runs exactly the float32 chain that lane (bin, q) of split c runs in cov_dma_kernel (frames c tc + q + 16 i, the same
// packed arithmetic, PkAcc2).  The 16 phases of a split are added in float64 in the same order, the four splits in the order
// update_bg_kernel adds them: V -- and therefore W -- has THE SAME BITS as the two-launch path (tested), the partials never
// leave LDS, and nothing waits on another workgroup.  Wave w then updates bin w of the four (update_chain.h, one wavefront per
// bin as in update_bg_kernel).
//
// Memory.  Per step a wave requests 16 frames x (4 bins x 64 bytes) by four LDS-DMA instructions -- 32 cache lines per
// instruction, as many as cov_dma_kernel's 4 frames x 1 KB; the workgroups of neighbouring bin quads walk the same frames, so
// the chip streams whole rows of X -- and the 32 activations r[t, k] of those frames by a fifth; 4-stage ring per wave,
// counted vmcnt waits, the LDS reads in assembly (hipcc drains the DMA queue in front of any LDS read it can see).
#include <cstdint>
#include <cstdlib>

#include "cov_arith.h"
#include "oiva_device.h"
#include "update_chain.h"

namespace oiva {
namespace {

constexpr int kCuBins = 4;                               // bins per workgroup = waves (one per bin in the update)
constexpr int kCuStages = 4;
constexpr int kCuX = 64 * 64;                            // bytes of X per stage per wave: 64 lanes x 8 channels x 8 bytes
constexpr int kCuStage = kCuX + 256;                     // + the stage's activations: 16 frames x 2 sources x 4 bytes (+ padding lanes)
constexpr int kCuChunk = 16;
constexpr int kCuLdsStride = kBlock + 1;
constexpr int kCuRing = kWaves * kCuStages * kCuStage;   // 69 632 bytes
constexpr int kCuPartOff = 32768;                        // float64 partials [split][bin][2 * 64] behind the reduction scratch
static_assert(kCuPartOff >= (int)sizeof(float) * kCuChunk * kCuLdsStride, "partials overlap the reduction scratch");
static_assert(kCuPartOff + (int)sizeof(double) * kWaves * kCuBins * 128 <= kCuRing, "partials outside the ring");

typedef __attribute__((address_space(1))) const void gvoid_cu_t;
typedef __attribute__((address_space(3))) void lvoid_cu_t;

template <typename R>
__global__ __launch_bounds__(kBlock, 2) void cov_update_kernel(const float2* __restrict__ X, const float* __restrict__ Rv,
                                                               float* __restrict__ wscale, int model, UpdateArgs a, int tc) {
    constexpr int M = 8, K = 2, NA = M * M;
    constexpr int PIECES = M / 2;
    __shared__ __attribute__((aligned(16))) unsigned char ring[kCuRing];

    const int T = a.T, F = a.F;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = frame split in the covariance phase, = bin in the update
    const int b = lane & (kCuBins - 1);
    const int q = lane >> 2;                                         // frame phase 0..15
    const int f0 = blockIdx.x * kCuBins;
    const int f = f0 + b;
    const int fc = f < F ? f : F - 1;
    const int t_begin = wave * tc;
    const int t_end = min(T, t_begin + tc);
    const int nsteps = (t_end - t_begin) >> 4;                      // whole steps (cov_update_supported)

    PkAcc2<M> pacc;
    pacc.clear();

    unsigned char* wring = ring + wave * (kCuStages * kCuStage);                       // wave-uniform
    const unsigned rd_base = (unsigned)(uintptr_t)wring + (unsigned)lane * 16u;        // this lane's 16-byte slot of a piece
    const unsigned rr_base = (unsigned)(uintptr_t)wring + (unsigned)kCuX + (unsigned)q * 8u;   // (r_0, r_1) of this lane's frame
    const size_t frame_stride = (size_t)F * M;
    // Addresses as a wave-uniform base (the first frame of the step: scalar registers) + a 32-bit lane offset that never changes
    // (frame phase and bin: < 16 rows of X): per-lane 64-bit pointers and their induction variables did not fit beside the 128
    // accumulators (172 bytes of scratch per lane).  Every split is a whole number of 16-frame steps (cov_update_supported), so
    // no lane of a consumed step is past its split; steps past the end (requested to keep vmcnt counting, never consumed)
    // re-read the split's first frames.
    const unsigned xoff = (unsigned)(((size_t)q * F + fc) * M * sizeof(float2));       // bytes
    const unsigned roff = (unsigned)(lane < 32 ? lane : 31) * 4u;                      // r[t0 + (l >> 1)][l & 1], K = 2: float l of the step's rows
    // (buffer form of the LDS-DMA: descriptor and step offset in scalar registers, ONE vector register of address for all
    //  five requests; X < 2 GB by cov_update_supported)
    const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(X), 0, (int)((size_t)T * frame_stride * sizeof(float2) < 0x7fffffffu ? (size_t)T * frame_stride * sizeof(float2) : 0x7fffffffu), 0x00020000);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Rv), 0, (int)(((size_t)T + 16) * K * sizeof(float)), 0x00020000);

    // issue(i, s): the four pieces of the samples of step i into stage s, and the activations of step i + 1 into stage (s + 1) & 3
    // -- one step AHEAD of the samples, so that the weights of step i + 1 are formed (a correctly rounded division, as in
    // cov_dma_kernel: same bits) while the products of step i retire, with the sample registers free, and a step costs one LDS
    // round trip, not two.
    auto issue_r = [&](int i, int s) {
        const int t0 = t_begin + 16 * (i < nsteps ? i : 0);                            // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lvoid_cu_t*)(wring + s * kCuStage + kCuX), 4, (int)roff, (int)((size_t)t0 * K * sizeof(float)), 0, 0);
    };
    auto issue = [&](int i, int s) {
        const int t0 = t_begin + 16 * (i < nsteps ? i : 0);                            // wave-uniform
        const unsigned xstep = (unsigned)((size_t)t0 * frame_stride * sizeof(float2));
        // (the piece's 16 bytes go into the SCALAR offset: the instruction's immediate offset would move the LDS address too)
        static_for<PIECES>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lvoid_cu_t*)(wring + s * kCuStage + j * 1024), 16, (int)xoff, (int)(xstep + j * 16), 0, 0);
        });
        issue_r(i + 1, (s + 1) & 3);
    };
    float ginv[K];
    v2f rn;                              // activations of the step about to be consumed (read one step ahead)
    auto consume = [&](int s) {
        // the weights first -- with the sample registers of the previous step dead: the division beside them and the 128
        // accumulators did not fit -- (cov_dma_kernel multiplies by live = 1.f for a frame inside its split: the same bits)
        v2f w = {activation_weight(rn.x, ginv[0]), activation_weight(rn.y, ginv[1])};
        asm volatile("" : "+v"(w));
        float4 v[PIECES];
        asm volatile(
            "s_waitcnt vmcnt(15)\n\t"
            "ds_read_b128 %0, %5\n\t"
            "ds_read_b128 %1, %5 offset:1024\n\t"
            "ds_read_b128 %2, %5 offset:2048\n\t"
            "ds_read_b128 %3, %5 offset:3072\n\t"
            "ds_read_b64 %4, %6\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(rn)
            : "v"(rd_base + (unsigned)(s * kCuStage)), "v"(rr_base + (unsigned)(((s + 1) & 3) * kCuStage))
            : "memory");
        v2f x[M];
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            x[2 * j] = v2f{v[j].x, v[j].y};
            x[2 * j + 1] = v2f{v[j].z, v[j].w};
        }
        pacc.add(x, w);
    };

    issue_r(0, 0);
    issue(0, 0);
    issue(1, 1);
    issue(2, 2);
    // scale normalisation of the activations (overiva.py:158-159) while the first three steps are on their way
    float ws[K];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
        const float gamma = (float)gamma_of(Rv, T, K, kk);
        ginv[kk] = 1.f / gamma;
        ws[kk] = model == OIVA_MODEL_LAPLACE ? gamma : sqrtf(gamma);                   // overiva.py:163 / :167
        if (blockIdx.x == 0 && tid == 0 && wscale != nullptr) wscale[kk] = ws[kk];
    }
    // activations of step 0: the oldest request
    asm volatile("s_waitcnt vmcnt(15)\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(rn) : "v"(rr_base) : "memory");

    int i = 0;
    for (; i + 4 <= nsteps; i += 4) {       // stage indices are compile-time constants in the unrolled body
        issue(i + 3, 3); consume(0);
        issue(i + 4, 0); consume(1);
        issue(i + 5, 1); consume(2);
        issue(i + 6, 2); consume(3);
    }
    if (i < nsteps) { issue(i + 3, 3); consume(0); }
    if (i + 1 < nsteps) { issue(i + 4, 0); consume(1); }
    if (i + 2 < nsteps) { issue(i + 5, 1); consume(2); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the DMA queue before the ring becomes scratch

    // ---- the 16 phases of every split in float64 (order of cov_dma_kernel's reduce_and_store_at), kept in LDS:
    //      part[split][bin][k * 64 + packed entry]
    float* lds = reinterpret_cast<float*>(ring);
    double* part = reinterpret_cast<double*>(ring + kCuPartOff);
    {
        // (index math of the phases below from OPAQUE copies of the lane number: whatever the compiler could derive from `lane`
        //  ahead of time -- addresses, element positions -- it would keep in registers across the accumulation, where the 128
        //  accumulators leave none to spare)
        int tl = lane;
        asm volatile("" : "+v"(tl));
        const int c = wave, bb = (tl >> 4) & 3, aa = tl & 15;
#pragma unroll
        for (int r0 = 0; r0 < K * NA; r0 += kCuChunk) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < kCuChunk; ++e) lds[e * kCuLdsStride + tid] = pacc.at(r0 + e);
            __syncthreads();
            double s = 0.;
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) s += (double)lds[aa * kCuLdsStride + c * 64 + qq * 4 + bb];
            part[(c * kCuBins + bb) * (K * NA) + r0 + aa] = s;
        }
    }
    __syncthreads();

    // ---- per-bin update (update_bg_kernel): wave w = bin f0 + w, lane (i, j) = element [i][j]
    int ul = lane;
    asm volatile("" : "+v"(ul));
    const Sq<8, R> sq(ul);
    const int ui = sq.i, uj = sq.j;
    const int fu_raw = f0 + wave;
    const bool fvalid = fu_raw < F;
    const int fu = fvalid ? fu_raw : F - 1;
    Cx<R> B;
    {
        R vr, vi;
        load_what<R>(a, ((size_t)fu * M + uj) * M + ui, vr, vi);
        B = {vr, -vi};
    }
    if (ui < K) {                            // overiva.py:163 / :167
        const R sc = R(1) / R(ui == 0 ? ws[0] : ws[1]);
        B.re *= sc;
        B.im *= sc;
    }
    int off = 0;
    float sgn = 0.f;
    herm_offsets(M, ui, uj, off, sgn);
    Cx<R> C = {R(0), R(0)};
    {
        const double* pc = a.Cx + (size_t)fu * NA + off;
        C.re = R(pc[0]);
        if (sgn != 0.f) C.im = R(sgn * pc[1]);
    }
    const R invT = R(1) / R(T);
    Cx<R> V[K];
    const int nsplit = (T + tc - 1) / tc;
#pragma unroll
    for (int s = 0; s < K; ++s) {
        // the splits in order, as sum_vpart adds them (update_bg_kernel)
        double sr = 0., si = 0.;
        for (int c = 0; c < nsplit; ++c) {
            const double* pp = part + (c * kCuBins + wave) * (K * NA) + s * NA + off;
            sr += pp[0];
            si += pp[1];
        }
        if (sgn == 0.f) si = 0.;
        V[s] = {R(sr) * invT, R(si) * R(sgn) * invT};
    }
    bg_chain<8, R, K>(sq, B, C, V, M);
    if (fvalid) store_what<R>(a, ((size_t)fu * M + uj) * M + ui, B.re, -B.im);
}

}  // namespace

// eligible: 8 channels, 2 sources, four frame splits of a whole number of 16-frame steps (what the plan chooses at the headline
// shape)
bool cov_update_supported(int M, int K, int T, int F, int nsplit, int tc) {
    // (32-bit buffer offsets; whole steps: every split, and with it T, a multiple of 16 frames)
    return M == 8 && K == 2 && nsplit == kWaves && tc % 16 == 0 && T % 16 == 0 && T > (kWaves - 1) * tc && F >= kCuBins &&
           (size_t)T * F * M * 8 < ((size_t)1 << 31);       // (buffer addressing of X: 32-bit offsets)
}

hipError_t launch_cov_update(hipStream_t s, const float2* X, const float* R, float* wscale, int model, const UpdateArgs& a, int tc) {
    const dim3 grid((unsigned)((a.F + kCuBins - 1) / kCuBins));
    if (a.use_double)
        return launch_dominant(cov_update_kernel<double>, grid, dim3(kBlock), 0, s, X, R, wscale, model, a, tc);
    return launch_dominant(cov_update_kernel<float>, grid, dim3(kBlock), 0, s, X, R, wscale, model, a, tc);
}

}  // namespace oiva
