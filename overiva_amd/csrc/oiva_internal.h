// Internal declarations shared by the translation units of liboveriva_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <string>

#include "overiva_hip.h"

namespace oiva {

// records the thread-local message oiva_last_error() returns and hands back `code` (defined in plan.hip)
int fail_with(int code, const std::string& msg);
// exchange.hip: rank / world / slot size and every rank's gather buffer as mapped in this process; -1 unless connected
int xchg_peers(oiva_xchg* x, char** peers, int* rank, int* world, size_t* slot_bytes);

// ---- lane geometry shared by the streaming kernels -------------------------------------------
// A wave is 16 bins x 4 frame phases: lane l -> bin (l & 15), phase (l >> 4).  The 16 bins of one
// frame are 16*M*8 contiguous bytes of the native (T, F, M) complex64 tensor, so one wave touches
// four contiguous runs per step and every byte of every cache line it opens is consumed.
constexpr int kBinsPerWave = 16;
constexpr int kPhasesPerWave = 4;
constexpr int kBlock = 256;  // 4 waves
constexpr int kWaves = kBlock / 64;

// packed Hermitian layout of one M x M covariance: M real diagonals, then for every c < d
// (row-major) the pair (re, im) of V[c][d] = sum w * x_c * conj(x_d).  M*M floats in total.
__host__ __device__ inline int herm_pair_index(int M, int c, int d) {  // c < d
    return M + 2 * (c * M - (c * (c + 1)) / 2 + (d - c - 1));
}

// Layout of the activation buffer R: (T, K) float32 activations r, kPhasesPerWave zeroed pad rows (kernels read
// whole groups of 4 frames), then -- 8-byte aligned -- one float64 partial sum of r per (block of kBlock frames,
// source), written by the activation kernel, from which every consumer derives gamma_k = mean_t r[t,k]
// (overiva.py:158) in the same fixed order.
__host__ __device__ inline size_t rsum_offset_floats(int T, int K) { return (((size_t)T + kPhasesPerWave) * K + 1) & ~(size_t)1; }
__host__ __device__ inline int rsum_blocks(int T) { return (T + kBlock - 1) / kBlock; }
__host__ __device__ inline size_t r_buffer_bytes(int T, int K) {
    return rsum_offset_floats(T, K) * sizeof(float) + (size_t)rsum_blocks(T) * K * sizeof(double);
}

// Launch of the dominant (covariance) kernel.  When the calling thread has armed a pair of events (arm_kernel_timer), the
// kernel is launched with them attached to its own dispatch (hipExtLaunchKernelGGL): their elapsed time is the kernel's
// duration as the profiler reports it, without the ~3 us of a separate event record in front and behind.
struct KernelTimer {
    hipEvent_t start = nullptr, stop = nullptr;
};
KernelTimer& kernel_timer();
inline void arm_kernel_timer(hipEvent_t start, hipEvent_t stop) {
    kernel_timer().start = start;
    kernel_timer().stop = stop;
}
template <typename Kern, typename... Args>
hipError_t launch_dominant(Kern kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t s, Args... args) {
    KernelTimer& t = kernel_timer();
    if (t.start != nullptr) {
        hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)shmem, s, t.start, t.stop, 0, args...);
        t.start = t.stop = nullptr;
    } else {
        hipLaunchKernelGGL(kernel, grid, block, shmem, s, args...);
    }
    return hipGetLastError();
}

struct CovGeom {
    int nsplit;   // frame splits (grid.y)
    int tc;       // frames per split (multiple of 16)
    int kc;       // sources per pass (template KC)
    int nbg;      // bin groups of 16 (grid.x)
    int hmfma = 0;  // with half16, 9..16 sources: the sources on the fp32 matrix cores (kernels_cov_hmfma.hip)
    int half16 = 0; // 10/12/14/16 channels (9..15 odd: padded copy): kernels_cov_half16.hip (2 bins per workgroup; float32: 5..16 sources, all per pass; float64: 3..16 sources, 4 or 8 per pass)
    int pair32 = 0; // 8 channels, >= 3 sources, float32: kernels_cov_pair32.hip (32 bins per workgroup, four sources per pass)
    int pad = 0;  // odd channel count on the vector-ALU kernels: they read the copy of X padded to M + 1 channels
    int quad = 0; // 10/12/14/16 channels, few sources, float32: the vector-ALU kernel of kernels_cov_quad.hip (float64 partials)
    int part32 = 0; // hmfma, float32 arithmetic: the partial blocks leave as float32 (each the float64 sum of its chains, rounded once)
};
struct PowGeom {
    int nb;       // bin batches of 64 (grid.x)
    int nsplit;   // frame splits (grid.y)
    int tcp;      // frames per split (multiple of 4, <= kPowMaxFrames)
    int kp;       // sources per pass
};
constexpr int kPowMaxFrames = 512;

// ---- launchers (one per kernel family; each .hip file owns its template instantiations) -------
// Weighted covariance pass, overiva.py:179 (and :87 with unit weights).
//   X (T,F,M) c64; R (T,K) f32 activations r (unnormalised) or nullptr for unit weights (then K must be 1)
//   weights: w[t,k] = 1 / max(R[t,k] / gamma_k, eps), gamma_k = mean_t R (overiva.py:158-173); with raw != 0
//   gamma is taken as 1.  wscale (K): out, gamma (laplace) | sqrt(gamma) (gauss), written by one workgroup.
//   Vpart [nsplit][F][K][M*M] packed partial sums (NOT divided by T), float32 or (f64 != 0) float64
//   f64: accumulate in float64 (the reference's arithmetic: its float64 r_inv promotes overiva.py:179 to complex128)
// Xpad: (T, F, M + 1) copy of X with one zero channel behind every bin's M (odd 9..15 channels; launch_pad_channels), read by
// the vector-ALU kernels when g.pad is set; else nullptr
hipError_t launch_cov(hipStream_t s, const float2* X, const float2* Xpad, const float* R, float* Wt, float* wscale, int model, int raw,
                      void* Vpart, bool f64, int T, int F, int M, int K, const CovGeom& g);
hipError_t launch_pad_channels(hipStream_t s, const float2* X, float2* Xpad, long long n_tf, int M);
// planar matrix-core kernel for 9..16 channels (grid = F bins x nsplit, tc frames per split, tc multiple of 4)
//   Wt (T,16): scratch for the final weights (written by a small pre-pass)
hipError_t launch_cov_mfma(hipStream_t s, const float2* X, const float* R, float* Wt, float* wscale, int model, int raw,
                           void* Vpart, bool f64, int T, int F, int M, int K, int nsplit, int tc);
// vector-ALU kernel for 10, 12, 14, 16 channels and K <= 4 sources, float32 arithmetic (kernels_cov_quad.hip): the Hermitian
// half split over four lanes per (bin, frame); tc multiple of 8; Vpart float64; R == nullptr: unit weights (K = 1);
// Wt (T,16): scratch for the final weights, as for launch_cov_mfma
bool cov_quad_supported(int M, int K);
int cov_quad_sources_per_pass(int K);
hipError_t launch_cov_quad(hipStream_t s, const float2* X, const float* R, float* Wt, float* wscale, int model, int raw,
                           double* Vpart, int T, int F, int M, int Mv, int K, const CovGeom& g);
// pre-pass of the 9..16-channel kernels: Wt (T, Kp) = 1 / max(r / gamma, eps), columns >= K zero; writes wscale (K)
hipError_t launch_cov_weights(hipStream_t s, const float* R, float* Wt, float* wscale, int model, int raw, int T, int K, int Kp);
// float64 vector-ALU kernel for 8 channels (kernels_cov_pair64.hip): the Hermitian half split over two lanes per (bin, frame),
// 32 bins per workgroup (grid.x = ceil(F / 32)), tc multiple of 8, Vpart float64; Wt: the (T, 16) float scratch, used as
// (T, 8) doubles; R == nullptr: unit weights (K = 1)
bool cov_pair64_supported(int M);
int cov_pair64_sources_per_pass(int K);
int cov_pair64_bins_per_block();
hipError_t launch_cov_pair64(hipStream_t s, const float2* X, const float* R, float* Wt, float* wscale, int model, int raw,
                             double* Vpart, int T, int F, int M, int K, const CovGeom& g);
// float32 kernel for 8 channels and three or more sources, FOUR per pass over X (kernels_cov_pair32.hip): the Hermitian half
// over two lanes per (bin, frame), 32 bins per workgroup, tc multiple of 8, Vpart float64; Wt: (T, 16) scratch as for
// launch_cov_mfma; R == nullptr: unit weights (K = 1) on the same geometry
bool cov_pair32_supported(int M, int K);
int cov_pair32_sources_per_pass();
int cov_pair32_bins_per_block();
hipError_t launch_cov_pair32(hipStream_t s, const float2* X, const float* R, float* Wt, float* wscale, int model, int raw,
                             double* Vpart, int T, int F, int M, int K, const CovGeom& g);
// float32 vector-ALU kernel for 10, 12, 14, 16 channels and up to 16 sources in ONE pass (kernels_cov_half16.hip): the
// Hermitian half over 32 lanes per (bin, frame), 2 bins per workgroup (grid.x = ceil(F / 2)), tc multiple of 16, Vpart
// float64; Wt: (T + 1, 16) scratch (row T is zeroed by the launcher); R == nullptr: unit weights (K = 1)
bool cov_half16_supported(int M, int K);
int cov_half16_sources_per_pass(int K);
hipError_t launch_cov_half16(hipStream_t s, const float2* X, const float* R, float* Wt, float* wscale, int model, int raw,
                             double* Vpart, int T, int F, int M, int Mv, int K, const CovGeom& g);
// the same decomposition with float64 sums (the `precise` arithmetic), 3..16 sources, 4 or 8 per pass; Wt: (T + 1, 16) DOUBLES
// 9..16 sources: the sources as rows of the fp32 matrix-core instruction, Hermitian products on the vector ALU (kernels_cov_hmfma.hip)
bool cov_hmfma_supported(int M, int K);
hipError_t launch_cov_hmfma(hipStream_t s, const float2* X, const float* Wt, double* Vpart, int T, int F, int M, int Mv, int K, const CovGeom& g);
bool cov_hmfma64_supported(int M, int K);
hipError_t launch_cov_hmfma64(hipStream_t s, const float2* X, const double* Wt, double* Vpart, int T, int F, int M, int Mv, int K, const CovGeom& g);
bool cov_half16_f64_supported(int M, int K);
int cov_half16_f64_sources_per_pass(int K);
hipError_t launch_cov_half16_f64(hipStream_t s, const float2* X, const float* R, float* Wt, float* wscale, int model, int raw,
                                 double* Vpart, int T, int F, int M, int Mv, int K, const CovGeom& g);
int cov_sources_per_pass(int M, int K, bool f64, bool short_axis = false);
hipError_t cov_blocks_per_cu(int M, int kc, bool f64, int* n);
bool cov_supported(int M);

// Demix + source power, overiva.py:140 + the norms at :153/:155.
//   What (F,M,M) c64 row-major;  Ppart [nb][T][K]
//   Xpad: (T, F, M + 1) zero-padded copy of X (odd 9..15 channels, many sources) or nullptr
hipError_t launch_power(hipStream_t s, const float2* X, const float2* Xpad, const float2* What, float* Ppart, int T, int F, int M,
                        int K, const PowGeom& g);
// matrix-core variant for 9..16 channels (same Ppart layout, one pass over X for all sources)
hipError_t launch_power_mfma(hipStream_t s, const float2* X, const float2* What, float* Ppart, int T, int F, int M, int Mp, int K);
int pow_sources_per_pass(int M, int K);
hipError_t pow_blocks_per_cu(int M, int kp, int tcp, int* n);

// Source activation, overiva.py:152-155: parts [nparts][T][K] -> R (T,K) = 2 sqrt(p) | p / F_total.
// activation with the exchange of the ranks' partial powers inside it (bins sharded over GPUs; kernels_misc.hip)
constexpr int kCanonBlocks = 8;      // the sum over the 64-bin parts is associated in at most this many blocks (activation_kernel)
hipError_t launch_activation_xchg(hipStream_t s, const float* parts, int nparts, char* const* gath, int rank, int world, int loopback,
                                  int nblk_own, int nblk_peer, unsigned* epochs, unsigned* ctrl, long long timeout_ticks, float* R, int T, int K,
                                  int model, int F_total);
hipError_t launch_activation(hipStream_t s, const float* parts, int nparts, float* R, int T, int K, int model,
                             int F_total);
// fixed-order float64 sum of partial buffers (float32, or float64 when f64): out[e] = scale * sum_i parts[i][e]
hipError_t launch_sum_parts(hipStream_t s, const void* parts, bool f64, int nparts, double* out, long long n, double scale);

// Per-bin sequential update, overiva.py:181-190 (+ :161-167 W scaling, + :96-98 J init when init_only).
struct UpdateArgs {
    float2* What;         // (F,M,M) in/out (complex64: what the streaming kernels read)
    double2* What64;      // (F,M,M) complex128 copy carried between iterations by the float64 variants (or nullptr)
    const double* Cx;     // [F][M*M] packed, already divided by T
    const void* Vpart;    // [nsplit][F][K][M*M] packed partial sums, float32 or (vpart_f64) float64
    int vpart_f64;
    const float* wscale;  // (K) or nullptr
    int nsplit;
    int T, F, M, K;
    int init_only;        // 1: only (re)compute J from W and Cx
    int use_double;       // per-bin algebra in fp64
    int layout;           // 0: one lane per matrix element (M <= 8), 1: one lane per matrix row
};
// W_hat element idx: the float64 variants keep their own complex128 copy so that nothing is rounded to
// float32 between iterations; the complex64 array is always written (the streaming kernels read it)
template <typename R>
__device__ __forceinline__ void load_what(const UpdateArgs& a, size_t idx, R& re, R& im) {
    if (sizeof(R) == 8 && a.What64 != nullptr) {
        const double2 v = a.What64[idx];
        re = (R)v.x;
        im = (R)v.y;
    } else {
        const float2 v = a.What[idx];
        re = (R)v.x;
        im = (R)v.y;
    }
}
template <typename R>
__device__ __forceinline__ void store_what(const UpdateArgs& a, size_t idx, R re, R im) {
    a.What[idx] = make_float2((float)re, (float)im);
    if (sizeof(R) == 8 && a.What64 != nullptr) a.What64[idx] = make_double2((double)re, (double)im);
}
// one packed partial as float64 whatever its storage type
__device__ __forceinline__ double load_vpart(const void* base, int f64, size_t idx) {
    return f64 ? static_cast<const double*>(base)[idx] : (double)static_cast<const float*>(base)[idx];
}
// fixed-order float64 sum of the nsplit frame-split partials of one packed element and of its neighbour idx + 1
// (the imaginary part of an off-diagonal entry; a valid address for every element of a packed matrix with M > 1,
// discarded by the caller where it means nothing).  Branch-free: the loads of 16 splits are issued together
// (splits past nsplit re-read the last one and are masked), so a sum costs one memory round trip per 16 splits.
template <int kBatch, typename P>
__device__ __forceinline__ void sum_vpart_b(const P* __restrict__ base, size_t idx, size_t stride, int nsplit, double& sr,
                                            double& si) {
    sr = 0.;
    si = 0.;
    for (int s0 = 0; s0 < nsplit; s0 += kBatch) {
        P vr[kBatch], vi[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int sp = s0 + u < nsplit ? s0 + u : nsplit - 1;
            vr[u] = base[idx + (size_t)sp * stride];
            vi[u] = base[idx + (size_t)sp * stride + 1];
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const double m = s0 + u < nsplit ? 1. : 0.;
            sr += m * (double)vr[u];
            si += m * (double)vi[u];
        }
    }
}
// (round 5: the batch follows the number of splits -- 2, 4, 8 or 16 loads per part in flight: with the batch of 16 whatever nsplit,
//  the four splits of the headline shape cost 12 masked re-reads of the last split and 12 multiplications by zero per part and
//  source: update 11.9 -> 10.1 us there, 17.3 -> 12.8 at 2049 x 235 x 5 / 5, 26.8 -> 23.2 at 8 / 4; same sums in the same order)
template <typename P>
__device__ __forceinline__ void sum_vpart_t(const P* __restrict__ base, size_t idx, size_t stride, int nsplit, double& sr,
                                            double& si) {
    if (nsplit <= 2)          // (uniform)
        sum_vpart_b<2>(base, idx, stride, nsplit, sr, si);
    else if (nsplit <= 4)
        sum_vpart_b<4>(base, idx, stride, nsplit, sr, si);
    else if (nsplit <= 8)
        sum_vpart_b<8>(base, idx, stride, nsplit, sr, si);
    else
        sum_vpart_b<16>(base, idx, stride, nsplit, sr, si);
}
__device__ __forceinline__ void sum_vpart(const void* base, int f64, size_t idx, size_t stride, int nsplit, bool pair,
                                          double& sr, double& si) {
    if (f64)          // uniform
        sum_vpart_t(static_cast<const double*>(base), idx, stride, nsplit, sr, si);
    else
        sum_vpart_t(static_cast<const float*>(base), idx, stride, nsplit, sr, si);
    if (!pair) si = 0.;
}
hipError_t launch_update(hipStream_t s, const UpdateArgs& a);
// covariance + per-bin update of the same bins in ONE launch (kernels_cov_update.hip): 8 channels, 2 sources + background,
// float32 products; frames in four splits of tc (one per wave of a 4-bin workgroup).  Same bits as launch_cov followed by
// launch_update when the plan's covariance geometry is those four splits.  a: What, What64, Cx, T, F, use_double.
bool cov_update_supported(int M, int K, int T, int F, int nsplit, int tc);
hipError_t launch_cov_update(hipStream_t s, const float2* X, const float* R, float* wscale, int model, const UpdateArgs& a, int tc);
hipError_t launch_update_wave16(hipStream_t s, const UpdateArgs& a);   // 9..16 channels, one wavefront per bin
// determined float64 update of 9..16 channels, one matrix row per lane and four bins per wavefront (kernels_update16r.hip)
bool update_det16r_applies(const UpdateArgs& a);
hipError_t launch_update_det16r(hipStream_t s, const UpdateArgs& a);

// Epilogue, overiva.py:192-199.
//   stats: per-bin sums for projection back: [nsplit][F][K][3] = (Re num, Im num, den)
hipError_t launch_demix_stats(hipStream_t s, const float2* X, const float2* What, float* Spart, int T, int F, int M,
                              int K, const CovGeom& g);
//   write Y (T,F,K) c64, scaled by conj(z) when Spart != nullptr
hipError_t launch_demix_write(hipStream_t s, const float2* X, const float2* What, const float* Spart, int nsplit,
                              float2* Y, int T, int F, int M, int K);
// OGIVE (ive.py:33-256): per-bin state and kernels (kernels_ogive.hip); the streaming passes are the AuxIVA ones
struct OgiveState {
    const double* Cx;    // [F][M*M] packed Hermitian, / T
    double2* CxInv;      // (F, M, M) complex128
    double* CxNorm;      // (F) Frobenius norm of Cx
    double2* A;          // (F, M) mixing vector a
    double2* Delta;      // (F, M) last step of every bin
    double* Lambda;      // (F) lambda_a
    int* DoA;            // (F) mixing-vector step selected
    int* DoW;            // (F) demixing-vector step selected
    double* Dnorm;       // (F) ||delta_f||
    int* ctrl;           // [0] stopping rule met, [1] epochs run, [2] step-kernel workgroups done this epoch
    double* maxdelta;    // [0] max_f ||delta_f|| of the last epoch, [1] running max of the current one (bit pattern)
    float2* What;        // (F, M, M): column 0 = w, what the streaming kernels read
    double2* What64;     // complex128 copy
};
constexpr int kModelOgiveLaplace = 2;   // activation r = sqrt(p) / sqrt(F) (ive.py:210); OIVA_MODEL_GAUSS serves ive.py:213
hipError_t launch_ogive_init(hipStream_t s, const OgiveState& st, int F, int M, int mode);
hipError_t launch_ogive_switch(hipStream_t s, const OgiveState& st, int F, int M);
hipError_t launch_ogive_step(hipStream_t s, const OgiveState& st, const void* Vpart, bool vpart_f64, int nsplit, int T, int F,
                             int M, double mu, double tol);

// W_hat = [eigenvectors of the K largest eigenvalues of Cx, ascending | [0; -I]]   (auxiva_pca.py:75-81); evals (F, M)
// ascending or nullptr; lapack_phase: W = conj(vecs) with every vector's largest component real (overiva.py:106-109)
hipError_t launch_pca_subspace(hipStream_t s, const double* Cx, float2* What, double2* What64, double* evals, int F, int M, int K,
                               bool lapack_phase);

// dense complex128 <-> complex64 conversion on the device
hipError_t launch_cast_c128_to_c64(hipStream_t s, const double2* in, float2* out, long long n);
hipError_t launch_cast_c64_to_c128(hipStream_t s, const float2* in, double2* out, long long n);

// unpack packed Hermitian float64 [nmat][M*M] -> full complex nmat x (M,M): complex64, or complex128 when out_f64
hipError_t launch_unpack_herm(hipStream_t s, const double* packed, void* full, bool out_f64, long long nmat, int M);

}  // namespace oiva
