// X-resident iteration: declarations shared by plan.hip and the kernels_resident_*.hip translation units.
#pragma once
#include "oiva_internal.h"

#include <algorithm>

namespace oiva {

// Geometry of the resident launch: workgroup (g, c) owns bin group g (16 bins) x frame split c (TW frames) of X for
// the whole launch; lane (b, q) of it owns the M-vector of bin b at the J frames q + 16 j, the first JR of them in
// registers and the other JL = J - JR in LDS.
struct ResidentGeom {
    int NB = 0;        // bin groups of 16
    int NS = 0;        // frame splits
    int TW = 0;        // frames per split (multiple of 16, <= 256)
    int J = 0;         // frames per lane = TW / 16
    int JR = 0;        // of them in registers (0 or kResidentRegFrames)
    int lds_bytes = 0;
};
constexpr int kResidentRegFrames = 8;
constexpr int kResidentF64ReduceFrames = 4;  // up to this many frames per lane the covariance sums are added over all 16 frame phases in float64
constexpr int kResidentTwoHopGroups = 40;    // from this many bin groups on, the powers are summed in two hops (reduce-scatter, all-gather)
constexpr int kResidentMaxTW = 256;          // one thread per frame of the split in the activation phase
constexpr int kResidentLdsXBytes = 128 << 10;   // LDS given to X (of 160 KB; the rest is reduction scratch and tables)

struct ResidentArgs {
    const float2* X;        // (T, F, M)
    const float2* What;     // (F, M, M) complex64, the state the launch starts from
    const double2* What64;  // complex128 copy kept by the float64 update, or nullptr
    float2* What_out;       // staging copies the final state is written to; the host moves them into What / What64 when
    double2* What64_out;    //   the launch finished without any workgroup giving up
    int what64_valid;       // the complex128 copy holds the current state
    const double* Cx;       // [F][M*M] packed Hermitian, / T
    // exchange buffers in this GPU's memory; every word of them is one agent-scope atomic access that carries its
    // epoch in spare mantissa bits (resident_kernel.inc)
    float* parts;           // [2 (epoch parity)][NB][NS * TW][K] partial source powers
    float* psum;            // [2 (epoch parity)][NS * TW][K] their sums over the bin groups (two-hop exchange, many bin groups)
    double* vpart;          // [NS][NB * 16][K][M*M] packed partial covariances
    double* rsum;           // [NB][(NS K + 1) & ~1] sum of the activations r over the split's frames, word c K + k of row g
    float2* wpub;           // [NB * 16][K][M] conj of the demixing vectors, for the power phase
    unsigned* ctrl;         // [0] give-up code (0 = fine)
    unsigned* xcc_tab;      // [NB][NS] (launch tag << 8) | (XCD of the workgroup + 1): which rows sit on one XCD
    // bins sharded over `world` GPUs (world == 1: unused): gath[r] = rank r's gather buffer as mapped here, fine-grained
    // memory, [2 (epoch parity)][world][NS * TW][K] sums of the ranks' parts; every rank must run the same NS x TW
    float* gath[OIVA_XCHG_MAX_RANKS];
    int rank, world;
    int loopback;           // world > 1 on ONE GPU: the leader stores the sums of all `world` slots (its own + zeros) into its own buffer
    unsigned long long* stamps;   // [n_iter][kResidentStamps] 100 MHz timestamps of workgroup 0 (stamp_all: [workgroup][n_iter][..]), or nullptr
    int stamp_all;          // every workgroup records its timestamps (tools/exp_resident_trace.py)
    int T, F, F_total, model;
    ResidentGeom g;
    int n_iter;
    unsigned epoch0;        // epochs epoch0 + 1 ... epoch0 + n_iter
    long long timeout_ticks;   // 100 MHz ticks a wait may take before the launch gives up
    int stall_block;        // test hook: this workgroup never publishes (-1: none) ...
    int stall_iter;         // ... from this iteration of the launch on
};
constexpr int kResidentStamps = 16;     // 0..8 phase boundaries, 9 payload passes of the parts hop, 10..15 sub-steps of the update (diagnostics)

// true when the shape can run resident on a chip of n_cu compute units (fills g)
// ns_req > 0 asks for that many frame splits (at most what the chip holds); the result may have fewer when TW rounds up
bool resident_geometry(int T, int F, int M, int K, int n_cu, int ns_req, ResidentGeom* g);
// cov_f64: the covariance sums in float64 (the `precise` arithmetic; 4 channels with the float64 update only)
hipError_t launch_resident(hipStream_t s, const ResidentArgs& a, int M, int K, bool update_f64, bool cov_f64);
// per-shape instantiations (kernels_resident_m4.hip, kernels_resident_m8.hip)
hipError_t launch_resident_m4(hipStream_t s, const ResidentArgs& a, int K, bool update_f64, bool cov_f64);
hipError_t launch_resident_m8(hipStream_t s, const ResidentArgs& a, int K, bool update_f64, bool cov_f64);
hipError_t launch_resident_m6(hipStream_t s, const ResidentArgs& a, int K, bool update_f64, bool cov_f64);
hipError_t launch_resident_m2(hipStream_t s, const ResidentArgs& a, int K, bool update_f64, bool cov_f64);

}  // namespace oiva
