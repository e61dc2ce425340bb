// Host side of the X-resident iteration (resident_kernel.inc): which shapes qualify, and the dispatch by channel count.
#include "resident.h"

namespace oiva {

// Workgroup (g, c) keeps 16 bins x TW frames on chip; the grid NB x NS must be resident as a whole, one workgroup
// per compute unit (each one takes most of a CU's LDS and, with JR > 0, all of its registers).
bool resident_geometry(int T, int F, int M, int K, int n_cu, int ns_req, ResidentGeom* out) {
    if (!(M == 2 || M == 4 || M == 6 || M == 8) || !(K == 1 || K == 2) || K >= M) return false;   // even channel counts; structured update: 1 or 2 sources + background
    ResidentGeom g;
    g.NB = (F + 15) / 16;
    if (g.NB > n_cu) return false;
    const int lane_frame_bytes = M * 8;
    const int jl_max = kResidentLdsXBytes / (kBlock * lane_frame_bytes);      // 8 channels: 8, 4 channels: 16
    const int j_max = std::min(kResidentMaxTW / 16, jl_max + (M >= 6 ? kResidentRegFrames : 0));
    int ns = n_cu / g.NB;
    ns = std::min(ns, 32);                                   // the update adds the NS partials of its bin: one or two rounds of loads
    ns = std::min(ns, std::max(1, T / 16));
    if (ns_req > 0) {                                        // the ranks of a sharded run agree on one split count
        if (ns_req > ns) return false;
        ns = ns_req;
    }
    if (ns < 1) return false;
    g.TW = ((T + ns - 1) / ns + 15) / 16 * 16;
    g.NS = (T + g.TW - 1) / g.TW;
    g.J = g.TW / 16;
    if (g.J > j_max) return false;                           // does not fit on chip
    // (+ 4 KB of static LDS in the kernel: the row's sums of r as fetched by each wave)
    // every bin of a group needs an update slot on the workgroups of its row: bin b -> workgroup b % NS, slot b / NS;
    // a workgroup has 4 waves x (64 / M^2) slots
    const int mp = M <= 2 ? 2 : (M <= 4 ? 4 : 8);            // the update's square lane layout (power of two)
    const int slots = 4 * (64 / (mp * mp));
    if ((16 + g.NS - 1) / g.NS > slots) return false;
    g.JR = g.J > jl_max ? kResidentRegFrames : 0;
    const int jl = g.J - g.JR;
    g.lds_bytes = jl * (M / 2) * kBlock * 16 + kResidentMaxTW * K * 8 /* weights: float, or double (float64 covariance) */ +
                  16 * (kBlock + 1) * 4 + kWaves * 8 + 16 + kResidentMaxTW * K * 4 /* the activations r, for the floor */;
    *out = g;
    return true;
}

hipError_t launch_resident(hipStream_t s, const ResidentArgs& a, int M, int K, bool update_f64, bool cov_f64) {
    if (M == 8) return launch_resident_m8(s, a, K, update_f64, cov_f64);
    if (M == 6) return launch_resident_m6(s, a, K, update_f64, cov_f64);
    if (M == 4) return launch_resident_m4(s, a, K, update_f64, cov_f64);
    if (M == 2) return launch_resident_m2(s, a, K, update_f64, cov_f64);
    return hipErrorInvalidValue;
}

}  // namespace oiva
