// X-resident iteration kernel, 2 channels (see resident_kernel.inc)
#include "resident_kernel.inc"

namespace oiva {
hipError_t launch_resident_m2(hipStream_t s, const ResidentArgs& a, int K, bool update_f64, bool cov_f64) {
    return launch_resident_m<2>(s, a, K, update_f64, cov_f64);
}
}  // namespace oiva
