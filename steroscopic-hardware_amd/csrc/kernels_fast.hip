// kernels_fast.hip -- placeholder until the column-wave kernel lands (next commit).
#include "sadhip_kernels.h"
namespace sadhip {
bool fast_supported(const Job&) { return false; }
hipError_t launch_fast(const Job& j, hipStream_t s) { return launch_generic(j, s); }
}
