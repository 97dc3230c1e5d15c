// relp_kernels_luf.hip -- the LU factorisation of the basis as a device kernel (reference:
// carry/lower_upper/decomposition/mod.rs:27-138, decomposition/pivoting.rs:45-81).  The algorithm is relp_lu_factor_core.h,
// compiled here with its parallel loops as thread-strided loops of ONE workgroup ending in barriers; the same source runs on
// the host in tests/cpp/test_lu_device_model.cpp.  One workgroup because the factorisation runs BESIDE the persistent pivot
// kernel (another CU, another stream): what it has to beat is the download / host threads / upload it replaces, not a
// chip-wide kernel.
#define RELP_LUF_DEVICE 1
#include "relp_lu_factor_core.h"

namespace relp {

namespace {
constexpr int kLufThreads = 512;

__global__ __launch_bounds__(kLufThreads) void k_lu_factor(LufMatrix M, const int32_t* basis, LufWork W, LufOut O) {
    luf_factor(M, basis, W, O);
}
}  // namespace

void launch_lu_factor(const LufMatrix& M, const int32_t* basis, const LufWork& W, const LufOut& O, hipStream_t s) {
    hipLaunchKernelGGL(k_lu_factor, dim3(1), dim3(kLufThreads), 0, s, M, basis, W, O);
}
int32_t luf_threads() { return kLufThreads; }

}  // namespace relp
