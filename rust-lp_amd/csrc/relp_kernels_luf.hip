// relp_kernels_luf.hip -- the LU factorisation of the basis as a device kernel (reference:
// carry/lower_upper/decomposition/mod.rs:27-138, decomposition/pivoting.rs:45-81).  The algorithm is relp_lu_factor_core.h,
// compiled here with its parallel loops as thread-strided loops of ONE workgroup ending in barriers; the same source runs on
// the host in tests/cpp/test_lu_device_model.cpp.  One workgroup because the factorisation runs BESIDE the persistent pivot
// kernel (another CU, another stream): what it has to beat is the download / host threads / upload it replaces, not a
// chip-wide kernel.
#define RELP_LUF_DEVICE 1
#include "relp_lu_factor_core.h"
#include "relp_lu_schedule_core.h"

namespace relp {

namespace {
constexpr int kLufThreads = 512;

__global__ __launch_bounds__(kLufThreads) void k_lu_factor(LufMatrix M, const int32_t* basis, LufWork W, LufOut O) {
    luf_factor(M, basis, W, O);
}
}  // namespace

// The four solve schedules from the factors the kernel above left: levels, "ELL by pass" images, trivial-row lists, reach
// arrays (relp_lu_schedule_core.h), and what the Forrest-Tomlin update needs to know about every pivot.
struct LufSchedAll { LufSchedIn in[4]; LufSchedOut out[4]; };
__global__ __launch_bounds__(kLufThreads) void k_lu_schedules(LufSchedAll A, LufSchedWork S, LufWork W, const int32_t* status, FtPivotInfo* pinfo) {
    if (status[0] != LUF_OK) return;                   // (the factorisation failed or gave up: nothing to schedule)
    for (int q = 0; q < 4; ++q) {
        luf_build_schedule(A.in[q], S, A.out[q], W);
        __syncthreads();
    }
    const int32_t m = A.in[0].m;
    for (int k = threadIdx.x; k < m; k += blockDim.x) {
        FtPivotInfo p;
        p.u_e0 = A.in[1].ptr[k]; p.u_e1 = A.in[1].ptr[k + 1];
        p.via_u0 = p.via_u1 = p.via_t0 = p.via_t1 = 0;   // (levels are not fused here: no substituted entries to cancel)
        p.lev_ub = A.out[2].level_of[k]; p.pad_ = 0;
        pinfo[k] = p;
    }
}

void launch_lu_schedules(const LufSchedIn in[4], const LufSchedOut out[4], const LufSchedWork& S, const LufWork& W, const int32_t* status,
                         FtPivotInfo* pinfo, hipStream_t s) {
    LufSchedAll A;
    for (int q = 0; q < 4; ++q) { A.in[q] = in[q]; A.out[q] = out[q]; }
    hipLaunchKernelGGL(k_lu_schedules, dim3(1), dim3(kLufThreads), 0, s, A, S, W, status, pinfo);
}

void launch_lu_factor(const LufMatrix& M, const int32_t* basis, const LufWork& W, const LufOut& O, hipStream_t s) {
    hipLaunchKernelGGL(k_lu_factor, dim3(1), dim3(kLufThreads), 0, s, M, basis, W, O);
}
int32_t luf_threads() { return kLufThreads; }

}  // namespace relp
