// relp_kernels_luf.hip -- the LU factorisation of the basis as device kernels (reference:
// carry/lower_upper/decomposition/mod.rs:27-138, decomposition/pivoting.rs:45-81).  The algorithms are relp_lu_factor_core.h
// and relp_lu_schedule_core.h, compiled here with their parallel loops as thread-strided loops of a workgroup ending in
// barriers; the same source runs on the host in tests/cpp/test_lu_device_model.cpp.
//   k_lu_factor     one workgroup: singleton peeling in parallel rounds, the bump in rounds of independent Markowitz pivots,
//                   L and U row-wise and column-wise
//   k_lu_schedules  four workgroups, one per solve schedule (L, U, U', L'): levels, fusion by local inversion, "ELL by pass"
//                   images, via lists, reach arrays
//   k_lu_pinfo      what a Forrest-Tomlin update needs to know about every pivot (one 32-byte record)
// The factorisation runs BESIDE the persistent pivot kernel (other CUs, another stream when the look-ahead is on): what it has to
// beat is the download / host threads / upload it replaces.
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

struct LufSchedAll { LufSchedIn in[4]; LufSchedWork work[4]; LufSchedOut out[4]; };
__global__ __launch_bounds__(kLufThreads) void k_lu_schedules(LufSchedAll A, const int32_t* status) {
    if (status[0] != LUF_OK) return;                   // (the factorisation failed or gave up: nothing to schedule)
    const int q = blockIdx.x;
    LufSchedIn T = A.in[q];
    luf_build_schedule(T, A.work[q], A.out[q]);
    __syncthreads();
    // (more right-hand-side copies than the layout has room for: the schedule once more, level by level -- what
    // Engine::lu_upload_factors does with the host's)
    if (T.fuse_lanes > 0 && A.out[q].desc[LUF_D_STATUS] == LUF_NO_ROOM) {
        __syncthreads();
        T.fuse_lanes = 0;
        luf_build_schedule(T, A.work[q], A.out[q]);
    }
}

__global__ void k_lu_pinfo(int32_t m, const int32_t* u_ptr, const int32_t* via_u, const int32_t* via_t, const int32_t* lev_ub, const int32_t* status,
                           FtPivotInfo* pinfo) {
    if (status[0] != LUF_OK) return;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < m; k += gridDim.x * blockDim.x) {
        FtPivotInfo p;
        p.u_e0 = u_ptr[k]; p.u_e1 = u_ptr[k + 1];
        p.via_u0 = via_u[k]; p.via_u1 = via_u[k + 1]; p.via_t0 = via_t[k]; p.via_t1 = via_t[k + 1];
        p.lev_ub = lev_ub[k]; p.pad_ = 0;
        pinfo[k] = p;
    }
}

void launch_lu_schedules(const LufSchedIn in[4], const LufSchedWork work[4], const LufSchedOut out[4], const int32_t* status,
                         FtPivotInfo* pinfo, hipStream_t s) {
    LufSchedAll A;
    for (int q = 0; q < 4; ++q) { A.in[q] = in[q]; A.work[q] = work[q]; A.out[q] = out[q]; }
    hipLaunchKernelGGL(k_lu_schedules, dim3(4), dim3(kLufThreads), 0, s, A, status);
    const int32_t m = in[0].m;
    hipLaunchKernelGGL(k_lu_pinfo, dim3((m + 255) / 256), dim3(256), 0, s, m, in[1].ptr, out[1].via_ptr, out[2].via_ptr, out[2].level_of, status, pinfo);
}

void launch_lu_factor(const LufMatrix& M, const int32_t* basis, const LufWork& W, const LufOut& O, hipStream_t s) {
    hipLaunchKernelGGL(k_lu_factor, dim3(1), dim3(kLufThreads), 0, s, M, basis, W, O);
}
int32_t luf_threads() { return kLufThreads; }

}  // namespace relp
