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

// The bump's working set in LDS when it fits (lds_nb rows / columns, lds_arena entries: a dependent LDS round trip is ~64 clocks,
// one to L2 ~700, and every phase of a round is a chain of them): the peel runs on the global arrays, then the bump and the views
// on a LufWork whose bump arrays, arena and hot counters point into LDS.  A bump beyond lds_nb rows takes the global arrays; an
// arena that overflows in LDS ends LUF_NO_ROOM and the host launches the kernel again with lds_nb = 0.
__global__ __launch_bounds__(kLufThreads) void k_lu_factor(LufMatrix M, const int32_t* basis, LufWork W, LufOut O, int32_t lds_nb, int32_t lds_arena) {
    extern __shared__ __align__(16) char luf_smem[];
    const int32_t k_peel = luf_peel(M, basis, W, O);
    if (k_peel < 0) return;
    const int32_t nb = M.m - k_peel;
    if (lds_nb > 0 && nb > 0 && nb <= lds_nb) {
        LufWork L = W;
        char* p = luf_smem;
        auto i32 = [&](int64_t n) { int32_t* r = reinterpret_cast<int32_t*>(p); p += (4 * n + 15) / 16 * 16; return r; };
        auto f64 = [&](int64_t n) { double* r = reinterpret_cast<double*>(p); p += 8 * n; return r; };
        auto u64 = [&](int64_t n) { unsigned long long* r = reinterpret_cast<unsigned long long*>(p); p += 8 * n; return r; };
        L.pval = f64(lds_nb); L.cmax = u64(lds_nb); L.rowmark = u64(lds_nb); L.colbest = u64(lds_nb); L.cprio = u64(lds_nb);
        L.eval = f64(lds_arena); L.red = u64(8);
        L.rbeg = i32(lds_nb); L.rlen = i32(lds_nb); L.rcap = i32(lds_nb); L.ract = i32(lds_nb); L.cact = i32(lds_nb); L.bcc = i32(lds_nb);
        L.bstep_row = i32(lds_nb); L.bstep_col = i32(lds_nb); L.cpiv = i32(lds_nb); L.prank = i32(lds_nb); L.acc = i32(lds_nb);
        L.ecol = i32(lds_arena); L.scalars = i32(16); L.counters = i32(128);
        // (unconditional, like everything carved here: a pointer that is LDS on one path and global on the other is a generic
        // pointer, and every access through it a FLAT instruction)
        L.dense = f64((int64_t)W.dense_cap * W.dense_cap); L.dint = i32(8 * (int64_t)W.dense_cap + 4);
        L.nb_cap = lds_nb; L.arena_cap = lds_arena;
        // (the views run when the arena is no longer read: their counts and buckets in its values' LDS)
        // (launch_lu_factor takes this variant only when m + 2 counters fit beside the buckets)
        L.vw = reinterpret_cast<int32_t*>(L.eval); L.vtmp = L.vw + ((M.m + 2 + 3) / 4) * 4; L.vtmp_cap = 2 * lds_arena - ((M.m + 2 + 3) / 4) * 4;
        for (int i = threadIdx.x; i < 128; i += blockDim.x) L.counters[i] = 0;
        __syncthreads();
        luf_bump(M, basis, L, O, k_peel);
        __syncthreads();
        // rounds and phase clocks back to where the host reads them
        if (threadIdx.x == 0) {
            W.counters[2] = L.counters[2]; W.counters[4] += L.counters[4];
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(W.counters + 8);
            const unsigned long long* src = reinterpret_cast<const unsigned long long*>(L.counters + 8);
            for (int i = 2; i < 24; ++i) dst[i] += src[i];
            W.counters[3] += 1;                            // factorisations whose bump ran in LDS
        }
    } else {
        luf_bump(M, basis, W, O, k_peel);
    }
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

// LDS for a bump of lds_nb rows with an arena of lds_arena entries (launch_lu_factor's dynamic shared memory)
static size_t luf_lds_bytes(int32_t lds_nb, int32_t lds_arena, int32_t dense_cap) {
    auto up = [](size_t b) { return (b + 15) / 16 * 16; };
    return 5 * 8 * (size_t)lds_nb + 8 * (size_t)lds_arena + 64 + 11 * up(4 * (size_t)lds_nb) + up(4 * (size_t)lds_arena) + up(64) + up(512) + 64 +
           8 * (size_t)dense_cap * dense_cap + up(4 * (8 * (size_t)dense_cap + 4));
}
void launch_lu_factor(const LufMatrix& M, const int32_t* basis, const LufWork& W, const LufOut& O, hipStream_t s, bool lds) {
    // (what one CU's 160 KB hold beside the kernel's static 100 bytes: 512 bump rows, the 64 x 64 block of the dense finish and
    // what is left as arena: 7,040 entries with the block, 9,216 without)
    if (M.m + 2 > 7040) lds = false;                   // (the views' counters sit in the arena's LDS)
    const int32_t lds_nb = lds ? 512 : 0, lds_arena = lds ? (W.dense_cap > 32 ? 7040 : W.dense_cap > 0 ? 8448 : 9216) : 0;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lu_factor), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512); attr_set = true; }
    hipLaunchKernelGGL(k_lu_factor, dim3(1), dim3(kLufThreads), lds ? luf_lds_bytes(lds_nb, lds_arena, W.dense_cap) : 0, s, M, basis, W, O, lds_nb, lds_arena);
}
int32_t luf_threads() { return kLufThreads; }

}  // namespace relp
