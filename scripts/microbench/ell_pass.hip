// ell_pass.hip -- ticks per pass of the persistent kernel's triangular solve (relp_lu_device.h: ell_solve) on a synthetic
// bidiagonal-plus-random schedule staged in LDS, one workgroup of 512 threads like the pivot kernel.
// Build on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rust-lp_amd/csrc scripts/microbench/ell_pass.hip rust-lp_amd/csrc/relp_lu.cpp -o scripts/microbench/ell_pass
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>

#include "relp_lu_device.h"
#include "relp_lu.hpp"

using namespace relp;

__global__ __launch_bounds__(512) void k_bench(EllSchedule s, int m, int reps, long long* out, double* xg) {
    extern __shared__ __align__(16) char lds[];
    double* x = reinterpret_cast<double*>(lds);
    char* stage = lds + lu_up16(8LL * (m + 1));
    for (int k = threadIdx.x; k <= m; k += 512) x[k] = 1.0 + 0.001 * k;
    __syncthreads();
    long long t0 = clock64();
    ell_solve<true, 512, 256>(s, stage, x, m, 0);
    long long t1 = clock64();
    for (int r = 1; r < reps; ++r) ell_solve<true, 512, 256>(s, stage, x, m, 0);
    long long t2 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (t2 - t1) / (reps > 1 ? reps - 1 : 1); }
    for (int k = threadIdx.x; k < m; k += 512) xg[k] = x[k];
}

int main(int argc, char** argv) {
    const int m = argc > 1 ? atoi(argv[1]) : 790, rows_per_level = argc > 2 ? atoi(argv[2]) : 6, nnz_per_row = argc > 3 ? atoi(argv[3]) : 4;
    // rows 0..R0-1 without entries (level 0), then levels of `rows_per_level` rows depending on the previous level
    TriangularSchedule t;
    const int r0 = m / 3;
    t.ptr.assign(m + 1, 0); t.diag.assign(m, 2.0);
    std::mt19937 rng(5);
    std::vector<int> lev(m, 0);
    for (int k = 0; k < m; ++k) {
        t.ptr[k] = (int)t.idx.size();
        if (k >= r0) {
            const int l = 1 + (k - r0) / rows_per_level;
            lev[k] = l;
            const int prev_lo = l == 1 ? 0 : r0 + (l - 2) * rows_per_level, prev_hi = l == 1 ? r0 : r0 + (l - 1) * rows_per_level;
            for (int e = 0; e < nnz_per_row; ++e) { t.idx.push_back(prev_lo + (int)(rng() % (prev_hi - prev_lo))); t.val.push_back(0.01 * (1 + e)); }
        }
    }
    t.ptr[m] = (int)t.idx.size();
    int nlev = 0; for (int k = 0; k < m; ++k) nlev = std::max(nlev, lev[k] + 1);
    t.level_ptr.assign(nlev + 1, 0);
    for (int k = 0; k < m; ++k) ++t.level_ptr[lev[k] + 1];
    for (int l = 0; l < nlev; ++l) t.level_ptr[l + 1] += t.level_ptr[l];
    t.level_rows.resize(m);
    { std::vector<int> fill(t.level_ptr.begin(), t.level_ptr.end() - 1); for (int k = 0; k < m; ++k) t.level_rows[fill[lev[k]]++] = k; }
    EllPacked e; ell_pack(t, true, &e);
    std::vector<char> buf;
    auto put = [&](const void* src, size_t bytes) { size_t o = buf.size(); buf.resize(o + (bytes + 15) / 16 * 16); if (bytes) memcpy(buf.data() + o, src, bytes); return o; };
    { std::vector<EllPassHost> hd(e.passes); hd.resize(hd.size() + 3, EllPassHost{0, 0, 0, 0}); put(hd.data(), 16 * hd.size()); } put(e.lvl_pass.data(), 4 * e.lvl_pass.size()); put(e.rdiag.data(), 8 * e.rdiag.size());
    put(e.sval.data(), 8 * e.sval.size()); put(e.oval.data(), 8 * e.oval.size()); put(e.rovf.data(), 4 * e.rovf.size());
    put(e.sidx.data(), 2 * e.sidx.size()); put(e.oidx.data(), 2 * e.oidx.size());
    char* d; hipMalloc(&d, buf.size()); hipMemcpy(d, buf.data(), buf.size(), hipMemcpyHostToDevice);
    auto up16 = [](int64_t b) { return (b + 15) / 16 * 16; };
    EllSchedule s{}; char* q = d;
    s.passes = (const EllPass*)q; q += up16(16 * (e.passes.size() + 3)); s.lvl_pass = (const int32_t*)q; q += up16(4 * e.lvl_pass.size());
    s.rdiag = (double*)q; q += up16(8 * e.rdiag.size()); s.sval = (const double*)q; q += up16(8 * e.sval.size());
    s.oval = (const double*)q; q += up16(8 * e.oval.size()); s.rovf = (const int32_t*)q; q += up16(4 * e.rovf.size());
    s.sidx = (const uint16_t*)q; q += up16(2 * e.sidx.size()); s.oidx = (const uint16_t*)q; q += up16(2 * e.oidx.size());
    s.n_passes = (int)e.passes.size(); s.n_levels = nlev; s.m = m; s.n_lanes = (int)e.sidx.size(); s.n_ovf = (int)e.oidx.size();
    s.bytes = (int)(q - d);
    long long* out; double* xg; hipMalloc(&out, 64); hipMalloc(&xg, 8 * m);
    const size_t lds = (size_t)up16(8 * (m + 1)) + s.bytes;
    hipFuncSetAttribute((const void*)k_bench, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    hipLaunchKernelGGL(k_bench, dim3(1), dim3(512), lds, 0, s, m, 20, out, xg);
    hipDeviceSynchronize();
    long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    std::vector<double> x(m); hipMemcpy(x.data(), xg, 8 * m, hipMemcpyDeviceToHost);
    printf("m %d, %d levels, %d passes, %d lanes, image %d bytes: first solve %lld ticks, later %lld ticks = %.0f per pass (incl. staging); x[m-1] = %.6g, err %s\n",
           m, nlev, s.n_passes, s.n_lanes, s.bytes, h[0], h[1], (double)h[1] / s.n_passes, x[m - 1], hipGetErrorString(hipGetLastError()));
    return 0;
}
