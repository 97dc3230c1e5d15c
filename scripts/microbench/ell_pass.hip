// ell_pass.hip -- ticks per pass of the persistent kernel's triangular solve (relp_lu_device.h: ell_solve) on a synthetic
// bidiagonal-plus-random schedule staged in LDS, one workgroup of 512 threads like the pivot kernel.
// Build on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rust-lp_amd/csrc scripts/microbench/ell_pass.hip rust-lp_amd/csrc/relp_lu.cpp -o scripts/microbench/ell_pass
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
#include <cmath>
#include <algorithm>

#include "relp_lu_device.h"
#include "relp_lu.hpp"

using namespace relp;

template <int kVariant>
__global__ __launch_bounds__(512) void k_bench(EllSchedule s, int m, int reps, long long* out, double* xg) {
    extern __shared__ __align__(16) char lds[];
    double* x = reinterpret_cast<double*>(lds);
    char* stage = lds + lu_up16(8LL * (2 * m + 1));
    for (int k = threadIdx.x; k <= m; k += 512) x[k] = 1.0 + 0.001 * k;
    __syncthreads();
    long long stage_ticks = 0, ts = 0;
    auto lap = [&]() { stage_ticks += clock64() - ts; };
    auto solve = [&]() {
        ts = clock64();
        if (kVariant == 0) ell_solve<true, 512, 256>(s, stage, x, m, 0, lap);
        else ell_solve_pp<true, 512>(s, stage, x, m, 0, lap);
    };
    long long t0 = clock64();
    solve();
    long long t1 = clock64();
    const long long first_stage = stage_ticks;
    for (int r = 1; r < reps; ++r) solve();
    long long t2 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (t2 - t1) / (reps > 1 ? reps - 1 : 1); out[2] = (stage_ticks - first_stage) / (reps > 1 ? reps - 1 : 1); }
    for (int k = threadIdx.x; k < m; k += 512) xg[k] = x[k];
}

static void run(const char* name, const TriangularSchedule& t, bool maskable, int variant, int fuse_cap) {
    const int m = (int)t.diag.size(), nlev = (int)t.level_ptr.size() - 1;
    FusedSchedule fs; fuse_levels(t, maskable, maskable, fuse_cap, &fs);
    EllPacked e; ell_pack(fs, maskable, &e);
    std::vector<char> buf;
    auto put = [&](const void* src, size_t bytes) { size_t o = buf.size(); buf.resize(o + (bytes + 15) / 16 * 16); if (bytes) memcpy(buf.data() + o, src, bytes); return o; };
    { std::vector<EllPassHost> hd(e.passes); hd.resize(hd.size() + kEllPadHeaders, EllPassHost{0, 0, 0, 0}); put(hd.data(), 16 * hd.size()); } put(e.lvl_pass.data(), 4 * e.lvl_pass.size()); put(e.rdiag.data(), 8 * e.rdiag.size());
    put(e.sval.data(), 8 * e.sval.size()); put(e.oval.data(), 8 * e.oval.size()); put(e.rovf.data(), 4 * e.rovf.size());
    put(e.sidx.data(), 2 * e.sidx.size()); put(e.oidx.data(), 2 * e.oidx.size());
    char* d; (void)hipMalloc(&d, buf.size()); (void)hipMemcpy(d, buf.data(), buf.size(), hipMemcpyHostToDevice);
    auto up16 = [](int64_t b) { return (b + 15) / 16 * 16; };
    EllSchedule s{}; char* q = d;
    s.passes = (const EllPass*)q; q += up16(16 * (e.passes.size() + kEllPadHeaders)); s.lvl_pass = (const int32_t*)q; q += up16(4 * e.lvl_pass.size());
    s.rdiag = (double*)q; q += up16(8 * e.rdiag.size()); s.sval = (double*)q; q += up16(8 * e.sval.size());
    s.oval = (const double*)q; q += up16(8 * e.oval.size()); s.rovf = (const int32_t*)q; q += up16(4 * e.rovf.size());
    s.sidx = (const uint16_t*)q; q += up16(2 * e.sidx.size()); s.oidx = (const uint16_t*)q; q += up16(2 * e.oidx.size());
    s.n_passes = (int)e.passes.size(); s.n_levels = (int)e.lvl_pass.size() - 1;
    for (int v : fs.s.idx) if (v >= fs.rhs_base) { s.rhs_base = fs.rhs_base; break; }
    s.m = m; s.n_lanes = (int)e.sidx.size(); s.n_ovf = (int)e.oidx.size(); s.n_triv = 0; s.triv = nullptr; s.reach = nullptr;
    s.bytes = (int)(q - d);
    long long* out; double* xg; (void)hipMalloc(&out, 64); (void)hipMalloc(&xg, 8 * m);
    const size_t lds = (size_t)up16(8 * (2 * m + 1)) + s.bytes;
    if (lds > 156 * 1024) { printf("%s: image %d bytes does not fit\n", name, s.bytes); return; }
    const int reps = getenv("ELL_REPS") ? atoi(getenv("ELL_REPS")) : 3;    // (the solve is applied `reps` times to its own result)
    (void)hipFuncSetAttribute((const void*)k_bench<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute((const void*)k_bench<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    if (variant == 0) hipLaunchKernelGGL(k_bench<0>, dim3(1), dim3(512), lds, 0, s, m, reps, out, xg);
    else hipLaunchKernelGGL(k_bench<1>, dim3(1), dim3(512), lds, 0, s, m, reps, out, xg);
    (void)hipDeviceSynchronize();
    long long h[3]; (void)hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
    std::vector<double> x(m); (void)hipMemcpy(x.data(), xg, 8 * m, hipMemcpyDeviceToHost);
    std::vector<double> xr(m); for (int k = 0; k < m; ++k) xr[k] = 1.0 + 0.001 * k;
    for (int rep = 0; rep < reps; ++rep) for (int l = 0; l < nlev; ++l) for (int i = t.level_ptr[l]; i < t.level_ptr[l + 1]; ++i) { const int r = t.level_rows[i]; double sum = xr[r]; for (int q2 = t.ptr[r]; q2 < t.ptr[r + 1]; ++q2) sum -= t.val[q2] * xr[t.idx[q2]]; xr[r] = sum / t.diag[r]; }
    double err = 0, nrm = 0; for (int k = 0; k < m; ++k) { err = std::max(err, std::fabs(x[k] - xr[k])); nrm = std::max(nrm, std::fabs(xr[k])); }
    printf("%-3s variant %d fuse %3d: %3d levels -> %3d groups, %3d passes, %5d lanes, image %6d bytes: %6lld ticks per solve, %5lld of them staging = %4.0f per pass without; diff to host %.1e of %.1e, %s\n",
           name, variant, fuse_cap, nlev, s.n_levels, s.n_passes, s.n_lanes, s.bytes, h[1], h[2], (double)(h[1] - h[2]) / s.n_passes, err, nrm, hipGetErrorString(hipGetLastError()));
    (void)hipFree(d); (void)hipFree(out); (void)hipFree(xg);
}

// usage: ell_pass M ROWS_PER_LEVEL NNZ_PER_ROW [VARIANT [FUSE_LANES]]   synthetic schedule
//        ell_pass BASIS_FILE [VARIANT [FUSE_LANES]]                      the four schedules of a dumped basis (RELP_DUMP_BASIS)
int main(int argc, char** argv) {
    if (argc > 1 && (argv[1][0] < '0' || argv[1][0] > '9')) {
        FILE* f = fopen(argv[1], "r"); if (!f) return 2;
        int m; if (fscanf(f, "%d", &m) != 1) return 2;
        std::vector<std::vector<std::pair<int32_t, double>>> cols(m);
        for (auto& c : cols) { int n; if (fscanf(f, "%d", &n) != 1) return 2; c.resize(n); for (auto& en : c) if (fscanf(f, "%d %lf", &en.first, &en.second) != 2) return 2; std::sort(c.begin(), c.end()); }
        LUFactors lu; std::string err;
        if (!lu_factor(m, cols, &lu, &err)) { printf("%s\n", err.c_str()); return 1; }
        const int variant = argc > 2 ? atoi(argv[2]) : 1, fuse_cap = argc > 3 ? atoi(argv[3]) : 0;
        run("L", lu.Lf, false, variant, fuse_cap); run("U", lu.Uf, true, variant, fuse_cap);
        run("U'", lu.Ub, true, variant, fuse_cap); run("L'", lu.Lb, false, variant, fuse_cap);
        return 0;
    }
    const int m = argc > 1 ? atoi(argv[1]) : 790, rows_per_level = argc > 2 ? atoi(argv[2]) : 6, nnz_per_row = argc > 3 ? atoi(argv[3]) : 4;
    const int variant = argc > 4 ? atoi(argv[4]) : 1, fuse_cap = argc > 5 ? atoi(argv[5]) : 0;
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
    run("syn", t, true, variant, fuse_cap);
    return 0;
}
