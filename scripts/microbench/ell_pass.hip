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


// Variant 2 -- no level barriers: every row that a pass computes publishes tag[row] = gen behind its result, and a lane whose
// operand is computed in this sweep (slvl[slot] = level of the pass that computes it, 0xffff = none / the row's own right-hand
// side) polls tag[idx] until it reads gen.  The two loads of a poll are issued tag first, x second: the LDS executes a wave's
// requests in order, the producer stores x before the tag, so a tag that reads gen is followed by the new x.  Passes are dealt to
// the two sets of four waves alternately as in ell_solve_pp and every wave walks its passes in ascending order; operands come from
// earlier passes only, so some wave can always proceed.  A poll that does not end within kSpinCap tries sets out[3] and goes on.
constexpr int kSpinCap = 1 << 20;
template <int NT>
__device__ __forceinline__ void ell_solve_sf(const EllSchedule& s, char* base, double* x, int dummy, const uint16_t* slvl_g, int* tag, int gen,
                                             int first_level, long long* out) {
    const EllImage<false> im = ell_stage<true, NT>(s, base, x);
    uint16_t* slvl = reinterpret_cast<uint16_t*>(base + lu_up16(s.bytes));
    for (int i = threadIdx.x; i < s.n_lanes; i += NT) slvl[i] = slvl_g[i];
    __syncthreads();
    const double* rdiag = im.rdiag; const double* sval = im.sval; const uint16_t* sidx = im.sidx;
    const int p0 = __builtin_amdgcn_readfirstlane(im.lvl_pass[first_level]), p1 = s.n_passes;
    const int set = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8), lt = threadIdx.x & 255;
    const int4* hdr = reinterpret_cast<const int4*>(im.passes);
    const int mm = s.m;
    auto slot_of = [&](const int4& h) { const int top = h.y > 0 ? h.y - 1 : 0; return h.x + (lt < top ? lt : top); };
    volatile int* vtag = tag;
    volatile double* vx = x;
    int4 hn = hdr[min(p0 + set, p1 + kEllPadHeaders - 1)];
    int n_iv = sidx[slot_of(hn)], n_lv = slvl[slot_of(hn)];
    double n_val = sval[slot_of(hn)], n_diag = rdiag[min(n_iv & kEllIdxMask, mm)];
    for (int p = p0 + set; p < p1; p += 2) {
        const int4 h = hn; const int iv = n_iv, lv = n_lv; const double val = n_val, diag = n_diag;
        // the next pass of this set: requested before the poll
        hn = hdr[min(p + 2, p1 + kEllPadHeaders - 1)];
        { const int sn = slot_of(hn); n_iv = sidx[sn]; n_lv = slvl[sn]; n_val = sval[sn]; }
        const bool act = lt < h.y;
        const int c_idx = iv & kEllIdxMask, lg = iv >> kEllLg;
        const bool need = act && lv != 0xffff && lv >= first_level;
        double xv;
        int spins = 0;
        for (;;) {
            const int t = vtag[c_idx];
            xv = vx[c_idx];
            if (__ballot(need && t != gen) == 0ull) break;
            if (++spins > kSpinCap) { if ((threadIdx.x & 63) == 0) out[3] = 1; break; }
        }
        n_diag = rdiag[min(n_iv & kEllIdxMask, mm)];
        double sum = act ? -val * xv : 0.0;
        const int info = __builtin_amdgcn_readfirstlane(h.z);
        if ((info & 0x2ff) <= 3) sum = ell_reduce<3>(sum, lg);
        else sum = ell_reduce<6>(sum, lg);              // (rows with overflow entries: not in this probe)
        const bool lead = act && (lt & ((1 << lg) - 1)) == 0;
        vx[lead ? c_idx : dummy] = sum * diag;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        vtag[lead ? c_idx : dummy] = gen;
    }
    __syncthreads();
}

template <int kVariant>
__global__ __launch_bounds__(512) void k_bench(EllSchedule s, int m, int reps, long long* out, double* xg, const uint16_t* slvl_g) {
    extern __shared__ __align__(16) char lds[];
    double* x = reinterpret_cast<double*>(lds);
    int* tag = reinterpret_cast<int*>(lds + lu_up16(8LL * (2 * m + 2)));
    char* stage = lds + lu_up16(8LL * (2 * m + 2)) + lu_up16(4LL * (2 * m + 2));
    for (int k = threadIdx.x; k <= m; k += 512) x[k] = 1.0 + 0.001 * k;
    for (int k = threadIdx.x; k < 2 * m + 2; k += 512) tag[k] = 0;
    int gen = 0;
    __syncthreads();
    long long stage_ticks = 0, ts = 0;
    auto lap = [&]() { stage_ticks += clock64() - ts; };
    auto solve = [&]() {
        ts = clock64();
        if (kVariant == 0) ell_solve<true, 512, 256>(s, stage, x, m, 0, lap);
        else if (kVariant == 1) ell_solve_pp<true, 512>(s, stage, x, m, 0, lap);
        else { ++gen; ell_solve_sf<512>(s, stage, x, 2 * m + 1, slvl_g, tag, gen, 0, out); }
    };
    long long t0 = clock64();
    solve();
    long long t1 = clock64();
    const long long first_stage = stage_ticks;
    for (int r = 1; r < reps; ++r) solve();
    long long t2 = clock64();
    if (threadIdx.x == 0) { if (kVariant != 2) out[3] = 0; out[0] = t1 - t0; out[1] = (t2 - t1) / (reps > 1 ? reps - 1 : 1); out[2] = (stage_ticks - first_stage) / (reps > 1 ? reps - 1 : 1); }
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
    long long* out; double* xg; (void)hipMalloc(&out, 64); (void)hipMemset(out, 0, 64); (void)hipMalloc(&xg, 8 * m);
    // level of the pass that computes a slot's operand (0xffff: none, or the row's own right-hand side)
    std::vector<int> row_level(2 * (size_t)m + 2, -1);
    for (const auto& ps : e.passes)
        for (int l0 = 0; l0 < ps.lanes;) { const int iv = e.sidx[ps.lane0 + l0], lg = iv >> kEllLgShift; row_level[iv & ((1 << kEllLgShift) - 1)] = ps.level; l0 += 1 << lg; }
    std::vector<uint16_t> slvl(e.sidx.size() + 8, 0xffff);
    for (const auto& ps : e.passes)
        for (int l0 = 0; l0 < ps.lanes;) {
            const int iv = e.sidx[ps.lane0 + l0], lg = iv >> kEllLgShift, w = 1 << lg;
            for (int j = 1; j < w; ++j) { const int idx = e.sidx[ps.lane0 + l0 + j] & ((1 << kEllLgShift) - 1); if (e.sval[ps.lane0 + l0 + j] != 0.0 && row_level[idx] >= 0) slvl[ps.lane0 + l0 + j] = (uint16_t)row_level[idx]; }
            l0 += w;
        }
    uint16_t* d_slvl; (void)hipMalloc(&d_slvl, 2 * slvl.size()); (void)hipMemcpy(d_slvl, slvl.data(), 2 * slvl.size(), hipMemcpyHostToDevice);
    const size_t lds = (size_t)up16(8 * (2 * m + 2)) + (size_t)up16(4 * (2 * m + 2)) + s.bytes + (variant == 2 ? (size_t)up16(2 * e.sidx.size()) : 0);
    if (lds > 156 * 1024) { printf("%s: image %d bytes does not fit\n", name, s.bytes); return; }
    const int reps = getenv("ELL_REPS") ? atoi(getenv("ELL_REPS")) : 3;    // (the solve is applied `reps` times to its own result)
    (void)hipFuncSetAttribute((const void*)k_bench<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute((const void*)k_bench<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute((const void*)k_bench<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    if (variant == 2 && e.oidx.size() > 0) { printf("%s: rows with overflow entries, not in the sync-free probe\n", name); return; }
    if (variant == 0) hipLaunchKernelGGL(k_bench<0>, dim3(1), dim3(512), lds, 0, s, m, reps, out, xg, d_slvl);
    else if (variant == 1) hipLaunchKernelGGL(k_bench<1>, dim3(1), dim3(512), lds, 0, s, m, reps, out, xg, d_slvl);
    else hipLaunchKernelGGL(k_bench<2>, dim3(1), dim3(512), lds, 0, s, m, reps, out, xg, d_slvl);
    (void)hipDeviceSynchronize();
    long long h[4]; (void)hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    if (h[3]) printf("%s: a poll ran into the spin cap\n", name);
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
