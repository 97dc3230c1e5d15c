// lu_probe.cpp -- host-only look at one mid-solve basis (RELP_DUMP_BASIS text dump): time lu_factor, print the level
// structure of the four schedules.  g++ -O2 -std=c++17 -I rust-lp_amd/csrc scripts/host/lu_probe.cpp rust-lp_amd/csrc/relp_lu.cpp
#include "relp_lu.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
using namespace relp;
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = std::fopen(argv[1], "r");
    if (!f) return 2;
    int m;
    if (std::fscanf(f, "%d", &m) != 1) return 2;
    std::vector<std::vector<std::pair<int32_t, double>>> cols(m);
    for (auto& c : cols) {
        int n; if (std::fscanf(f, "%d", &n) != 1) return 2;
        c.resize(n);
        for (auto& e : c) if (std::fscanf(f, "%d %lf", &e.first, &e.second) != 2) return 2;
        std::sort(c.begin(), c.end());
    }
    LUFactors lu; std::string err;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 20;
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        const auto t0 = std::chrono::steady_clock::now();
        if (!lu_factor(m, cols, &lu, &err)) { std::printf("singular: %s\n", err.c_str()); return 1; }
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    std::printf("m %d nnz_l %lld nnz_u %lld lu_factor best %.0f us\n", m, (long long)lu.nnz_l, (long long)lu.nnz_u, best);
    const TriangularSchedule* s[4] = {&lu.Lf, &lu.Uf, &lu.Ub, &lu.Lb};
    const char* nm[4] = {"L", "U", "U'", "L'"};
    for (int k = 0; k < 4; ++k) {
        const auto& t = *s[k];
        const int nl = (int)t.level_ptr.size() - 1;
        std::printf("%s: %d levels; rows(entries) per level:", nm[k], nl);
        for (int l = 0; l < nl; ++l) {
            long e = 0;
            for (int i = t.level_ptr[l]; i < t.level_ptr[l + 1]; ++i) { const int r = t.level_rows[i]; e += t.ptr[r + 1] - t.ptr[r]; }
            std::printf(" %d(%ld)", t.level_ptr[l + 1] - t.level_ptr[l], e);
        }
        std::printf("\n");
    }
    // level fusion: groups, passes, lanes; and the fused solve against the plain one on a random right-hand side
    const int cap = argc > 3 ? std::atoi(argv[3]) : 256;
    for (int k = 0; k < 4; ++k) {
        FusedSchedule fs, plain;
        const bool mask = k == 1 || k == 2;
        double us = 1e30, us_pack = 1e30;
        for (int r = 0; r < reps; ++r) {
            const auto t0 = std::chrono::steady_clock::now();
            fuse_levels(*s[k], mask, mask, cap, &fs);
            const auto t1 = std::chrono::steady_clock::now();
            EllPacked tmp; ell_pack(fs, mask, &tmp);
            us = std::min(us, std::chrono::duration<double, std::micro>(t1 - t0).count());
            us_pack = std::min(us_pack, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count());
        }
        fuse_levels(*s[k], mask, mask, 0, &plain);
        EllPacked e, e0;
        ell_pack(fs, mask, &e);
        ell_pack(plain, mask, &e0);
        std::vector<double> b(m), x(m), xr;
        unsigned long long st = 12345 + k;
        for (auto& v : b) { st = st * 6364136223846793005ull + 1442695040888963407ull; v = ((st >> 33) % 7 == 0) ? (double)((st >> 40) % 1000) / 100.0 - 5.0 : 0.0; }
        xr = b;
        {   const auto& t = *s[k];
            for (size_t l = 0; l + 1 < t.level_ptr.size(); ++l) for (int i = t.level_ptr[l]; i < t.level_ptr[l + 1]; ++i) {
                const int r = t.level_rows[i]; double sum = xr[r];
                for (int q = t.ptr[r]; q < t.ptr[r + 1]; ++q) sum -= t.val[q] * xr[t.idx[q]];
                xr[r] = sum / t.diag[r]; } }
        x = b;
        {   const auto& t = fs.s;
            for (size_t l = 0; l + 1 < t.level_ptr.size(); ++l) for (int i = t.level_ptr[l]; i < t.level_ptr[l + 1]; ++i) {
                const int r = t.level_rows[i]; double sum = x[r];
                for (int q = t.ptr[r]; q < t.ptr[r + 1]; ++q) sum -= t.val[q] * (t.idx[q] >= fs.rhs_base ? b[t.idx[q] - fs.rhs_base] : x[t.idx[q]]);
                x[r] = sum / t.diag[r]; } }
        double err = 0, nrm = 0;
        for (int i = 0; i < m; ++i) { err = std::max(err, std::fabs(x[i] - xr[i])); nrm = std::max(nrm, std::fabs(xr[i])); }
        { int hist[8] = {0}; for (auto& ps : e.passes) ++hist[ps.info & 7]; std::printf("%s passes by widest row (2^lg lanes), lg 0..6:", nm[k]); for (int q = 0; q < 7; ++q) std::printf(" %d", hist[q]); std::printf("\n"); }
        std::printf("%s: %zu levels / %zu passes / %zu lanes  ->  %zu groups / %zu passes / %zu lanes (+%zu ovf), via %zu, fuse %.0f us, pack %.0f us, |dx| %.2e of %.2e\n",
                    nm[k], s[k]->level_ptr.size() - 1, e0.passes.size(), e0.sidx.size(), fs.s.level_ptr.size() - 1, e.passes.size(),
                    e.sidx.size(), e.oidx.size(), e.via_pos.size(), us, us_pack, err, nrm);
    }
    return 0;
}
