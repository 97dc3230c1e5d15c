// lu_probe.cpp -- host-only look at one mid-solve basis (RELP_DUMP_BASIS text dump): time lu_factor, print the level
// structure of the four schedules.  g++ -O2 -std=c++17 -I rust-lp_amd/csrc scripts/host/lu_probe.cpp rust-lp_amd/csrc/relp_lu.cpp
#include "relp_lu.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
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
    return 0;
}
