// Microbenchmark: what HBM rate does the access pattern of the tableau flush (T0 += W R0) reach with no arithmetic?
// A column-major m x n f64 matrix is updated tile by tile (128 columns x 128 rows per 512-thread workgroup, each
// lane touching the elements the f64 MFMA accumulator layout gives it: 16 lanes x 8 bytes contiguous, four
// columns per instruction), (a) in place, (b) out of place into a second matrix, and (c) as a plain linear
// 16-byte-per-lane copy for reference.  Build: hipcc --offload-arch=gfx950 -O3 tile_rmw.hip -o tile_rmw
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool ROWS_FIRST>
__global__ __launch_bounds__(512) void k_tile(const double* __restrict__ src, double* __restrict__ dst, int m, int n, int64_t ld) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wc = wave % 2, wr = wave / 2;
    const int bx = ROWS_FIRST ? blockIdx.x : blockIdx.y, by = ROWS_FIRST ? blockIdx.y : blockIdx.x;
    const int c_wave = by * 128 + wc * 64, i_wave = bx * 128 + wr * 32;
    const int lm = lane & 15, lk = lane >> 4;
    double v[4][2][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = c_wave + a * 16 + lk + 4 * g, i = i_wave + b * 16 + lm;
                v[a][b][g] = (c < n && i < m) ? src[(int64_t)c * ld + i] : 0.0;
            }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = c_wave + a * 16 + lk + 4 * g, i = i_wave + b * 16 + lm;
                if (c < n && i < m) dst[(int64_t)c * ld + i] = v[a][b][g] + 1.0;
            }
}

__global__ __launch_bounds__(256) void k_linear(const double2* __restrict__ src, double2* __restrict__ dst, int64_t count) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        double2 v = src[i];
        v.x += 1.0; v.y += 1.0;
        dst[i] = v;
    }
}

int main() {
    const int m = 10000, n = 20000;
    const int64_t ld = 10000, count = ld * n;
    double *a, *b;
    CK(hipMalloc(&a, count * sizeof(double)));
    CK(hipMalloc(&b, count * sizeof(double)));
    CK(hipMemset(a, 0, count * sizeof(double)));
    CK(hipMemset(b, 0, count * sizeof(double)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 16.0 * count;
    auto report = [&](const char* name, float ms, int reps) {
        printf("%-44s %8.1f us  %6.2f TB/s\n", name, ms * 1e3 / reps, bytes * reps / (ms * 1e-3) / 1e12);
        fflush(stdout);
    };
    const int reps = 10;
    float ms;
    dim3 grid_rows((m + 127) / 128, (n + 127) / 128), grid_cols((n + 127) / 128, (m + 127) / 128);
    for (int pass = 0; pass < 2; ++pass) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_tile<true>, grid_rows, dim3(512), 0, 0, a, a, m, n, ld);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (pass) report("tiles, in place, consecutive WGs down a column", ms, reps);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_tile<false>, grid_cols, dim3(512), 0, 0, a, a, m, n, ld);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (pass) report("tiles, in place, consecutive WGs along a row", ms, reps);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_tile<true>, grid_rows, dim3(512), 0, 0, a, b, m, n, ld);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (pass) report("tiles, out of place", ms, reps);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r)
            hipLaunchKernelGGL(k_linear, dim3(256 * 16), dim3(256), 0, 0, (const double2*)a, (double2*)a, count / 2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (pass) report("linear 16-byte lanes, in place", ms, reps);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r)
            hipLaunchKernelGGL(k_linear, dim3(256 * 16), dim3(256), 0, 0, (const double2*)a, (double2*)b, count / 2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (pass) report("linear 16-byte lanes, out of place (copy)", ms, reps);
    }
    return 0;
}
