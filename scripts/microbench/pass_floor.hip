// pass_floor.hip -- what one level of a level-scheduled triangular solve costs at the very least on one CU: 512 threads, the
// first 256 work.  Variants add one ingredient at a time (ticks per pass printed).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CTRL> __device__ __forceinline__ double dpp_shl(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// variant bits: 1 = barrier per pass, 2 = idle waves spin on an LDS flag read per pass, 4 = header decode via readfirstlane,
// 8 = masked reduction (fma with 0/1), 16 = working waves = 1 instead of 4
template <int V>
__global__ __launch_bounds__(512) void k_floor(const unsigned short* gidx, const double* gval, int passes, int m, long long* out, double* xg) {
    __shared__ double x[1024];
    __shared__ unsigned short sidx[256 * 32];
    __shared__ double sval[256 * 32];
    __shared__ int4 hdr[32];
    const int tid = threadIdx.x;
    for (int i = tid; i < 256 * 32; i += 512) { sidx[i] = gidx[i]; sval[i] = gval[i]; }
    for (int i = tid; i < 1024; i += 512) x[i] = 1.0 + i * 1e-3;
    if (tid < 32) hdr[tid] = make_int4(tid * 256, 48, 3 | (1 << 8), 0);
    __syncthreads();
    constexpr int NW = (V & 16) ? 64 : 256;
    const long long t0 = clock64();
    if (tid >= NW) {
        if (V & 1) for (int p = 0; p < passes; ++p) { if (V & 2) { if ((hdr[p & 31].z >> 8) & 1) __syncthreads(); } else __syncthreads(); }
    } else {
        int c_idx = sidx[tid]; double c_val = sval[tid];
        int lanes = 48, lane0 = 0;
        for (int p = 0; p < passes; ++p) {
            const double xv = x[c_idx];
            const int np = (p + 1) & 31;
            int nl0 = np * 256;
            if (V & 4) { const int4 h = hdr[np]; nl0 = __builtin_amdgcn_readfirstlane(h.x); lanes = __builtin_amdgcn_readfirstlane(h.y); }
            const int n_idx = sidx[nl0 + tid]; const double n_val = sval[nl0 + tid];
            double sum = tid < lanes ? -c_val * xv : 0.0;
            if (V & 8) {
                const int lg = 3;
                sum = fma(dpp_shl<0x104>(sum), lg >= 3 ? 1.0 : 0.0, sum);
                sum = fma(dpp_shl<0x102>(sum), lg >= 2 ? 1.0 : 0.0, sum);
                sum = fma(dpp_shl<0x101>(sum), lg >= 1 ? 1.0 : 0.0, sum);
            } else {
                sum += dpp_shl<0x104>(sum); sum += dpp_shl<0x102>(sum); sum += dpp_shl<0x101>(sum);
            }
            const bool lead = tid < lanes && (tid & 7) == 0;
            x[lead ? (c_idx + 1) & 1023 : 1023] = (xv + sum) * 0.5;
            if (V & 1) __syncthreads();
            c_idx = n_idx; c_val = n_val; lane0 = nl0;
        }
        if (lane0 == -1) x[0] = 0;
    }
    const long long t1 = clock64();
    if (tid == 0) out[0] = t1 - t0;
    if (tid < 256) xg[tid] = x[tid];
}

template <int V> void run(const unsigned short* gi, const double* gv, long long* out, double* xg, const char* what) {
    const int passes = 2000;
    hipLaunchKernelGGL(k_floor<V>, dim3(1), dim3(512), 0, 0, gi, gv, passes, 790, out, xg);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    printf("variant %2d  %-70s %6.0f ticks per pass  (%s)\n", V, what, (double)h / passes, hipGetErrorString(hipGetLastError()));
}

int main() {
    std::vector<unsigned short> idx(256 * 32); std::vector<double> val(256 * 32);
    for (size_t i = 0; i < idx.size(); ++i) { idx[i] = (unsigned short)((i * 37 + 11) % 790); val[i] = 1e-3 * (1 + i % 7); }
    unsigned short* gi; double* gv; long long* out; double* xg;
    hipMalloc(&gi, idx.size() * 2); hipMalloc(&gv, val.size() * 8); hipMalloc(&out, 64); hipMalloc(&xg, 2048);
    hipMemcpy(gi, idx.data(), idx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(gv, val.data(), val.size() * 8, hipMemcpyHostToDevice);
    run<0>(gi, gv, out, xg, "4 waves, gather + 3-step reduce + store, NO barrier");
    run<16>(gi, gv, out, xg, "1 wave, no barrier");
    run<1>(gi, gv, out, xg, "4 waves + 4 idle, barrier per pass");
    run<17>(gi, gv, out, xg, "1 wave + 7 idle, barrier per pass");
    run<3>(gi, gv, out, xg, "barrier, idle waves read a flag from LDS per pass");
    run<5>(gi, gv, out, xg, "barrier + header decode (readfirstlane)");
    run<9>(gi, gv, out, xg, "barrier + masked reduction");
    run<15>(gi, gv, out, xg, "barrier + flag-reading idle waves + header decode + masked reduction");
    return 0;
}
