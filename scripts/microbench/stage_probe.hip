// stage_probe.hip -- what one workgroup (512 threads, one CU) pays to bring a 36-64 KB image from global memory (L2-resident)
// into LDS, the "staging" in front of every triangular solve of the persistent pivot kernel; per repetition, so that the
// first touch (HBM) is seen apart from the steady state (L2).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/microbench/stage_probe.hip -o scripts/microbench/stage_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NT = 512, REPS = 6;

template <int kVariant>
__global__ __launch_bounds__(NT) void k_stage(const int4* src, int n16, long long* out, int* sink, const int* other = nullptr, int other_n = 0) {
    extern __shared__ __align__(16) char lds[];
    int4* dst = reinterpret_cast<int4*>(lds);
    const int tid = threadIdx.x;
    for (int r = 0; r < REPS; ++r) {
        if (other_n < 0) {                                // no memory traffic at all for -other_n clocks (LDS / ALU work only)
            const long long until = clock64() - other_n;
            while (clock64() < until) { dst[tid].x += 1; }
        } else
        if (other) {                                      // what the pivot kernel does between two stagings: other memory
            int acc = 0;
            for (int i = tid; i < other_n; i += NT) acc += other[i];
            if (acc == 0x12345678) sink[tid] = acc;
        }
        __syncthreads();
        const long long t0 = clock64();
        if (kVariant == 0) {                              // 8 x 16-byte loads in flight per thread, then 8 LDS stores
            for (int i0 = tid; i0 < n16; i0 += 8 * NT) {
                int4 buf[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + u * NT; buf[u] = src[i < n16 ? i : n16 - 1]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + u * NT; if (i < n16) dst[i] = buf[u]; }
            }
        } else if (kVariant == 1) {                       // 2 in flight
            for (int i0 = tid; i0 < n16; i0 += 2 * NT) {
                int4 buf[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) { const int i = i0 + u * NT; buf[u] = src[i < n16 ? i : n16 - 1]; }
#pragma unroll
                for (int u = 0; u < 2; ++u) { const int i = i0 + u * NT; if (i < n16) dst[i] = buf[u]; }
            }
        } else if (kVariant == 2) {                       // per-thread contiguous 128 bytes (8 x int4 back to back)
            for (int i0 = tid * 8; i0 < n16; i0 += 8 * NT) {
                int4 buf[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + u; buf[u] = src[i < n16 ? i : n16 - 1]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + u; if (i < n16) dst[i] = buf[u]; }
            }
        } else if (kVariant == 3) {                       // loads only (sum into a register): is it the LDS side?
            int acc = 0;
            for (int i0 = tid; i0 < n16; i0 += 8 * NT) {
                int4 buf[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + u * NT; buf[u] = src[i < n16 ? i : n16 - 1]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += buf[u].x ^ buf[u].w;
            }
            if (acc == 0x12345678) sink[tid] = acc;
        } else if (kVariant == 4) {                       // nontemporal / streaming loads
            for (int i0 = tid; i0 < n16; i0 += 8 * NT) {
                int4 buf[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + u * NT; typedef int v4i __attribute__((ext_vector_type(4))); const v4i v = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(src) + (i < n16 ? i : n16 - 1)); buf[u] = make_int4(v.x, v.y, v.z, v.w); }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + u * NT; if (i < n16) dst[i] = buf[u]; }
            }
        }
        __syncthreads();
        if (tid == 0) out[r] = clock64() - t0;
    }
    if (dst[tid].x == 0x7fffffff) sink[0] = 1;
}

int main(int argc, char** argv) {
    const int bytes = argc > 1 ? atoi(argv[1]) : 60000;
    const int n16 = bytes / 16;
    std::vector<int> h(n16 * 4, 3);
    int4* d; (void)hipMalloc(&d, n16 * 16); (void)hipMemcpy(d, h.data(), n16 * 16, hipMemcpyHostToDevice);
    long long* out; int* sink; (void)hipMalloc(&out, 8 * REPS); (void)hipMalloc(&sink, 4 * NT);
    auto go = [&](auto kern, const char* name) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        hipLaunchKernelGGL(kern, dim3(1), dim3(NT), (size_t)n16 * 16, 0, d, n16, out, sink, (const int*)nullptr, 0);
        (void)hipDeviceSynchronize();
        long long t[REPS]; (void)hipMemcpy(t, out, 8 * REPS, hipMemcpyDeviceToHost);
        printf("%-34s %6d bytes:", name, n16 * 16);
        for (int r = 0; r < REPS; ++r) printf(" %6lld", t[r]);
        printf("  clocks (%.1f bytes/clock in the last)\n", n16 * 16.0 / t[REPS - 1]);
    };
    go(k_stage<0>, "8 x 16 B in flight, strided");
    {   // the same with 1 MB of other traffic between the repetitions (L1 and TLB no longer hold the image)
        const int other_n = argc > 2 ? atoi(argv[2]) : 256 * 1024;
        int* other; (void)hipMalloc(&other, 4 * (size_t)(other_n > 0 ? other_n : 1)); if (other_n > 0) (void)hipMemset(other, 0, 4 * (size_t)other_n);
        (void)hipFuncSetAttribute((const void*)k_stage<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        hipLaunchKernelGGL(k_stage<0>, dim3(1), dim3(NT), (size_t)n16 * 16, 0, d, n16, out, sink, (const int*)other, other_n);
        (void)hipDeviceSynchronize();
        long long t[REPS]; (void)hipMemcpy(t, out, 8 * REPS, hipMemcpyDeviceToHost);
        printf("%-34s %6d bytes:", other_n < 0 ? "  ... after an idle gap" : "  ... with other traffic between", n16 * 16);
        for (int r = 0; r < REPS; ++r) printf(" %6lld", t[r]);
        printf("  clocks\n");
    }
    go(k_stage<1>, "2 x 16 B in flight, strided");
    go(k_stage<2>, "8 x 16 B, contiguous per thread");
    go(k_stage<3>, "loads only");
    go(k_stage<4>, "nontemporal loads");
    return 0;
}
