// Microbenchmark: cost of a device-wide barrier inside one persistent kernel on gfx950, to decide whether a
// persistent pivot kernel (grid barriers between the PRICE / RATIO / UPDATE phases) can beat three dependent
// kernel launches (~6.5 us each).  Build: hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
namespace cg = cooperative_groups;

__global__ void k_cg(int iters, double* data) {
    cg::grid_group grid = cg::this_grid();
    double v = data[blockIdx.x * blockDim.x + threadIdx.x];
    for (int it = 0; it < iters; ++it) {
        v = v * 1.0000001 + 1.0;
        data[blockIdx.x * blockDim.x + threadIdx.x] = v;
        grid.sync();
        v += data[((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x] * 1e-9;   // read a neighbour's value
    }
    data[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

// hand-rolled sense-reversing barrier with a bounded spin (never hangs: gives up after ~1e7 polls)
__device__ bool bar(unsigned* counter, unsigned* sense, unsigned& local_sense, unsigned nblocks) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        local_sense ^= 1u;
        if (atomicAdd(counter, 1u) == nblocks - 1) {
            atomicExch(counter, 0u);
            __threadfence();
            atomicExch(sense, local_sense);
        } else {
            long spins = 0;
            while (atomicAdd(sense, 0u) != local_sense) { if (++spins > 10000000) { ok = false; break; } __builtin_amdgcn_s_sleep(1); }
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}

__global__ void k_own(int iters, double* data, unsigned* counter, unsigned* sense, int* failed) {
    __shared__ unsigned s_local;
    if (threadIdx.x == 0) s_local = 0;
    __syncthreads();
    unsigned local = 0;
    double v = data[blockIdx.x * blockDim.x + threadIdx.x];
    for (int it = 0; it < iters; ++it) {
        v = v * 1.0000001 + 1.0;
        data[blockIdx.x * blockDim.x + threadIdx.x] = v;
        if (threadIdx.x == 0) local = s_local;
        bool ok = bar(counter, sense, local, gridDim.x);
        if (threadIdx.x == 0) { s_local = local; if (!ok) *failed = 1; }
        __syncthreads();
        if (*failed) return;
        v += data[((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x] * 1e-9;
    }
    data[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

int main() {
    const int threads = 256;
    for (int blocks : {64, 128, 256, 512}) {
        double* data; unsigned *counter, *sense; int* failed;
        CK(hipMalloc(&data, sizeof(double) * blocks * threads)); CK(hipMemset(data, 0, sizeof(double) * blocks * threads));
        CK(hipMalloc(&counter, 4)); CK(hipMalloc(&sense, 4)); CK(hipMalloc(&failed, 4));
        CK(hipMemset(counter, 0, 4)); CK(hipMemset(sense, 0, 4)); CK(hipMemset(failed, 0, 4));
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int iters : {1, 1001}) {
            int it = iters; void* args[] = {&it, &data};
            CK(hipEventRecord(e0));
            hipError_t err = hipLaunchCooperativeKernel((const void*)k_cg, dim3(blocks), dim3(threads), args, 0, nullptr);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("cooperative_groups grid.sync  blocks %3d iters %4d: %8.1f us total (%s)\n", blocks, iters, ms * 1e3, hipGetErrorString(err));
        }
        for (int iters : {1, 1001}) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_own, dim3(blocks), dim3(threads), 0, nullptr, iters, data, counter, sense, failed);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            int f = 0; CK(hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost));
            printf("hand-rolled atomic barrier    blocks %3d iters %4d: %8.1f us total (failed %d)\n", blocks, iters, ms * 1e3, f);
        }
        CK(hipFree(data)); CK(hipFree(counter)); CK(hipFree(sense)); CK(hipFree(failed));
    }
    return 0;
}
