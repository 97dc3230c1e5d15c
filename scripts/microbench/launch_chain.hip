// Microbenchmark: time per kernel in a chain of dependent kernels on one stream on gfx950, launched (a) one by
// one from the host and (b) as one hipGraph of the same kernels, for grids of 1 / 40 / 320 workgroups and for
// kernels with 0 / 1 / 3 dependent global-memory round trips.  Decides whether capturing the pivot loop (three
// dependent launches per pivot) in a hipGraph would shorten the pivot.
// Build: hipcc --offload-arch=gfx950 -O3 launch_chain.hip -o launch_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// `hops` dependent loads (pointer chase through idx), then one store that the next kernel reads
__global__ void k_link(const int* __restrict__ idx, double* data, int hops) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int j = i;
    for (int h = 0; h < hops; ++h) j = idx[j];
    data[i] = data[j] + 1.0;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    const int kChain = 192, kReps = 20;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int max_threads = 320 * 256;
    int* idx; double* data;
    CK(hipMalloc(&idx, max_threads * sizeof(int)));
    CK(hipMalloc(&data, max_threads * sizeof(double)));
    int* h = new int[max_threads];
    for (int i = 0; i < max_threads; ++i) h[i] = (int)(((long)i * 7919 + 13) % max_threads);
    CK(hipMemcpy(idx, h, max_threads * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemset(data, 0, max_threads * sizeof(double)));
    CK(hipDeviceSynchronize());
    const int grids[] = {1, 40, 320};
    const int hops_list[] = {0, 1, 3, 8};   // 8 hops: a ~6 us kernel, the host enqueue runs ahead of the GPU
    for (int grid : grids) for (int hops : hops_list) {
        // (a) host launches
        for (int w = 0; w < 32; ++w) hipLaunchKernelGGL(k_link, dim3(grid), dim3(256), 0, s, idx, data, hops);
        CK(hipStreamSynchronize(s));
        double t0 = now_us();
        for (int rep = 0; rep < kReps; ++rep)
            for (int k = 0; k < kChain; ++k) hipLaunchKernelGGL(k_link, dim3(grid), dim3(256), 0, s, idx, data, hops);
        double t_enq = now_us() - t0;
        CK(hipStreamSynchronize(s));
        double t_stream = now_us() - t0;
        // (b) the same chain captured once, replayed
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int k = 0; k < kChain; ++k) hipLaunchKernelGGL(k_link, dim3(grid), dim3(256), 0, s, idx, data, hops);
        CK(hipStreamEndCapture(s, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        CK(hipGraphLaunch(exec, s));
        CK(hipStreamSynchronize(s));
        t0 = now_us();
        for (int rep = 0; rep < kReps; ++rep) CK(hipGraphLaunch(exec, s));
        CK(hipStreamSynchronize(s));
        double t_graph = now_us() - t0;
        CK(hipGraphExecDestroy(exec));
        CK(hipGraphDestroy(graph));
        const double n = (double)kChain * kReps;
        printf("grid %3d hops %d: stream %.2f us/kernel (host enqueue %.2f us/launch), hipGraph %.2f us/kernel\n", grid, hops,
               t_stream / n, t_enq / n, t_graph / n);
        fflush(stdout);
    }
    return 0;
}
