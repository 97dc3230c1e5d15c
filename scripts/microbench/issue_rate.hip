// issue_rate.hip -- what one wavefront pays per instruction on an otherwise idle CU (s_memtime ticks): dependent f64 FMA
// chain, independent f64 FMAs, dependent LDS round trips (ds_read_b64 -> address), DPP row shifts, uniform branches.
// Build & run on the GPU box: hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o issue_rate && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_probe(long long* out, double seed, int n, int waves_active) {
    __shared__ double lds[4096];
    __shared__ int chain[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += blockDim.x) { lds[i] = seed + i; chain[i] = (i * 97 + 13) & 4095; }
    __syncthreads();
    if ((tid >> 6) >= waves_active) return;
    long long t0, t1;
    // 1. dependent f64 FMA chain
    double a = seed + tid;
    t0 = clock64();
    for (int i = 0; i < n; ++i) a = fma(a, 1.0000001, 0.5);
    t1 = clock64();
    if (tid == 0) out[0] = t1 - t0;
    // 2. four independent chains
    double b0 = a, b1 = a + 1, b2 = a + 2, b3 = a + 3;
    t0 = clock64();
    for (int i = 0; i < n; ++i) { b0 = fma(b0, 1.0000001, 0.5); b1 = fma(b1, 1.0000001, 0.5); b2 = fma(b2, 1.0000001, 0.5); b3 = fma(b3, 1.0000001, 0.5); }
    t1 = clock64();
    if (tid == 0) out[1] = t1 - t0;
    // 3. dependent LDS round trips
    int p = tid & 4095;
    t0 = clock64();
    for (int i = 0; i < n; ++i) p = chain[p];
    t1 = clock64();
    if (tid == 0) out[2] = t1 - t0;
    // 4. LDS gather of doubles feeding an FMA chain (load -> fma -> address)
    double acc = b0 + b1 + b2 + b3;
    t0 = clock64();
    for (int i = 0; i < n; ++i) { acc = fma(lds[p], 0.999, acc); p = (p + 61) & 4095; }
    t1 = clock64();
    if (tid == 0) out[3] = t1 - t0;
    // 5. uniform branches: a ladder of scalar compares like the reduction-width switch
    int sel = n & 7;
    double c = acc;
    t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (sel >= 6) c += 1.0;
        if (sel >= 5) c += 2.0;
        if (sel >= 4) c += 3.0;
        if (sel >= 3) c += 4.0;
        if (sel >= 2) c += 5.0;
        if (sel >= 1) c += 6.0;
        sel = (sel + 1) & 7;
    }
    t1 = clock64();
    if (tid == 0) out[4] = t1 - t0;
    // 6. workgroup barriers
    t0 = clock64();
    t1 = clock64();
    if (tid == 0) { out[5] = t1 - t0; out[6] = (long long)(a + c + p); }
}

int main() {
    long long* d; hipMalloc(&d, 64);
    const int n = 4096;
    for (int threads : {64, 256, 512}) {
        for (int wa : {1, threads / 64}) {
            hipMemset(d, 0, 64);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k_probe, dim3(1), dim3(threads), 0, 0, d, 1.0, n, wa);   // warm
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_probe, dim3(1), dim3(threads), 0, 0, d, 1.0, n, wa);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
            long long total = h[0] + h[1] + h[2] + h[3] + h[4];
            printf("threads %3d, waves running %d: dependent fma %.1f, 4 independent fma %.1f (per 4), LDS chain %.1f, gather+fma %.1f, branch ladder %.1f ticks per iteration; "
                   "memtime pair %lld; kernel %.3f ms for %lld ticks -> %.2f GHz if ticks are cycles\n",
                   threads, wa, (double)h[0] / n, (double)h[1] / n, (double)h[2] / n, (double)h[3] / n, (double)h[4] / n, h[5], ms, total, total / (ms * 1e6));
        }
    }
    return 0;
}
