// microbenchmark: dependent-load latency under the occupancy pattern of k_tiles (diagnostics only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void chase(const unsigned* __restrict__ buf, unsigned n, int hops, unsigned* out, int lds_pad) {
    extern __shared__ unsigned pad[];
    unsigned i = (blockIdx.x * 64 + threadIdx.x) % n;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int h = 0; h < hops; ++h) i = buf[i];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lds_pad && threadIdx.x == 0) pad[0] = i;
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = i; out[blockIdx.x * 2 + 1] = (unsigned)(t1 - t0); }
}
int main() {
    const unsigned n = 1 << 18;  // 1 MB table
    std::vector<unsigned> h(n);
    for (unsigned i = 0; i < n; ++i) h[i] = (i * 2654435761u + 12345u) % n;
    unsigned *d, *o; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    const int blocks = 8100; hipMalloc(&o, blocks * 8);
    for (int lds : {0, 10 * 1024, 17 * 1024}) for (int hops : {1, 8, 32}) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), lds, 0, d, n, hops, o, lds);
        hipDeviceSynchronize();
        hipEventRecord(a); 
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), lds, 0, d, n, hops, o, lds);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned> ho(blocks * 2); hipMemcpy(ho.data(), o, blocks * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int k = 0; k < blocks; ++k) s += ho[2 * k + 1];
        printf("lds %5d B hops %2d: kernel %.1f us, mean wave clocks %.0f => %.0f clocks/hop\n", lds, hops, ms / 20 * 1e3, s / blocks, s / blocks / hops);
    }
    return 0;
}
