// store_shape.cpp -- how fast can 16200 wavefronts write a 3840x2160 RGBA8 frame, 2 KB each, as a function of the SHAPE of a
// wavefront's block: 512x1, 256x2, 128x4, 64x8 pixels (one 16-byte store per lane and row group)?  The tile pass of swfr writes 64x8.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/store_shape.cpp -o build/store_shape ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

template <int BW, int BH>      // block of BW x BH pixels per wavefront, BW * BH = 512
__global__ __launch_bounds__(64) void k_store(uint32_t* fb, int width, int height, uint32_t colour) {
    const int lane = threadIdx.x;
    const int blocks_x = width / BW;
    const int bx = blockIdx.x % blocks_x, by = blockIdx.x / blocks_x;
    constexpr int LANES_PER_ROW = BW / 4;                 // 16 bytes = 4 pixels per lane
    constexpr int ROWS_PER_STORE = 64 / LANES_PER_ROW > 0 ? 64 / LANES_PER_ROW : 1;
    constexpr int STORES = BH / ROWS_PER_STORE > 0 ? BH / ROWS_PER_STORE : (BW * BH) / 256;
    if (LANES_PER_ROW <= 64) {
        const int lr = lane / LANES_PER_ROW, lc = lane % LANES_PER_ROW;
#pragma unroll
        for (int s = 0; s < STORES; ++s) {
            const int y = by * BH + lr + s * ROWS_PER_STORE, x = bx * BW + 4 * lc;
            if (y < height) *reinterpret_cast<uint4*>(fb + (size_t)y * width + x) = make_uint4(colour, colour + lane, colour, colour);
        }
    } else {                                              // 512x1: two stores along the row
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int y = by * BH, x = bx * BW + 256 * s + 4 * lane;
            if (y < height) *reinterpret_cast<uint4*>(fb + (size_t)y * width + x) = make_uint4(colour, colour + lane, colour, colour);
        }
    }
}

template <int BW, int BH>
static void run(const char* name, uint32_t* fb, int W, int H) {
    const int grid = (W / BW) * ((H + BH - 1) / BH);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_store<BW, BH>), dim3(grid), dim3(64), 0, 0, fb, W, H, 0x01020304u + i);
    CK(hipEventRecord(a, 0));
    const int N = 50;
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL((k_store<BW, BH>), dim3(grid), dim3(64), 0, 0, fb, W, H, 0x01020304u + i);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / N, gbs = 4.0 * W * H / (us * 1e-6) / 1e9;
    std::printf("%-8s grid %6d  %7.2f us per frame  %8.1f GB/s  (%.3f of 8 TB/s)\n", name, grid, us, gbs, gbs / 8000.0);
}

int main() {
    const int W = 3840, H = 2160;
    uint32_t* fb; CK(hipMalloc(&fb, (size_t)W * H * 4));
    run<64, 8>("64x8", fb, W, H);
    run<128, 4>("128x4", fb, W, H);
    run<256, 2>("256x2", fb, W, H);
    run<512, 1>("512x1", fb, W, H);
    run<64, 8>("64x8", fb, W, H);
    CK(hipMemset(fb, 1, (size_t)W * H * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < 50; ++i) CK(hipMemsetAsync(fb, i, (size_t)W * H * 4, 0));
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    std::printf("memset            %7.2f us per frame  %8.1f GB/s\n", ms * 1e3 / 50, 4.0 * W * H / (ms * 1e-3 / 50) / 1e9);
    return 0;
}
