// Development aid, NOT part of the product: a stand-in for <hip/hip_runtime.h> that lets the kernels of
// swf_renderer_amd/csrc compile as plain C++ (g++) and run on the CPU in a lock-step wavefront emulator
// (tools/emu/emu_rt.cpp): every work-item is a fiber, cross-lane operations (__ballot, __shfl, DPP, readlane,
// __syncthreads) are rendezvous points among the lanes that reach them.  Used to debug kernels and to run them
// under the CPU sanitizers where no GPU is at hand.  libswfr.so never contains any of this; the library built
// here (tools/emu/libswfr_emu.so) is only ever loaded by tools/emu/run.py.
#pragma once
#include <limits.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#define SWFR_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline uint32_t __umul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline int32_t __mul24(int32_t a, int32_t b) { return (int32_t)(((int32_t)(a << 8) >> 8) * (int64_t)((int32_t)(b << 8) >> 8)); }
struct int4 { int32_t x, y, z, w; };
static inline int4 make_int4(int32_t x, int32_t y, int32_t z, int32_t w) { return int4{x, y, z, w}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

namespace emu {
struct Idx { unsigned x, y, z; };
extern Idx thread_idx, block_idx, block_dim, grid_dim;
enum Kind { K_SYNC = 1, K_BALLOT, K_SHFL, K_SHFL_XOR, K_DPP, K_READLANE, K_READFIRST };
uint64_t collective(int kind, uint64_t a, uint64_t b, uint64_t c, uint64_t d);
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
unsigned lane_id();
}  // namespace emu

#define threadIdx (emu::thread_idx)
#define blockIdx (emu::block_idx)
#define blockDim (emu::block_dim)
#define gridDim (emu::grid_dim)

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })

// ---- cross-lane operations ------------------------------------------------------------------------------------------
static inline void __syncthreads() { emu::collective(emu::K_SYNC, 0, 0, 0, 0); }
static inline unsigned long long __ballot(int pred) { return emu::collective(emu::K_BALLOT, pred ? 1 : 0, 0, 0, 0); }
static inline int __shfl(int v, int src) { return (int)(uint32_t)emu::collective(emu::K_SHFL, (uint32_t)v, (uint32_t)src, 0, 0); }
static inline int __shfl_xor(int v, int mask) { return (int)(uint32_t)emu::collective(emu::K_SHFL_XOR, (uint32_t)v, (uint32_t)mask, 0, 0); }
static inline int emu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    return (int)(uint32_t)emu::collective(emu::K_DPP, (uint32_t)old, (uint32_t)src, (uint32_t)ctrl,
                                          (uint32_t)row_mask | ((uint32_t)bank_mask << 8) | ((uint32_t)bound_ctrl << 16));
}
static inline int emu_readlane(int v, int lane) { return (int)(uint32_t)emu::collective(emu::K_READLANE, (uint32_t)v, (uint32_t)lane, 0, 0); }
static inline int emu_readfirstlane(int v) { return (int)(uint32_t)emu::collective(emu::K_READFIRST, (uint32_t)v, 0, 0, 0); }
static inline uint32_t emu_mbcnt_lo(uint32_t mask, uint32_t add) { const unsigned l = emu::lane_id(); return add + (uint32_t)__builtin_popcount(l >= 32 ? mask : (mask & ((1u << l) - 1u))); }
static inline uint32_t emu_mbcnt_hi(uint32_t mask, uint32_t add) { const unsigned l = emu::lane_id(); return add + (l <= 32 ? 0u : (uint32_t)__builtin_popcount(mask & ((1u << (l - 32)) - 1u))); }
unsigned long long emu_clock();
#define __builtin_amdgcn_update_dpp emu_update_dpp
#define __builtin_amdgcn_readlane emu_readlane
#define __builtin_amdgcn_readfirstlane emu_readfirstlane
#define __builtin_amdgcn_mbcnt_lo emu_mbcnt_lo
#define __builtin_amdgcn_mbcnt_hi emu_mbcnt_hi
#define __builtin_amdgcn_s_memtime emu_clock
#define __builtin_amdgcn_s_memrealtime emu_clock
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)

static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline double __dsqrt_rn(double v) { return sqrt(v); }
template <class T> static inline T min(T a, T b) { return a < b ? a : b; }
template <class T> static inline T max(T a, T b) { return a > b ? a : b; }
static inline long long min(long long a, int b) { return a < b ? a : b; }
static inline long long max(long long a, int b) { return a > b ? a : b; }

// ---- atomics: fibers are cooperative, plain read-modify-write is atomic -----------------------------------------------
template <class T, class U> static inline T atomicAdd(T* p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class U> static inline T atomicOr(T* p, U v) { T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class U> static inline T atomicAnd(T* p, U v) { T o = *p; *p = (T)(o & (T)v); return o; }
template <class T, class U> static inline T atomicMin(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicExch(T* p, U v) { T o = *p; *p = (T)v; return o; }

// ---- host runtime: everything is synchronous host memory --------------------------------------------------------------
typedef int hipError_t;
typedef struct emu_stream* hipStream_t;
typedef struct emu_event* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipEventDisableTiming = 2 };
// graphs: the emulator executes launches eagerly, so "capturing" a frame runs it and launching the graph runs nothing more -- the
// product is therefore told (SWFR_GRAPHS defaults to 0 under SWFR_EMU) not to use them
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
typedef void* hipGraphNode_t;
enum { hipStreamCaptureModeRelaxed = 2 };
static inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return hipErrorInvalidValue; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorInvalidValue; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, hipGraphNode_t*, char*, size_t) { *e = nullptr; return hipErrorInvalidValue; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorInvalidValue; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, int kind);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int kind, hipStream_t st);
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, int kind, hipStream_t st);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t st);
hipError_t hipStreamSynchronize(hipStream_t st);
hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t e, unsigned flags);
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
