// host stand-in for <hip/hip_runtime.h>: enough of the HIP surface to compile the elementwise kernels of libsdmi as plain C++ and run
// them thread by thread on the CPU (no barriers, no cross-lane traffic: kernels that need them are not emulated)
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__
#define __restrict__
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipMemcpyDeviceToDevice = 3, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "host emulation"; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n); return hipSuccess; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline void __syncthreads() {}
inline int __any(int p) { return p; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
template <typename T> inline T __shfl_xor(T v, int) { return v; }
template <typename T> inline T __shfl_down(T v, int) { return v; }
template <typename T> inline T __shfl(T v, int) { return v; }
inline float __expf(float x) { return std::exp(x); }
inline float __logf(float x) { return std::log(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
inline unsigned __umul24(unsigned a, unsigned b) { return a * b; }
template <typename T> inline T min(T a, T b) { return a < b ? a : b; }
template <typename T> inline T max(T a, T b) { return a > b ? a : b; }
// every thread of every block, one after the other
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                                              \
    do {                                                                                                          \
        const dim3 g_ = (grid), b_ = (block);                                                                     \
        gridDim = g_; blockDim = b_;                                                                              \
        for (unsigned bz = 0; bz < g_.z; ++bz) for (unsigned by = 0; by < g_.y; ++by) for (unsigned bx = 0; bx < g_.x; ++bx) \
            for (unsigned tz = 0; tz < b_.z; ++tz) for (unsigned ty = 0; ty < b_.y; ++ty) for (unsigned tx = 0; tx < b_.x; ++tx) { \
                blockIdx = dim3(bx, by, bz); threadIdx = dim3(tx, ty, tz);                                        \
                kernel(__VA_ARGS__);                                                                              \
            }                                                                                                     \
    } while (0)
