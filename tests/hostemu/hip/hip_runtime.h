// Host stand-in for <hip/hip_runtime.h> (TEST INFRASTRUCTURE, tests/test_cpu_kernel_emulation.py): enough of the HIP surface to compile
// the elementwise and normalisation kernels of libsdmi as plain C++ and run them on the CPU.
//   sequential mode (default)  every thread of every block runs to completion in turn: exact for kernels without workgroup barriers or
//                              cross-lane traffic (__syncthreads is a no-op, shuffles return the caller's own value)
//   threaded mode (emu_set_threaded(1))  one OS thread per thread of the running block; __syncthreads is a barrier over the block,
//                              __shfl_xor / __shfl_down / __shfl exchange through a per-wavefront (64 lanes) slot array between two
//                              barriers over the wave — every lane of a wave must reach the shuffle, as on the hardware when EXEC is full.
// `__shared__` is function-static storage: blocks run one after the other, so a block's threads share it and the next block reuses it.
// The gfx950 builtins of the MFMA kernels (gemm.hip, attention.hip: the two MFMA shapes, LDS-DMA, the permlane swaps, ballot) are
// emulated per wave in threaded mode; LDS-DMA completes at issue, so nothing here models the hardware's asynchrony.
#pragma once
#include <pthread.h>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipMemcpyDeviceToDevice = 3, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "host emulation"; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n); return hipSuccess; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }

struct EmuBlock {                                              // the running block in threaded mode
    pthread_barrier_t all;
    pthread_barrier_t wave[16];
    uint64_t slot[16][64];
    unsigned wave_lanes[16];
    alignas(16) unsigned char ma[16][64][16], mb[16][64][16];   // MFMA operand fragments of a wave's 64 lanes
};
extern thread_local EmuBlock* emu_block;                       // null in sequential mode
extern thread_local unsigned emu_tid;                          // linear thread id inside the block
extern int emu_threaded;

inline void __syncthreads() { if (emu_block) pthread_barrier_wait(&emu_block->all); }
template <typename T> inline T emu_exchange(T v, unsigned src_lane_of_me) {
    static_assert(sizeof(T) <= 8, "shuffles move at most 8 bytes");
    if (!emu_block) return v;
    const unsigned w = emu_tid >> 6, lane = emu_tid & 63;
    std::memcpy(&emu_block->slot[w][lane], &v, sizeof v);
    pthread_barrier_wait(&emu_block->wave[w]);
    T r;
    std::memcpy(&r, &emu_block->slot[w][src_lane_of_me & 63], sizeof r);
    pthread_barrier_wait(&emu_block->wave[w]);
    return r;
}
inline int __any(int p) {                                      // wave-wide OR (threaded mode; sequentially a lane only sees itself)
    if (!emu_block) return p;
    const unsigned w = emu_tid >> 6, lane = emu_tid & 63;
    emu_block->slot[w][lane] = p ? 1u : 0u;
    pthread_barrier_wait(&emu_block->wave[w]);
    uint64_t any = 0;
    for (unsigned l = 0; l < emu_block->wave_lanes[w]; ++l) any |= emu_block->slot[w][l];
    pthread_barrier_wait(&emu_block->wave[w]);
    return any != 0;
}
template <typename T> inline T __shfl_xor(T v, int m) { return emu_exchange(v, (emu_tid & 63) ^ (unsigned)m); }
template <typename T> inline T __shfl_down(T v, int d) { return emu_exchange(v, ((emu_tid & 63) + (unsigned)d) > 63 ? (emu_tid & 63) : (emu_tid & 63) + (unsigned)d); }
template <typename T> inline T __shfl(T v, int l) { return emu_exchange(v, (unsigned)l); }
// ---- gfx950 builtins of the MFMA kernels (threaded mode only) -----------------------------------------------------------------------
struct uint4 { unsigned x, y, z, w; };
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
typedef _Float16 emu_h8 __attribute__((ext_vector_type(8)));
typedef float emu_f4 __attribute__((ext_vector_type(4)));
typedef unsigned emu_u2 __attribute__((ext_vector_type(2)));
// v_mfma_f32_16x16x32_f16: D (16 x 16) = A (16 x 32) B (32 x 16) + C.  Lane l supplies A[l % 16][8 (l / 16) .. + 7] and
// B[8 (l / 16) .. + 7][l % 16] and owns D[4 (l / 16) + r][l % 16], r = 0 .. 3.  fp32 accumulation in k order.
inline emu_f4 __builtin_amdgcn_mfma_f32_16x16x32_f16(emu_h8 a, emu_h8 b, emu_f4 c, int, int, int) {
    EmuBlock* blk = emu_block;
    if (!blk) std::abort();                                   // a wave-wide operation: threaded mode only
    const unsigned w = emu_tid >> 6, lane = emu_tid & 63;
    std::memcpy(blk->ma[w][lane], &a, 16);
    std::memcpy(blk->mb[w][lane], &b, 16);
    pthread_barrier_wait(&blk->wave[w]);
    const unsigned col = lane & 15;
    emu_f4 d = c;
    for (int r = 0; r < 4; ++r) {
        const unsigned row = 4 * (lane >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            _Float16 av, bv;
            std::memcpy(&av, blk->ma[w][(k >> 3) * 16 + row] + 2 * (k & 7), 2);
            std::memcpy(&bv, blk->mb[w][(k >> 3) * 16 + col] + 2 * (k & 7), 2);
            acc += (float)av * (float)bv;
        }
        d[r] = acc;
    }
    pthread_barrier_wait(&blk->wave[w]);
    return d;
}
// v_mfma_f32_32x32x16_f16: D (32 x 32) = A (32 x 16) B (16 x 32) + C.  Lane l supplies A[l % 32][8 (l / 32) .. + 7] and
// B[8 (l / 32) .. + 7][l % 32] and owns D[8 (r / 4) + 4 (l / 32) + r % 4][l % 32], r = 0 .. 15.
typedef float emu_f16v __attribute__((ext_vector_type(16)));
inline emu_f16v __builtin_amdgcn_mfma_f32_32x32x16_f16(emu_h8 a, emu_h8 b, emu_f16v c, int, int, int) {
    EmuBlock* blk = emu_block;
    if (!blk) std::abort();
    const unsigned w = emu_tid >> 6, lane = emu_tid & 63;
    std::memcpy(blk->ma[w][lane], &a, 16);
    std::memcpy(blk->mb[w][lane], &b, 16);
    pthread_barrier_wait(&blk->wave[w]);
    const unsigned col = lane & 31;
    emu_f16v d = c;
    for (int r = 0; r < 16; ++r) {
        const unsigned row = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            _Float16 av, bv;
            std::memcpy(&av, blk->ma[w][(k >> 3) * 32 + row] + 2 * (k & 7), 2);
            std::memcpy(&bv, blk->mb[w][(k >> 3) * 32 + col] + 2 * (k & 7), 2);
            acc += (float)av * (float)bv;
        }
        d[r] = acc;
    }
    pthread_barrier_wait(&blk->wave[w]);
    return d;
}
// v_permlane32_swap_b32 x, y (inline assembly in attention.hip, replaced textually by this call): the upper 32 lanes of x swap with the
// lower 32 lanes of y
inline void emu_permlane32_swap(float& x, float& y) {
    EmuBlock* blk = emu_block;
    if (!blk) std::abort();
    const unsigned w = emu_tid >> 6, lane = emu_tid & 63;
    uint32_t xb, yb;
    std::memcpy(&xb, &x, 4); std::memcpy(&yb, &y, 4);
    blk->slot[w][lane] = ((uint64_t)yb << 32) | xb;
    pthread_barrier_wait(&blk->wave[w]);
    const uint64_t other = blk->slot[w][lane ^ 32];
    pthread_barrier_wait(&blk->wave[w]);
    if (lane >= 32) xb = (uint32_t)(other >> 32);              // upper half of x := lower half of y
    else yb = (uint32_t)other;                                 // lower half of y := upper half of x
    std::memcpy(&x, &xb, 4); std::memcpy(&y, &yb, 4);
}
inline uint64_t __builtin_amdgcn_ballot_w64(bool p) {
    EmuBlock* blk = emu_block;
    if (!blk) return p ? 1 : 0;
    const unsigned w = emu_tid >> 6, lane = emu_tid & 63;
    blk->slot[w][lane] = p ? 1u : 0u;
    pthread_barrier_wait(&blk->wave[w]);
    uint64_t m = 0;
    for (unsigned l = 0; l < blk->wave_lanes[w]; ++l) m |= blk->slot[w][l] << l;
    pthread_barrier_wait(&blk->wave[w]);
    return m;
}
// global_load_lds (LDS-DMA): lane l copies `size` bytes from ITS global pointer to (wave-uniform LDS base) + offset + l * size; here the
// copy completes at issue (the kernels wait with s_waitcnt before they read, and never issue into a buffer that is still being read)
inline void __builtin_amdgcn_global_load_lds(const void* g, void* lds, unsigned size, unsigned offset, unsigned) {
    std::memcpy((char*)lds + offset + (emu_tid & 63) * size, g, size);
}
inline void __builtin_amdgcn_s_barrier() { if (emu_block) pthread_barrier_wait(&emu_block->all); }
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_s_setprio(int) {}
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }  // (the kernels pass wave-uniform values)
inline long long __builtin_amdgcn_s_memtime() { return 0; }
inline float __builtin_amdgcn_exp2f(float x) { return std::exp2(x); }
// v_permlane16_swap: with the wave as four rows of 16 lanes, the odd rows of the first operand swap with the even rows of the second:
// x' = [x.row0, y.row0, x.row2, y.row2], y' = [x.row1, y.row1, x.row3, y.row3]; returns (x', y')
inline emu_u2 __builtin_amdgcn_permlane16_swap(unsigned x, unsigned y, bool, bool) {
    EmuBlock* blk = emu_block;
    if (!blk) std::abort();
    const unsigned w = emu_tid >> 6, lane = emu_tid & 63, row = lane >> 4;
    blk->slot[w][lane] = ((uint64_t)y << 32) | x;
    pthread_barrier_wait(&blk->wave[w]);
    const uint64_t below = blk->slot[w][(lane + 48) & 63], above = blk->slot[w][(lane + 16) & 63];      // lanes one row down / up
    const unsigned xn = (row & 1) ? (unsigned)(below >> 32) : x;             // odd row of x' = the y of the row below
    const unsigned yn = (row & 1) ? y : (unsigned)above;                     // even row of y' = the x of the row above
    pthread_barrier_wait(&blk->wave[w]);
    return emu_u2{xn, yn};
}
inline float __expf(float x) { return std::exp(x); }
inline float __logf(float x) { return std::log(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
inline unsigned __umul24(unsigned a, unsigned b) { return a * b; }
template <typename T> inline T min(T a, T b) { return a < b ? a : b; }
template <typename T> inline T max(T a, T b) { return a > b ? a : b; }

inline void emu_launch(dim3 g, dim3 b, const std::function<void()>& body) {
    const unsigned nt = b.x * b.y * b.z;
    for (unsigned bz = 0; bz < g.z; ++bz) for (unsigned by = 0; by < g.y; ++by) for (unsigned bx = 0; bx < g.x; ++bx) {
        if (!emu_threaded) {
            gridDim = g; blockDim = b; blockIdx = dim3(bx, by, bz);
            for (unsigned tz = 0; tz < b.z; ++tz) for (unsigned ty = 0; ty < b.y; ++ty) for (unsigned tx = 0; tx < b.x; ++tx) {
                threadIdx = dim3(tx, ty, tz);
                body();
            }
            continue;
        }
        EmuBlock blk;
        pthread_barrier_init(&blk.all, nullptr, nt);
        const unsigned nw = (nt + 63) / 64;
        for (unsigned w = 0; w < nw; ++w) { blk.wave_lanes[w] = std::min(64u, nt - w * 64); pthread_barrier_init(&blk.wave[w], nullptr, blk.wave_lanes[w]); }
        std::vector<std::thread> ts;
        ts.reserve(nt);
        for (unsigned t = 0; t < nt; ++t)
            ts.emplace_back([&, t]() {
                gridDim = g; blockDim = b; blockIdx = dim3(bx, by, bz);
                threadIdx = dim3(t % b.x, (t / b.x) % b.y, t / (b.x * b.y));
                emu_block = &blk; emu_tid = t;
                body();
            });
        for (auto& th : ts) th.join();
        pthread_barrier_destroy(&blk.all);
        for (unsigned w = 0; w < nw; ++w) pthread_barrier_destroy(&blk.wave[w]);
    }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch((grid), (block), [&]() { kernel(__VA_ARGS__); })
