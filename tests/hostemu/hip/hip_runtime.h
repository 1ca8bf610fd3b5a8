// Host stand-in for <hip/hip_runtime.h> (TEST INFRASTRUCTURE, tests/test_cpu_kernel_emulation.py): enough of the HIP surface to compile
// the elementwise and normalisation kernels of libsdmi as plain C++ and run them on the CPU.
//   sequential mode (default)  every thread of every block runs to completion in turn: exact for kernels without workgroup barriers or
//                              cross-lane traffic (__syncthreads is a no-op, shuffles return the caller's own value)
//   auto mode (emu_set_threaded(2))      a block starts sequentially; the first barrier / cross-lane operation it meets makes the launch
//                              site threaded from then on (the block is re-run: kernels write nothing before their first barrier that a re-run
//                              does not rewrite identically).  What the whole-library build (tests/test_cpu_emulated_library.py) runs in.
//   threaded mode (emu_set_threaded(1))  every thread of the running block is a fiber (its own stack, switched in user space by one OS thread,
//                              round-robin, deterministic); __syncthreads is a barrier over the block,
//                              __shfl_xor / __shfl_down / __shfl exchange through a per-wavefront (64 lanes) slot array between two
//                              barriers over the wave — every lane of a wave must reach the shuffle, as on the hardware when EXEC is full.
// `__shared__` is function-static storage: blocks run one after the other, so a block's threads share it and the next block reuses it.
// The gfx950 builtins of the MFMA kernels (gemm.hip, attention.hip: the two MFMA shapes, LDS-DMA, the permlane swaps, ballot) are
// emulated per wave in threaded mode; LDS-DMA completes at issue, so nothing here models the hardware's asynchrony.
#pragma once
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
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
struct EmuEvent { double ms; };
typedef EmuEvent* hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipMemcpyDeviceToDevice = 3, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8, hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
struct hipDeviceProp_t { char gcnArchName[64]; int multiProcessorCount; };
// "device" memory is host memory, every stream is the one in-order host thread: a launch has completed when it returns
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "host emulation"; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
// SDMI_HOSTEMU_POISON=1: fresh "device" memory is filled with 0xFF (fp16 / fp32 NaN, huge integers) instead of whatever the allocator
// returns — a result that depends on uninitialised device memory (fresh GPU pages are often zero, which hides it) turns into NaN
inline hipError_t hipMalloc(void** p, size_t n) {
    const size_t bytes = (n + 255) / 256 * 256 + 256;
    *p = std::aligned_alloc(256, bytes);
    if (!*p) return 2;
    static const bool poison = [] { const char* e = std::getenv("SDMI_HOSTEMU_POISON"); return e && e[0] == '1'; }();
    if (poison) std::memset(*p, 0xFF, bytes);
    return hipSuccess;
}
template <typename T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
// the stand-in "device" calls itself gfx950 only inside a test process that asked for the emulation (SDMI_HOSTEMU=1, the switch of
// tests/conftest.py and tests/hostemu/run.py): pointed at this library by SDMI_LIB alone, the product's require_device() still refuses
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    const char* e = std::getenv("SDMI_HOSTEMU");
    std::strcpy(p->gcnArchName, e && e[0] == '1' ? "gfx950:hostemu" : "hostemu");
    p->multiProcessorCount = 256;
    return hipSuccess;
}
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new EmuEvent{0.0}; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    e->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->ms - a->ms); return hipSuccess; }

#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
extern "C" void __tsan_acquire(void*);
extern "C" void __tsan_release(void*);
#define EMU_TSAN_RELEASE(a) __tsan_release(a)
#define EMU_TSAN_ACQUIRE(a) __tsan_acquire(a)
#endif
#endif
#define EMU_RUNTIME __attribute__((no_sanitize("thread")))   // the emulator's own bookkeeping is not what a sanitizer build looks at
#ifndef EMU_TSAN_RELEASE
#define EMU_TSAN_RELEASE(a) ((void)0)
#define EMU_TSAN_ACQUIRE(a) ((void)0)
#endif
struct EmuBlock {                                              // the running block in threaded mode
    unsigned wave_lanes[16];
    // barriers: ids 0 .. 15 = the waves, 16 = the block.  `need` falls when a thread returns from the kernel, as on the hardware
    // (a finished wave no longer takes part in s_barrier)
    unsigned arrived[17], need[17], gen[17];
    char tsan_sync[17];
    unsigned short wait_id[1024];                              // what a parked thread waits for: the scheduler resumes it when gen[wait_id] has moved on
    unsigned wait_gen[1024];
    uint64_t slot[16][64];
    // MFMA operand fragments of a wave's 64 lanes, double-buffered by the parity of the lane's MFMA count: ONE barrier per MFMA (a lane
    // can only write buffer p again after the barrier of its next MFMA, which every lane reaches after it has read buffer p)
    alignas(16) unsigned char ma[2][16][64][16], mb[2][16][64][16];
    unsigned char mfma_parity[1024];
};
extern thread_local EmuBlock* emu_block;                       // null in sequential mode
extern thread_local unsigned emu_tid;                          // linear thread id inside the block
void emu_yield();                                              // to the block's scheduler (tests/hostemu/emu.cpp)
extern int emu_threaded;                                      // 0 sequential, 1 threaded, 2 auto
struct EmuNeedThreads {};
EMU_RUNTIME inline void emu_barrier(unsigned id) {                        // the last arrival releases the others; waiting = yielding to the other threads
    EmuBlock* b = emu_block;
    const unsigned g = b->gen[id];
    EMU_TSAN_RELEASE(&b->tsan_sync[id]);                       // (thread-sanitizer builds: every arrival happens-before every departure)
    if (++b->arrived[id] >= b->need[id]) { b->arrived[id] = 0; ++b->gen[id]; EMU_TSAN_ACQUIRE(&b->tsan_sync[id]); return; }
    b->wait_id[emu_tid] = (unsigned short)id; b->wait_gen[emu_tid] = g;
    while (b->gen[id] == g) emu_yield();
    EMU_TSAN_ACQUIRE(&b->tsan_sync[id]);
}
EMU_RUNTIME inline bool emu_lone() {                                      // true: no running block (sequential execution); auto mode escalates instead
    if (emu_block) return false;
    if (emu_threaded == 2) throw EmuNeedThreads{};
    return true;
}

inline void __syncthreads() { if (!emu_lone()) emu_barrier(16); }
template <typename T> inline T emu_exchange(T v, unsigned src_lane_of_me) {
    static_assert(sizeof(T) <= 8, "shuffles move at most 8 bytes");
    if (emu_lone()) return v;
    const unsigned w = emu_tid >> 6, lane = emu_tid & 63;
    std::memcpy(&emu_block->slot[w][lane], &v, sizeof v);
    emu_barrier(w);
    T r;
    std::memcpy(&r, &emu_block->slot[w][src_lane_of_me & 63], sizeof r);
    emu_barrier(w);
    return r;
}
inline int __any(int p) {                                      // wave-wide OR (threaded mode; sequentially a lane only sees itself)
    if (emu_lone()) return p;
    const unsigned w = emu_tid >> 6, lane = emu_tid & 63;
    emu_block->slot[w][lane] = p ? 1u : 0u;
    emu_barrier(w);
    uint64_t any = 0;
    for (unsigned l = 0; l < emu_block->wave_lanes[w]; ++l) any |= emu_block->slot[w][l];
    emu_barrier(w);
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
    if (emu_lone()) std::abort();                             // a wave-wide operation: threaded (or auto) mode only
    EmuBlock* blk = emu_block;
    const unsigned w = emu_tid >> 6, lane = emu_tid & 63;
    const unsigned par = blk->mfma_parity[emu_tid] ^= 1;
    std::memcpy(blk->ma[par][w][lane], &a, 16);
    std::memcpy(blk->mb[par][w][lane], &b, 16);
    emu_barrier(w);
    const unsigned col = lane & 15;
    emu_f4 d = c;
    for (int r = 0; r < 4; ++r) {
        const unsigned row = 4 * (lane >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            _Float16 av, bv;
            std::memcpy(&av, blk->ma[par][w][(k >> 3) * 16 + row] + 2 * (k & 7), 2);
            std::memcpy(&bv, blk->mb[par][w][(k >> 3) * 16 + col] + 2 * (k & 7), 2);
            acc += (float)av * (float)bv;
        }
        d[r] = acc;
    }
    return d;
}
// v_mfma_f32_32x32x16_f16: D (32 x 32) = A (32 x 16) B (16 x 32) + C.  Lane l supplies A[l % 32][8 (l / 32) .. + 7] and
// B[8 (l / 32) .. + 7][l % 32] and owns D[8 (r / 4) + 4 (l / 32) + r % 4][l % 32], r = 0 .. 15.
typedef float emu_f16v __attribute__((ext_vector_type(16)));
inline emu_f16v __builtin_amdgcn_mfma_f32_32x32x16_f16(emu_h8 a, emu_h8 b, emu_f16v c, int, int, int) {
    if (emu_lone()) std::abort();
    EmuBlock* blk = emu_block;
    const unsigned w = emu_tid >> 6, lane = emu_tid & 63;
    const unsigned par = blk->mfma_parity[emu_tid] ^= 1;
    std::memcpy(blk->ma[par][w][lane], &a, 16);
    std::memcpy(blk->mb[par][w][lane], &b, 16);
    emu_barrier(w);
    const unsigned col = lane & 31;
    emu_f16v d = c;
    for (int r = 0; r < 16; ++r) {
        const unsigned row = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            _Float16 av, bv;
            std::memcpy(&av, blk->ma[par][w][(k >> 3) * 32 + row] + 2 * (k & 7), 2);
            std::memcpy(&bv, blk->mb[par][w][(k >> 3) * 32 + col] + 2 * (k & 7), 2);
            acc += (float)av * (float)bv;
        }
        d[r] = acc;
    }
    return d;
}
// v_permlane32_swap_b32 x, y (inline assembly in attention.hip, replaced textually by this call): the upper 32 lanes of x swap with the
// lower 32 lanes of y
inline void emu_permlane32_swap(float& x, float& y) {
    if (emu_lone()) std::abort();
    EmuBlock* blk = emu_block;
    const unsigned w = emu_tid >> 6, lane = emu_tid & 63;
    uint32_t xb, yb;
    std::memcpy(&xb, &x, 4); std::memcpy(&yb, &y, 4);
    blk->slot[w][lane] = ((uint64_t)yb << 32) | xb;
    emu_barrier(w);
    const uint64_t other = blk->slot[w][lane ^ 32];
    emu_barrier(w);
    if (lane >= 32) xb = (uint32_t)(other >> 32);              // upper half of x := lower half of y
    else yb = (uint32_t)other;                                 // lower half of y := upper half of x
    std::memcpy(&x, &xb, 4); std::memcpy(&y, &yb, 4);
}
inline uint64_t __builtin_amdgcn_ballot_w64(bool p) {
    if (emu_lone()) return p ? 1 : 0;
    EmuBlock* blk = emu_block;
    const unsigned w = emu_tid >> 6, lane = emu_tid & 63;
    blk->slot[w][lane] = p ? 1u : 0u;
    emu_barrier(w);
    uint64_t m = 0;
    for (unsigned l = 0; l < blk->wave_lanes[w]; ++l) m |= blk->slot[w][l] << l;
    emu_barrier(w);
    return m;
}
// global_load_lds (LDS-DMA): lane l copies `size` bytes from ITS global pointer to (wave-uniform LDS base) + offset + l * size; here the
// copy completes at issue (the kernels wait with s_waitcnt before they read, and never issue into a buffer that is still being read)
inline void __builtin_amdgcn_global_load_lds(const void* g, void* lds, unsigned size, unsigned offset, unsigned) {
    std::memcpy((char*)lds + offset + (emu_tid & 63) * size, g, size);
}
inline void __builtin_amdgcn_s_barrier() { if (!emu_lone()) emu_barrier(16); }
inline void __builtin_amdgcn_sched_barrier(int) {}
// a wave's LDS operations execute in program order on the hardware (the builtin only pins the compiler's schedule); with one fiber per
// lane the order between DIFFERENT lanes of a wave needs a real rendezvous
inline void __builtin_amdgcn_wave_barrier() { if (!emu_lone()) emu_barrier(emu_tid >> 6); }
inline void __builtin_amdgcn_s_setprio(int) {}
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }  // (the kernels pass wave-uniform values)
inline long long __builtin_amdgcn_s_memtime() { return 0; }
inline float __builtin_amdgcn_exp2f(float x) { return std::exp2(x); }
// v_permlane16_swap: with the wave as four rows of 16 lanes, the odd rows of the first operand swap with the even rows of the second:
// x' = [x.row0, y.row0, x.row2, y.row2], y' = [x.row1, y.row1, x.row3, y.row3]; returns (x', y')
inline emu_u2 __builtin_amdgcn_permlane16_swap(unsigned x, unsigned y, bool, bool) {
    if (emu_lone()) std::abort();
    EmuBlock* blk = emu_block;
    const unsigned w = emu_tid >> 6, lane = emu_tid & 63, row = lane >> 4;
    blk->slot[w][lane] = ((uint64_t)y << 32) | x;
    emu_barrier(w);
    const uint64_t below = blk->slot[w][(lane + 48) & 63], above = blk->slot[w][(lane + 16) & 63];      // lanes one row down / up
    const unsigned xn = (row & 1) ? (unsigned)(below >> 32) : x;             // odd row of x' = the y of the row below
    const unsigned yn = (row & 1) ? y : (unsigned)above;                     // even row of y' = the x of the row above
    emu_barrier(w);
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

// one block with one OS thread per GPU thread, on a persistent pool per block size (tests/hostemu/emu.cpp)
void emu_run_block_threaded(dim3 g, dim3 b, dim3 bi, const std::function<void()>& body);
inline void emu_launch(dim3 g, dim3 b, const std::function<void()>& body, int* site_threaded) {
    bool threaded = emu_threaded == 1 || (emu_threaded == 2 && *site_threaded);
    for (unsigned bz = 0; bz < g.z; ++bz) for (unsigned by = 0; by < g.y; ++by) for (unsigned bx = 0; bx < g.x; ++bx) {
        if (!threaded) {
            gridDim = g; blockDim = b; blockIdx = dim3(bx, by, bz);
            try {
                for (unsigned tz = 0; tz < b.z; ++tz) for (unsigned ty = 0; ty < b.y; ++ty) for (unsigned tx = 0; tx < b.x; ++tx) {
                    threadIdx = dim3(tx, ty, tz);
                    body();
                }
                continue;
            } catch (const EmuNeedThreads&) {                  // auto mode: this kernel synchronises; re-run the block with threads
                threaded = true;
                *site_threaded = 1;
            }
        }
        emu_run_block_threaded(g, b, dim3(bx, by, bz), body);
    }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    do { static int emu_site_threaded = 0; emu_launch((grid), (block), [&]() { kernel(__VA_ARGS__); }, &emu_site_threaded); } while (0)
