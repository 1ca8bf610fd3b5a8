// Microbenchmark (tuning aid, not part of the library): cost of s_barrier for an 8-wave workgroup, alone and in the
// ping-pong arrangement (two 4-wave groups one barrier apart, one group issuing 16 MFMAs per slot).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_barrier(long long* out, int iters) {
    long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_barrier();
    long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int NM, bool PING>
__global__ __launch_bounds__(512) void k_pingpong(long long* out, float* sink, int iters) {
    const int wave = threadIdx.x >> 6;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    f4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
    if (PING && wave >= 4) __builtin_amdgcn_s_barrier();
    long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    for (int i = 0; i < iters; ++i) {
        __builtin_amdgcn_s_barrier();                 // end of "L" (empty)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m % 16] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m % 16], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (PING) __builtin_amdgcn_s_barrier();       // end of "M"
        __builtin_amdgcn_sched_barrier(0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

int main() {
    long long* out; float* sink;
    hipMalloc(&out, 64 * 8); hipMalloc(&sink, 4);
    long long h[8];
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_barrier, dim3(256), dim3(512), 0, 0, out, iters);
        hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
        printf("bare s_barrier, 8 waves            : %.1f cycles per barrier\n", (double)h[0] / iters);
        hipLaunchKernelGGL((k_pingpong<16, false>), dim3(256), dim3(512), 0, 0, out, sink, iters);
        hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("lockstep: barrier + 16 MFMA/wave   : %.1f cycles per iteration (MFMA floor 2 waves/SIMD x 16 x 16 = 512)\n", (double)h[0] / iters);
        hipLaunchKernelGGL((k_pingpong<16, true>), dim3(256), dim3(512), 0, 0, out, sink, iters);
        hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("ping-pong: 2 barriers + 16 MFMA    : %.1f / %.1f cycles per phase (floor 512)\n", (double)h[0] / iters, (double)h[4] / iters);
        hipLaunchKernelGGL((k_pingpong<20, true>), dim3(256), dim3(512), 0, 0, out, sink, iters);
        hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("ping-pong: 2 barriers + 20 MFMA    : %.1f / %.1f cycles per phase (floor 640)\n", (double)h[0] / iters, (double)h[4] / iters);
        hipLaunchKernelGGL((k_pingpong<32, true>), dim3(256), dim3(512), 0, 0, out, sink, iters);
        hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("ping-pong: 2 barriers + 32 MFMA    : %.1f / %.1f cycles per phase (floor 1024)\n", (double)h[0] / iters, (double)h[4] / iters);
    }
    return 0;
}
