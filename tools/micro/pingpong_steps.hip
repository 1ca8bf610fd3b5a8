// Microbenchmark (tuning aid): ping-pong phase cost as the real kernel's ingredients are added one at a time.
//   DS  : M section ends with NDS ds_read_b128 whose results feed the next phase's MFMAs (lgkmcnt(0) in the L section)
//   GL  : L section issues NGL LDS-direct global loads (16 B/lane) and a counted vmcnt wait
//   VAL : L section carries NVAL dependent VALU instructions (address arithmetic stand-in)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int NM, int NDS, int NGL, int NVAL, int RD = 0, int GLPOS = 0>
__global__ __launch_bounds__(512) void k(long long* out, float* sink, const char* src, long span, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    h8 fr[8];
    for (int r = 0; r < 8; ++r)
        for (int e = 0; e < 8; ++e) fr[r][e] = (_Float16)(tid * 0.001f + e + r);
    f4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
    const int lr = lane & 15, lk = lane >> 4;
    const int rd = (wave * 16 + lr) * 128 + ((lk ^ (lr & 7)) << 4);       // conflict-free b128 pattern of the GEMM
    const char* g = src + ((long)blockIdx.x * 512 + tid) * 16;
    long off = 0;
    int v = tid;
    if (wave >= 4) __builtin_amdgcn_s_barrier();
    long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    for (int i = 0; i < iters; i += 2) {
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        // ---- L
        if (RD == 1) {
#pragma unroll
            for (int n = 0; n < NDS; ++n) fr[n & 3] = *reinterpret_cast<const h8*>(smem + rd + (n & 3) * 16384 + par * 2048);
        }
#pragma unroll
        for (int n = 0; n < (GLPOS == 0 ? NGL : 0); ++n) {
            __builtin_amdgcn_global_load_lds((gptr_t)(g + off), (lptr_t)(smem + 65536 + ((i & 7) * 8 + wave) * 1024 + n * 32768 % 65536), 16, 0, 0);
            off += 512 * 256 * 16;
            if (off >= span) off = 0;
        }
#pragma unroll
        for (int n = 0; n < NVAL; ++n) v = v * 3 + n;
        if (NGL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NGL * 4) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- M
        if (RD == 2) {                       // prefetch the NEXT phase's fragments into the other register set
#pragma unroll
            for (int n = 0; n < NDS; ++n) fr[(par ^ 1) * 4 + (n & 3)] = *reinterpret_cast<const h8*>(smem + rd + (n & 3) * 16384 + par * 2048);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(1);
        constexpr int FB = 0;
#pragma unroll
        for (int m = 0; m < NM; ++m)
            acc[m % 16] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[(RD == 2 ? par * 4 : FB) + (m & 1)], fr[(RD == 2 ? par * 4 : FB) + 2 + ((m >> 1) & 1)], acc[m % 16], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (RD == 0) {
#pragma unroll
            for (int n = 0; n < NDS; ++n) fr[n & 3] = *reinterpret_cast<const h8*>(smem + rd + (n & 3) * 16384 + par * 2048);
        }
#pragma unroll
        for (int n = 0; n < (GLPOS == 1 ? NGL : 0); ++n) {
            __builtin_amdgcn_global_load_lds((gptr_t)(g + off), (lptr_t)(smem + 65536 + ((i & 7) * 8 + wave) * 1024 + n * 32768 % 65536), 16, 0, 0);
            off += 512 * 256 * 16;
            if (off >= span) off = 0;
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    float s = (float)v;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

template <int NM, int NDS, int NGL, int NVAL, int RD = 0, int GLPOS = 0>
void run(const char* name, long long* out, float* sink, const char* src, long span) {
    const int iters = 2000;
    auto kern = k<NM, NDS, NGL, NVAL, RD, GLPOS>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    long long h[8];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, 0, out, sink, src, span, iters);
        (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    }
    printf("%-52s: %7.1f / %7.1f cycles per phase (MFMA floor %d)\n", name, (double)h[0] / iters, (double)h[4] / iters, NM * 32);
}

int main() {
    long long* out; float* sink; char* src;
    const long span = 1L << 30;                        // 1 GiB source: misses L2, streams from HBM / Infinity Cache
    (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 4); (void)hipMalloc(&src, span + (1 << 22));
    (void)hipMemset(src, 0, span + (1 << 22));
    run<16, 0, 0, 0>("16 MFMA", out, sink, src, span);
    run<16, 4, 0, 0, 0>("16 MFMA + 4 ds_read at M end (same regs)", out, sink, src, span);
    run<16, 4, 0, 0, 1>("16 MFMA + 4 ds_read at L start (same regs)", out, sink, src, span);
    run<16, 4, 0, 0, 2>("16 MFMA + 4 ds_read at M start (other regs)", out, sink, src, span);
    run<16, 12, 0, 0, 2>("16 MFMA + 12 ds_read at M start (other regs)", out, sink, src, span);
    run<16, 0, 2, 0, 0, 0>("16 MFMA + 2 glds in L (L2)", out, sink, src, 8L << 20);
    run<16, 0, 2, 0, 0, 1>("16 MFMA + 2 glds at M end (L2)", out, sink, src, 8L << 20);
    run<16, 4, 2, 0, 2, 0>("16 MFMA + 4 rd M-start + 2 glds in L (L2)", out, sink, src, 8L << 20);
    run<16, 4, 2, 0, 2, 1>("16 MFMA + 4 rd M-start + 2 glds at M end (L2)", out, sink, src, 8L << 20);
    run<16, 4, 2, 0, 1, 0>("16 MFMA + 4 rd L-start + 2 glds in L (L2)", out, sink, src, 8L << 20);
    run<16, 4, 3, 24, 2, 0>("16 MFMA + 4 rd M-start + 3 glds L + 24 VALU (L2)", out, sink, src, 8L << 20);
    run<16, 4, 3, 24, 2, 0>("16 MFMA + 4 rd M-start + 3 glds L + 24 VALU (HBM)", out, sink, src, span);
    run<20, 4, 3, 24, 2, 0>("20 MFMA + 4 rd M-start + 3 glds L + 24 VALU (L2)", out, sink, src, 8L << 20);
    run<32, 8, 5, 24, 2, 0>("32 MFMA + 8 rd M-start + 5 glds L + 24 VALU (L2)", out, sink, src, 8L << 20);
    return 0;
}
