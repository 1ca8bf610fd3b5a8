// fill_rate — how fast can one CU pull cache-resident bytes?  (round 5: is the GEMM family bound by the L1 miss path?)
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/fill_rate tools/micro/fill_rate.hip && tools/micro/fill_rate
//
// Every workgroup streams a window of a shared buffer (2 MB: resident in every XCD's 4 MB L2; 64 MB: Infinity Cache; 1 GB: HBM) with
// 16-byte-per-lane loads, `depth` wave-instructions (1 KB each) in flight per wave, as LDS-direct loads (global_load_lds_dwordx4, what
// the GEMM kernels use) or as register loads (global_load_dwordx4), for 1 / 2 / 4 waves per SIMD.  Prints TB/s over the chip and bytes
// per nanosecond per CU (x 1 / clock GHz = B/clk/CU; the MFMA kernels run at 2.2-2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef __attribute__((address_space(3))) void* lptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// mode 0: LDS-direct; 1: register loads.  DEPTH loads in flight per wave; `rounds` x DEPTH KiB per wave.
template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void fill_kernel(const char* buf, size_t mask, int rounds, int stride_wg, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwave = blockDim.x >> 6;
    size_t off = ((size_t)blockIdx.x * stride_wg + (size_t)wave * DEPTH * 1024) & mask;
    const size_t step = (size_t)nwave * DEPTH * 1024;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < rounds; ++r) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < DEPTH; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(buf + ((off + i * 1024) & mask) + lane * 16), (lptr_t)(smem + (wave * DEPTH + i) * 1024), 16, 0, 0);
            wait_vm<0>();
        } else {
            float4 v[DEPTH];
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) v[i] = *reinterpret_cast<const float4*>(buf + ((off + i * 1024) & mask) + lane * 16);
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
        }
        off = (off + step) & mask;
    }
    if (MODE == 0) acc.x = reinterpret_cast<float*>(smem)[threadIdx.x];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* buf, size_t bytes, int waves, int wgs, float* sink) {
    const int rounds = 256;
    const int smem = MODE == 0 ? waves * DEPTH * 1024 : 1024;
    auto kern = fill_kernel<MODE, DEPTH>;
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    // windows of neighbouring workgroups overlap (stride 64 KB over a small buffer): the L2 serves them; one warm-up launch
    for (int rep = 0; rep < 2; ++rep) {
        HIP_OK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(waves * 64), smem, 0, buf, bytes - 1, rounds, 65536, sink);
        HIP_OK(hipEventRecord(e1));
        HIP_OK(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    const double total = (double)wgs * waves * DEPTH * 1024.0 * rounds;
    printf("  %-34s %2d waves/CU x %2d KiB in flight per wave  %7.1f us  %6.2f TB/s  %6.1f B/ns/CU\n", name, waves, DEPTH, ms * 1e3, total / ms * 1e-9,
           total / wgs / (ms * 1e6));
}

int main() {
    float* sink;
    HIP_OK(hipMalloc(&sink, 64));
    const size_t sizes[] = {(size_t)2 << 20, (size_t)64 << 20, (size_t)1 << 30};
    const char* sname[] = {"2 MB buffer (every XCD's L2)", "64 MB buffer (Infinity Cache)", "1 GB buffer (HBM)"};
    for (int s = 0; s < 3; ++s) {
        char* buf;
        HIP_OK(hipMalloc(&buf, sizes[s]));
        HIP_OK(hipMemset(buf, 1, sizes[s]));
        printf("%s, 256 workgroups\n", sname[s]);
        for (int waves : {4, 8, 16}) {
            run<0, 4>("LDS-direct (global_load_lds x4)", buf, sizes[s], waves, 256, sink);
            run<0, 8>("LDS-direct (global_load_lds x4)", buf, sizes[s], waves, 256, sink);
            if (waves <= 8) run<0, 16>("LDS-direct (global_load_lds x4)", buf, sizes[s], waves, 256, sink);
            run<1, 4>("registers (global_load_dwordx4)", buf, sizes[s], waves, 256, sink);
            run<1, 8>("registers (global_load_dwordx4)", buf, sizes[s], waves, 256, sink);
            run<1, 16>("registers (global_load_dwordx4)", buf, sizes[s], waves, 256, sink);
        }
        if (s == 0) {
            printf("%s, 64 workgroups (a quarter of the CUs)\n", sname[s]);
            for (int waves : {4, 8}) {
                run<0, 8>("LDS-direct (global_load_lds x4)", buf, sizes[s], waves, 64, sink);
                run<1, 8>("registers (global_load_dwordx4)", buf, sizes[s], waves, 64, sink);
            }
        }
        HIP_OK(hipFree(buf));
    }
    return 0;
}
