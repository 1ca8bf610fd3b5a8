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

// read + write streams (what a normalisation kernel is): out[i] = in[i] over `bytes`, 16 bytes per lane, `per_wg` bytes per workgroup
// visit, grid-stride over the tensor (a one-shot grid when wgs * per_wg >= bytes).  UNROLL loads are issued before the first store.
typedef float f4v __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const f4v* __restrict__ in, f4v* __restrict__ out, size_t n16, int per_wg16) {
    const size_t stride = (size_t)gridDim.x * per_wg16;
    for (size_t base = (size_t)blockIdx.x * per_wg16; base < n16; base += stride) {
        for (int i = threadIdx.x; i < per_wg16; i += 256 * UNROLL) {
            f4v v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const size_t j = base + i + u * 256;
                if (i + u * 256 < per_wg16 && j < n16) v[u] = NT ? __builtin_nontemporal_load(in + j) : in[j];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const size_t j = base + i + u * 256;
                if (i + u * 256 < per_wg16 && j < n16) {
                    if (NT) __builtin_nontemporal_store(v[u], out + j); else out[j] = v[u];
                }
            }
        }
    }
}

template <int UNROLL, bool NT>
static void run_copy(const char* name, const f4v* in, f4v* out, size_t bytes, int wgs, int per_wg_bytes) {
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    const size_t n16 = bytes / 16;
    if (wgs == 0) wgs = (int)((bytes + per_wg_bytes - 1) / per_wg_bytes);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        HIP_OK(hipEventRecord(e0));
        hipLaunchKernelGGL((copy_kernel<UNROLL, NT>), dim3(wgs), dim3(256), 0, 0, in, out, n16, per_wg_bytes / 16);
        HIP_OK(hipEventRecord(e1));
        HIP_OK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    printf("  %-28s %6d workgroups x %6d B per visit, %d loads ahead  %7.1f us  %5.2f TB/s (read + write)\n", name, wgs, per_wg_bytes, UNROLL, best * 1e3,
           2.0 * bytes / best * 1e-9);
}

static void copies() {
    for (size_t mb : {21, 42, 84, 168}) {
        const size_t bytes = mb << 20;
        f4v *in, *out;
        HIP_OK(hipMalloc(&in, bytes)); HIP_OK(hipMalloc(&out, bytes));
        HIP_OK(hipMemset(in, 1, bytes)); HIP_OK(hipMemset(out, 0, bytes));
        printf("copy of %zu MB (the tensor was written by the previous launch: cache state of a consumer)\n", mb);
        run_copy<1, false>("one-shot", in, out, bytes, 0, 10240);
        run_copy<1, false>("one-shot", in, out, bytes, 0, 16384);
        run_copy<4, false>("one-shot", in, out, bytes, 0, 16384);
        run_copy<4, false>("one-shot", in, out, bytes, 0, 65536);
        run_copy<4, false>("grid-stride", in, out, bytes, 1024, 16384);
        run_copy<4, false>("grid-stride", in, out, bytes, 2048, 16384);
        run_copy<4, false>("grid-stride", in, out, bytes, 2048, 65536);
        run_copy<8, false>("grid-stride", in, out, bytes, 2048, 32768);
        run_copy<4, true>("grid-stride, nontemporal", in, out, bytes, 2048, 16384);
        HIP_OK(hipFree(in)); HIP_OK(hipFree(out));
    }
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'c') { copies(); return 0; }
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
