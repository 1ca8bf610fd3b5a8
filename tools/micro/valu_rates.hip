// Microbenchmark (tuning aid): issue cost of the VALU instructions of the attention softmax on gfx950, per wave64 instruction,
// with 1, 2 and 4 waves per SIMD; and of the same instructions issued beside another wave's MFMAs.
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2v __attribute__((ext_vector_type(2)));

template <int OP, int OPB = OP>
__global__ __launch_bounds__(1024) void k(long long* out, float* sink, int iters, int mfma_waves) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float a[8];
    f2v b[8];
    for (int i = 0; i < 8; ++i) { a[i] = tid * 0.001f + i; b[i] = f2v{a[i], a[i] + 1.f}; }
    f16v acc = {0};
    h8 fa, fb;
    for (int e = 0; e < 8; ++e) { fa[e] = (_Float16)(tid + e); fb[e] = (_Float16)(tid - e); }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (wave < mfma_waves) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int op = ((wave >> 2) & 1) ? OPB : OP;      // waves 0-3 sit on SIMDs 0-3, waves 4-7 again: every SIMD holds both kinds
                    if (op != OP) {
                        // odd waves of a mixed launch: the second instruction kind
                        if (OPB == 0) a[i] = __builtin_amdgcn_exp2f(a[i]);
                        else if (OPB == 3) a[i] = __builtin_fmaxf(__builtin_fmaxf(a[i], a[(i + 1) & 7]), a[(i + 2) & 7]);
                        else if (OPB == 2) b[i] = __builtin_elementwise_fma(b[i], f2v{1.0001f, 1.0001f}, f2v{0.5f, 0.5f});
                        continue;
                    }
                    if (OP == 0) a[i] = __builtin_amdgcn_exp2f(a[i]);
                    else if (OP == 1) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
                    else if (OP == 2) b[i] = __builtin_elementwise_fma(b[i], f2v{1.0001f, 1.0001f}, f2v{0.5f, 0.5f});
                    else if (OP == 3) a[i] = __builtin_fmaxf(__builtin_fmaxf(a[i], a[(i + 1) & 7]), a[(i + 2) & 7]);
                    else if (OP == 4) { auto h = __builtin_amdgcn_cvt_pkrtz(a[i], a[(i + 1) & 7]); a[i] += (float)h[0]; }
                    else if (OP == 5) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                    else if (OP == 6) asm volatile("v_exp_f16 %0, %1" : "=v"(a[i]) : "v"(a[i]));
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + b[i].x + b[i].y;
    for (int i = 0; i < 16; ++i) s += acc[i];
    sink[blockIdx.x * 1024 + tid] = s;
    if ((tid & 63) == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int OP, int OPB>
void run_mixed(const char* name, long long* d_out, float* d_sink) {
    const int iters = 2000;
    for (int waves_per_simd : {2, 4}) {
        const int threads = 256 * waves_per_simd;
        hipLaunchKernelGGL((k<OP, OPB>), dim3(256), dim3(threads), 0, 0, d_out, d_sink, iters, 0);
        hipDeviceSynchronize();
        long long h[16];
        hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
        double ev = 0, od = 0;
        const int nw = threads / 64;
        for (int w = 0; w < nw; ++w) (((w >> 2) & 1) ? od : ev) += h[w];
        printf("%-24s %d waves/SIMD: first kind %6.2f, second kind %6.2f cycles per instruction per wave\n", name, waves_per_simd,
               ev / (nw / 2) / (iters * 32.0), od / (nw / 2) / (iters * 32.0));
    }
}

template <int OP>
void run(const char* name, long long* d_out, float* d_sink) {
    const int iters = 2000;
    for (int waves_per_simd : {1, 2, 4}) {
        for (int mfma : {0, 1}) {
            if (mfma && waves_per_simd == 1) continue;
            const int threads = 256 * waves_per_simd;
            const int mfma_waves = mfma ? 4 : 0;     // waves 0..3 (one per SIMD) issue MFMAs back to back instead
            hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, d_out, d_sink, iters, mfma_waves);
            hipDeviceSynchronize();
            long long h[16];
            hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
            const int nw = threads / 64;
            double valu = 0, mf = 0;
            int nv = 0, nm = 0;
            for (int w = 0; w < nw; ++w) {
                if (w < mfma_waves) { mf += h[w]; ++nm; } else { valu += h[w]; ++nv; }
            }
            const double per = valu / nv / (iters * 32.0);
            printf("%-14s %d waves/SIMD%s : %6.2f cycles per instruction per wave (%5.2f SIMD cycles per instruction)", name, waves_per_simd,
                   mfma ? " (1 of them MFMA)" : "", per, per / (waves_per_simd - (mfma ? 1 : 0)));
            if (nm) printf("   MFMA wave: %6.2f cycles per MFMA", mf / nm / (iters * 8.0));
            printf("\n");
        }
    }
}

int main() {
    long long* d_out;
    float* d_sink;
    hipMalloc(&d_out, 256 * 16 * sizeof(long long));
    hipMalloc(&d_sink, 256 * 1024 * sizeof(float));
    run<0>("v_exp_f32", d_out, d_sink);
    run<6>("v_exp_f16", d_out, d_sink);
    run<1>("v_fma_f32", d_out, d_sink);
    run<2>("v_pk_fma_f32", d_out, d_sink);
    run<3>("v_max3_f32", d_out, d_sink);
    run<5>("v_cvt_pk_f16", d_out, d_sink);
    run_mixed<0, 2>("exp (w0-3) | pk_fma (w4-7)", d_out, d_sink);
    run_mixed<0, 3>("exp (w0-3) | max3 (w4-7)", d_out, d_sink);
    run_mixed<2, 3>("pk_fma (w0-3) | max3 (w4-7)", d_out, d_sink);
    return 0;
}
