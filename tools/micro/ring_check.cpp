// ring_check — standalone check + timing of the ring-buffered 4-wave GEMM tiles (gemm_mfma_kernel NS > 2) on the short-K 1x1 / linear
// shapes of the C1 UNet, through the C ABI (no Python).  Build: see conv_check.cpp.
//
// Per shape: the generic kernel is the reference; the engine's default choice (tuned table / score model) and every forced tile
// configuration in `cfgs` are compared with it bitwise-or-rounding (rel-L2) and timed with sdmi_bench_conv_gemm.  A ring tile must
// produce the bits of its two-stage twin (same accumulation order): checked for 128x160 (9 vs 10 / 13), 128x128 (0 vs 11), 128x64 (7 vs 12).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "sdmi.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define SDMI_OK(x) do { if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, sdmi_last_error()); exit(2); } } while (0)
typedef _Float16 half_t;

static unsigned long long rng_state = 0xC0FFEEull;
static float frand() {
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((rng_state >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f;
}

struct Shape { const char* name; int M, N, K, resid, flags; };

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 50;
    if (!sdmi_device_ok()) { fprintf(stderr, "no gfx950 device: %s\n", sdmi_last_error()); return 2; }
    const Shape shapes[] = {
        {"L2 proj/out  M4096 N1280 K1280",        4096, 1280, 1280, 1, 0},
        {"L2 q         M4096 N1280 K1280",        4096, 1280, 1280, 0, 0},
        {"L1 proj/out  M16384 N640 K640",        16384,  640,  640, 1, 0},
        {"L0 proj/out  M65536 N320 K320",        65536,  320,  320, 1, 0},
        {"L0 q         M65536 N320 K320",        65536,  320,  320, 0, 0},
        {"L0 qk        M65536 N640 K320",        65536,  640,  320, 0, 0},
        {"L2 ff2       M4096 N1280 K5120",        4096, 1280, 5120, 1, 0},
        {"L1 ff2       M16384 N640 K2560",       16384,  640, 2560, 1, 0},
        {"L3 mid       M1024 N1280 K1280",        1024, 1280, 1280, 1, 0},
        {"L2 vT        M4096 N1280 K1280 (tr)",   4096, 1280, 1280, 0, SDMI_EP_TRANSPOSE},
        {"L2 geglu     M4096 N10240 K1280",       4096, 10240, 1280, 0, SDMI_EP_GEGLU},
        {"ragged       M1000 N320 K192",          1000,  320,  192, 1, 0},
    };
    const int cfgs[] = {-1, 7, 12, 0, 11, 9, 13, 10, 8};      // -1 = the engine's own choice
    const char* cname[] = {"default", "128x64", "128x64r3", "128x128", "128x128r4", "128x160", "128x160r3", "128x160r4", "128x320pp"};
    const int ncfg = sizeof(cfgs) / sizeof(cfgs[0]);
    int bad = 0;
    for (const Shape& sh : shapes) {
        const int rows_per_img = sh.M >= 256 ? sh.M / 16 : sh.M;          // 16 "images" (EP_TRANSPOSE stores per image)
        const int B = sh.M / rows_per_img;
        const long M = (long)B * rows_per_img;
        const int nout_cols = (sh.flags & SDMI_EP_GEGLU) ? sh.N / 2 : sh.N;
        const long na = M * sh.K, nw = (long)sh.N * sh.K, no = M * nout_cols;
        std::vector<half_t> ha(na), hw(nw), hr(no);
        std::vector<float> hb(sh.N);
        const float ws = 1.7f / std::sqrt((float)sh.K);
        for (auto& v : ha) v = (half_t)frand();
        for (auto& v : hw) v = (half_t)(frand() * ws);
        for (auto& v : hr) v = (half_t)frand();
        for (auto& v : hb) v = frand() * 0.1f;
        half_t *a, *w, *r, *o_ref, *o;
        float* b;
        HIP_OK(hipMalloc(&a, na * 2)); HIP_OK(hipMemcpy(a, ha.data(), na * 2, hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&w, nw * 2)); HIP_OK(hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&r, no * 2)); HIP_OK(hipMemcpy(r, hr.data(), no * 2, hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&b, sh.N * 4)); HIP_OK(hipMemcpy(b, hb.data(), sh.N * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&o_ref, no * 2)); HIP_OK(hipMalloc(&o, no * 2));
        sdmi_conv_desc d;
        memset(&d, 0, sizeof d);
        d.a0 = a; d.w = w; d.bias = b; d.resid = sh.resid ? r : nullptr;
        d.c0 = sh.K; d.lda0 = sh.K;
        d.B = B; d.Hi = rows_per_img; d.Wi = 1; d.Ho = rows_per_img; d.Wo = 1;
        d.taps = 1; d.stride = 1; d.pad = 0; d.N = sh.N; d.n_real = sh.N;
        d.ldo = (sh.flags & SDMI_EP_TRANSPOSE) ? rows_per_img : nout_cols; d.ldr = sh.N; d.flags = sh.flags; d.alpha = 1.0f; d.batch = 1;
        std::vector<half_t> ref(no), got(no);
        std::vector<std::vector<half_t>> outs(ncfg);
        d.out = o_ref; d.force_generic = 1;
        SDMI_OK(sdmi_conv_gemm(&d, nullptr));
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(ref.data(), o_ref, no * 2, hipMemcpyDeviceToHost));
        printf("%s\n", sh.name);
        const double flop = 2.0 * M * sh.N * sh.K;
        for (int c = 0; c < ncfg; ++c) {
            SDMI_OK(sdmi_debug_set("gemm_cfg", cfgs[c]));
            d.out = o; d.force_generic = 0;
            HIP_OK(hipMemset(o, 0xFF, no * 2));
            if (sdmi_conv_gemm(&d, nullptr) != 0) { printf("    %-10s refused: %s\n", cname[c], sdmi_last_error()); continue; }
            HIP_OK(hipDeviceSynchronize());
            float ms = 0.f;
            SDMI_OK(sdmi_bench_conv_gemm(&d, iters, &ms, nullptr));
            HIP_OK(hipMemcpy(got.data(), o, no * 2, hipMemcpyDeviceToHost));
            outs[c] = got;
            double n_ref = 0, e = 0;
            long nan = 0;
            for (long i = 0; i < no; ++i) {
                const double x = (double)ref[i], y = (double)got[i];
                if (!(y == y)) { ++nan; continue; }
                n_ref += x * x; e += (y - x) * (y - x);
            }
            const double rel = std::sqrt(e / (n_ref + 1e-30));
            const bool ok = nan == 0 && rel < 6e-4;
            if (!ok) ++bad;
            printf("    %-10s %8.1f us %8.1f TFLOP/s  rel-L2 vs generic %.2e%s\n", cname[c], ms * 1000.f, flop / ms * 1e-9, rel,
                   ok ? "" : (nan ? "  <-- NaN / unwritten" : "  <-- MISMATCH"));
        }
        // the register-staged form of the two-stage tiles (global_load_dwordx4 -> VGPR -> ds_write_b128 instead of LDS-direct loads): is the
        // LDS-DMA path what bounds these shapes?  (force_generic = 2; timing only, the bits are the LDS-direct kernel's: tests/test_gpu_ops.py)
        if (argc > 2) {
            for (int c = 0; c < ncfg; ++c) {
                if (cfgs[c] == 12 || cfgs[c] == 11 || cfgs[c] == 13 || cfgs[c] == 10 || cfgs[c] == 8) continue;      // ring / ping-pong: LDS-direct only
                SDMI_OK(sdmi_debug_set("gemm_cfg", cfgs[c]));
                d.out = o; d.force_generic = 2;
                if (sdmi_conv_gemm(&d, nullptr) != 0) continue;
                HIP_OK(hipDeviceSynchronize());
                float ms = 0.f;
                SDMI_OK(sdmi_bench_conv_gemm(&d, iters, &ms, nullptr));
                printf("    %-10s %8.1f us %8.1f TFLOP/s  (register-staged)\n", cname[c], ms * 1000.f, flop / ms * 1e-9);
            }
        }
        // ring tiles against their two-stage twins: same bits
        const int twins[][2] = {{1, 2}, {3, 4}, {5, 6}, {5, 7}};
        for (auto& t : twins) {
            if (outs[t[0]].empty() || outs[t[1]].empty()) continue;
            if (memcmp(outs[t[0]].data(), outs[t[1]].data(), no * 2) != 0) { printf("    %s != %s bitwise  <-- MISMATCH\n", cname[t[0]], cname[t[1]]); ++bad; }
        }
        fflush(stdout);
        hipFree(a); hipFree(w); hipFree(r); hipFree(b); hipFree(o_ref); hipFree(o);
    }
    sdmi_debug_set("gemm_cfg", -1);
    printf(bad ? "FAILED: %d\n" : "all ok\n", bad);
    return bad ? 1 : 0;
}
