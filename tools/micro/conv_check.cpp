// conv_check — standalone (no Python, no torch) check + timing of the 3x3 implicit-GEMM K walks of libsdmi through its C ABI.
//
//   hipcc -O2 -std=c++17 tools/micro/conv_check.cpp -Iinclude -Lstable-diffusion-webui_amd/lib -lsdmi \
//         -Wl,-rpath,'$ORIGIN/../../stable-diffusion-webui_amd/lib' -o tools/micro/conv_check
//   tools/micro/conv_check [iters]
//
// For every conv shape of the C1 UNet / VAE that the row-shared walk (conv_korder = 2, gemm_mfma_pingpong_dx_kernel) admits, plus a few
// it must refuse (W = 8, stride 2, fused upsample): the one-thread-per-output generic kernel is the reference; the tap-major MFMA
// path (conv_korder 0) and the row-shared path (2) are compared with it (rel-L2, max |diff|) and with each other, and both are timed
// with sdmi_bench_conv_gemm.  Exit code 1 when a result is off.  A fresh box runs this in seconds — the Python suite needs ~2 minutes
// of imports first — so kernel iterations are checked here and confirmed in tests/test_gpu_ops.py afterwards.
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

static unsigned long long rng_state = 0x5D15C0DEull;
static float frand() {                                   // uniform in [-1, 1)
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((rng_state >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f;
}

struct Shape { const char* name; int B, H, W, c0, c1, N, stride, up, resid, rowbias; };

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    if (!sdmi_device_ok()) { fprintf(stderr, "no gfx950 device: %s\n", sdmi_last_error()); return 2; }
    const Shape shapes[] = {
        // name                         B   H    W   c0    c1    N  stride up resid rowbias
        {"L0 320->320",                16,  64,  64,  320,    0,  320, 1, 0, 0, 1},
        {"L0 320->320 +resid",         16,  64,  64,  320,    0,  320, 1, 0, 1, 0},
        {"L0 320+320->320",            16,  64,  64,  320,  320,  320, 1, 0, 0, 1},
        {"L0 640+320->320",            16,  64,  64,  640,  320,  320, 1, 0, 0, 1},
        {"L1 640->640",                16,  32,  32,  640,    0,  640, 1, 0, 0, 1},
        {"L1 320->640",                16,  32,  32,  320,    0,  640, 1, 0, 0, 1},
        {"L1 640+640->640",            16,  32,  32,  640,  640,  640, 1, 0, 0, 1},
        {"L1 1280+640->640",           16,  32,  32, 1280,  640,  640, 1, 0, 0, 1},
        {"L2 1280->1280 (split-K)",    16,  16,  16, 1280,    0, 1280, 1, 0, 0, 1},
        {"L2 1280+1280->1280",         16,  16,  16, 1280, 1280, 1280, 1, 0, 0, 1},
        {"L2 640->1280",               16,  16,  16,  640,    0, 1280, 1, 0, 0, 1},
        {"L3 1280->1280 (W=8: tap)",   16,   8,   8, 1280,    0, 1280, 1, 0, 0, 1},
        {"VAE 64^2 512->512",           8,  64,  64,  512,    0,  512, 1, 0, 1, 0},
        {"VAE 128^2 512->512",          4, 128, 128,  512,    0,  512, 1, 0, 0, 0},
        {"VAE 256^2 256->256",          2, 256, 256,  256,    0,  256, 1, 0, 0, 0},
        {"ragged M (B=3, 48x48)",       3,  48,  48,  320,    0,  320, 1, 0, 0, 1},
        {"L0 down stride 2 (tap)",     16,  64,  64,  320,    0,  320, 2, 0, 0, 0},
        {"L1 up x2 (tap)",             16,  16,  16, 1280,    0, 1280, 1, 1, 0, 0},
    };
    int bad = 0;
    printf("%-28s %7s %6s %6s | %9s %9s %9s | %8s %8s %7s | %8s %8s\n", "shape", "M", "N", "K", "tap:relL2", "dx:relL2", "dx-tap max",
           "tap us", "dx us", "dx/tap", "tap TF/s", "dx TF/s");
    for (const Shape& sh : shapes) {
        const int cin = sh.c0 + sh.c1, K = 9 * cin;
        const int Ho = sh.up ? sh.H * 2 : sh.H / sh.stride, Wo = sh.up ? sh.W * 2 : sh.W / sh.stride;
        const long M = (long)sh.B * Ho * Wo;
        const long na0 = (long)sh.B * sh.H * sh.W * sh.c0, na1 = (long)sh.B * sh.H * sh.W * sh.c1, nw = (long)sh.N * K, no = M * sh.N;
        std::vector<half_t> ha0(na0), ha1(na1 > 0 ? na1 : 1), hw(nw), hr(no);
        std::vector<float> hb(sh.N), hrb((long)sh.B * sh.N);
        const float wscale = 1.0f / std::sqrt((float)K);
        for (auto& v : ha0) v = (half_t)frand();
        for (auto& v : ha1) v = (half_t)frand();
        for (auto& v : hw) v = (half_t)(frand() * wscale * 1.7f);
        for (auto& v : hr) v = (half_t)frand();
        for (auto& v : hb) v = frand() * 0.1f;
        for (auto& v : hrb) v = frand() * 0.1f;
        half_t *a0, *a1 = nullptr, *w, *r, *o_ref, *o_tap, *o_dx;
        float *b, *rb;
        void* ws = nullptr;
        HIP_OK(hipMalloc(&a0, na0 * 2)); HIP_OK(hipMemcpy(a0, ha0.data(), na0 * 2, hipMemcpyHostToDevice));
        if (na1) { HIP_OK(hipMalloc(&a1, na1 * 2)); HIP_OK(hipMemcpy(a1, ha1.data(), na1 * 2, hipMemcpyHostToDevice)); }
        HIP_OK(hipMalloc(&w, nw * 2)); HIP_OK(hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&r, no * 2)); HIP_OK(hipMemcpy(r, hr.data(), no * 2, hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&b, sh.N * 4)); HIP_OK(hipMemcpy(b, hb.data(), sh.N * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&rb, (long)sh.B * sh.N * 4)); HIP_OK(hipMemcpy(rb, hrb.data(), (long)sh.B * sh.N * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&o_ref, no * 2)); HIP_OK(hipMalloc(&o_tap, no * 2)); HIP_OK(hipMalloc(&o_dx, no * 2));
        const long wsb = sdmi_conv_splitk_workspace_bytes((int)M, sh.N, K, 1);
        if (wsb > 0) HIP_OK(hipMalloc(&ws, wsb));

        sdmi_conv_desc d;
        memset(&d, 0, sizeof d);
        d.a0 = a0; d.a1 = a1; d.w = w; d.bias = b; d.rowbias = sh.rowbias ? rb : nullptr; d.resid = sh.resid ? r : nullptr;
        d.c0 = sh.c0; d.c1 = sh.c1; d.lda0 = sh.c0; d.lda1 = sh.c1;
        d.B = sh.B; d.Hi = sh.H; d.Wi = sh.W; d.Ho = Ho; d.Wo = Wo;
        d.taps = 9; d.stride = sh.stride; d.pad = 1; d.up = sh.up;
        d.N = sh.N; d.n_real = sh.N; d.ldo = sh.N; d.ldr = sh.N; d.alpha = 1.0f; d.batch = 1;
        d.splitk_workspace = ws; d.splitk_workspace_bytes = wsb;

        auto run = [&](half_t* out, int generic, int korder, float* us) {
            SDMI_OK(sdmi_debug_set("conv_korder", korder));
            d.out = out; d.force_generic = generic;
            HIP_OK(hipMemset(out, 0xFF, no * 2));            // NaN pattern: an unwritten output is seen
            SDMI_OK(sdmi_conv_gemm(&d, nullptr));
            HIP_OK(hipDeviceSynchronize());
            if (us) { float ms = 0.f; SDMI_OK(sdmi_bench_conv_gemm(&d, iters, &ms, nullptr)); *us = ms * 1000.f; }
        };
        float us_tap = 0.f, us_dx = 0.f;
        run(o_ref, 1, 0, nullptr);
        run(o_tap, 0, 0, &us_tap);
        run(o_dx, 0, 2, &us_dx);
        std::vector<half_t> ref(no), tap(no), dx(no);
        HIP_OK(hipMemcpy(ref.data(), o_ref, no * 2, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(tap.data(), o_tap, no * 2, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(dx.data(), o_dx, no * 2, hipMemcpyDeviceToHost));
        double n_ref = 0, e_tap = 0, e_dx = 0, dmax = 0;
        long nan = 0;
        for (long i = 0; i < no; ++i) {
            const double x = (double)ref[i], t = (double)tap[i], y = (double)dx[i];
            if (!(y == y) || !(t == t)) { ++nan; continue; }
            n_ref += x * x; e_tap += (t - x) * (t - x); e_dx += (y - x) * (y - x);
            dmax = std::fmax(dmax, std::fabs(y - t));
        }
        const double r_tap = std::sqrt(e_tap / (n_ref + 1e-30)), r_dx = std::sqrt(e_dx / (n_ref + 1e-30));
        const double flop = 2.0 * M * sh.N * K;
        const bool ok = nan == 0 && r_tap < 6e-4 && r_dx < 6e-4;          // fp16 store: tests/test_gpu_ops.py states 6e-4 for conv outputs
        if (!ok) ++bad;
        printf("%-28s %7ld %6d %6d | %9.2e %9.2e %9.2e | %8.1f %8.1f %7.3f | %8.1f %8.1f %s\n", sh.name, M, sh.N, K, r_tap, r_dx, dmax, us_tap, us_dx,
               us_dx / us_tap, flop / us_tap * 1e-6, flop / us_dx * 1e-6, ok ? "" : (nan ? "  <-- NaN / unwritten" : "  <-- MISMATCH"));
        fflush(stdout);
        hipFree(a0); if (a1) hipFree(a1); hipFree(w); hipFree(r); hipFree(b); hipFree(rb); hipFree(o_ref); hipFree(o_tap); hipFree(o_dx);
        if (ws) hipFree(ws);
    }
    sdmi_debug_set("conv_korder", 0);
    printf(bad ? "FAILED: %d shape(s)\n" : "all shapes ok\n", bad);
    return bad ? 1 : 0;
}
