// hipblaslt_yardstick — what does the vendor's GEMM library reach on the engine's 1x1 / linear shapes (and on the 3x3 convs' GEMM
// equivalents)?  A YARDSTICK, not a code path: the engine links no BLAS library, this tool is the only place hipBLASLt is touched, and
// its numbers answer one question for DESIGN.md section 9: is a shape's distance from the MFMA roof a property of the shape on this
// chip (the library is no faster) or of the engine's kernel (the library is)?
//
// Per shape, same process, same buffers, isolated loop (weights and activations cache-resident — the numbers of
// profiles/r04_ring_check.txt are of this kind; in-job times are 10-20 % higher):
//   engine    sdmi_bench_conv_gemm with the engine's own tile choice (fp16 in / out, fp32 accumulate, bias [+ residual] in the epilogue)
//   hipBLASLt D = A W^T (+ bias through HIPBLASLT_EPILOGUE_BIAS, + residual as beta * C), fp16 in / out, fp32 compute: the fastest of the
//             first `NALGO` heuristic results, each timed over `iters` launches
// and the two outputs are compared (rel-L2): the library call computes the same function.
//
// Build (links the engine's C ABI and hipBLASLt; needs no Python):
//   hipcc -O2 -std=c++17 tools/micro/hipblaslt_yardstick.cpp -Iinclude -Lstable-diffusion-webui_amd/lib -lsdmi -lhipblaslt \
//         -Wl,-rpath,'$ORIGIN/../../stable-diffusion-webui_amd/lib' -o tools/micro/hipblaslt_yardstick
// Run on the GPU box:  tools/micro/hipblaslt_yardstick [iters] > gpurun_out/hipblaslt_yardstick.txt
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <cmath>
#include <cstdio>
#include <string>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "sdmi.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define SDMI_OK(x) do { if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, sdmi_last_error()); exit(2); } } while (0)
#define LT_OK(x) do { hipblasStatus_t s_ = (x); if (s_ != HIPBLAS_STATUS_SUCCESS) { fprintf(stderr, "%s: hipblas status %d\n", #x, (int)s_); exit(2); } } while (0)
typedef _Float16 half_t;

static unsigned long long rng_state = 0xC0FFEEull;
static float frand() {
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((rng_state >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f;
}

// n values from a 4 M-entry pool of uniform [-1, 1) values, read from a varying offset (a fresh draw per element would cost more host time
// than the whole measurement: the largest operand has 189 M elements)
static void fill(std::vector<half_t>& v, float scale) {
    static std::vector<float> pool;
    if (pool.empty()) { pool.resize(1 << 22); for (auto& x : pool) x = frand(); }
    size_t off = (size_t)((frand() * 0.5f + 0.5f) * 4000000.0f) % pool.size();
    for (size_t i = 0; i < v.size(); ++i) { v[i] = (half_t)(pool[off] * scale); if (++off == pool.size()) off = 0; }
}

struct Shape { const char* name; int M, N, K, resid; };
static const int NALGO = 8;

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 30;
    if (!sdmi_device_ok()) { fprintf(stderr, "no gfx950 device: %s\n", sdmi_last_error()); return 2; }
    const Shape shapes[] = {
        // the 1x1 / linear class of the C1 forward (launches per forward in brackets)
        {"L0 proj / out-proj [23]   M65536 N320  K320  +res", 65536,  320,  320, 1},
        {"L0 q               [ 5]   M65536 N320  K320      ", 65536,  320,  320, 0},
        {"L0 q|k             [ 4]   M65536 N640  K320      ", 65536,  640,  320, 0},
        {"L0 ff2             [ 5]   M65536 N320  K1280 +res", 65536,  320, 1280, 1},
        {"L1 proj / out-proj [25]   M16384 N640  K640  +res", 16384,  640,  640, 1},
        {"L1 q|k             [ 5]   M16384 N1280 K640      ", 16384, 1280,  640, 0},
        {"L1 ff2             [ 5]   M16384 N640  K2560 +res", 16384,  640, 2560, 1},
        {"L2 proj / out-proj [25]   M4096  N1280 K1280 +res",  4096, 1280, 1280, 1},
        {"L2 q|k             [ 5]   M4096  N2560 K1280     ",  4096, 2560, 1280, 0},
        {"L2 ff2             [ 5]   M4096  N1280 K5120 +res",  4096, 1280, 5120, 1},
        {"mid proj / out-proj[ 5]   M1024  N1280 K1280 +res",  1024, 1280, 1280, 1},
        // ff1 without its GEGLU epilogue (the library has no gated form): the GEMM part of the launch only
        {"L0 ff1 (GEMM part) [ 5]   M65536 N2560 K320      ", 65536, 2560,  320, 0},
        {"L1 ff1 (GEMM part) [ 5]   M16384 N5120 K640      ", 16384, 5120,  640, 0},
        {"L2 ff1 (GEMM part) [ 5]   M4096  N10240 K1280    ",  4096, 10240, 1280, 0},
        // the plain-GEMM equivalents of the 3x3 convs (im2col already done: an UPPER bound on what a library conv could do)
        {"L0 conv 320->320  as GEMM M65536 N320  K2880     ", 65536,  320, 2880, 0},
        {"L1 conv 640->640  as GEMM M16384 N640  K5760     ", 16384,  640, 5760, 0},
        {"L2 conv 1280->1280 as GEMM M4096 N1280 K11520    ",  4096, 1280, 11520, 0},
        // the 8x8-latent level (16 rows x 64 pixels): weight-streaming shapes, split-K 8 + a reduce pass in the engine
        {"L3 conv 1280->1280 as GEMM M1024 N1280 K11520    ",  1024, 1280, 11520, 0},
        {"L3 conv 2560->1280 as GEMM M1024 N1280 K23040    ",  1024, 1280, 23040, 0},
    };
    // second table (argv[2] == "ksweep"): the level-1 / level-2 projection shapes at K = 320 ... 5120 — time against K separates a launch's
    // fixed part (ramp, first loads, epilogue with its residual, tail) from its K loop, for the engine and for the library alike
    static const Shape ksweep[] = {
        {"M4096  N1280 K320   +res", 4096, 1280,  320, 1}, {"M4096  N1280 K640   +res", 4096, 1280,  640, 1},
        {"M4096  N1280 K1280  +res", 4096, 1280, 1280, 1}, {"M4096  N1280 K2560  +res", 4096, 1280, 2560, 1},
        {"M4096  N1280 K5120  +res", 4096, 1280, 5120, 1},
        {"M16384 N640  K320   +res", 16384, 640,  320, 1}, {"M16384 N640  K640   +res", 16384, 640,  640, 1},
        {"M16384 N640  K1280  +res", 16384, 640, 1280, 1}, {"M16384 N640  K2560  +res", 16384, 640, 2560, 1},
        {"M4096  N1280 K1280      ", 4096, 1280, 1280, 0}, {"M16384 N640  K640       ", 16384, 640,  640, 0},
    };
    const bool sweep = argc > 2 && std::string(argv[2]) == "ksweep";
    if (const char* ov = getenv("SDMI_GEMM_OVERRIDE")) SDMI_OK(sdmi_debug_set_str("gemm_override", ov));   // force tiles per shape (engine column)
    hipblasLtHandle_t lt;
    LT_OK(hipblasLtCreate(&lt));
    const size_t ws_bytes = 256u << 20;
    void* ws = nullptr;
    HIP_OK(hipMalloc(&ws, ws_bytes));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    printf("%-52s %10s %10s %8s %10s %10s %6s %9s\n", "shape", "engine us", "TFLOP/s", "", "hipBLASLt", "TFLOP/s", "algos", "rel-L2");
    std::vector<Shape> todo;
    if (sweep) todo.assign(std::begin(ksweep), std::end(ksweep)); else todo.assign(std::begin(shapes), std::end(shapes));
    for (const Shape& sh : todo) {
        const long M = sh.M, N = sh.N, K = sh.K;
        const long na = M * K, nw = N * K, no = M * N;
        std::vector<half_t> ha(na), hw(nw), hr(no);
        std::vector<float> hb(N);
        std::vector<half_t> hb16(N);
        const float wsc = 1.7f / std::sqrt((float)K);
        fill(ha, 1.0f); fill(hw, wsc); fill(hr, 1.0f);
        for (long i = 0; i < N; ++i) { hb[i] = frand() * 0.1f; hb16[i] = (half_t)hb[i]; hb[i] = (float)hb16[i]; }   // both sides see the fp16-representable bias
        half_t *a, *w, *r, *o_e, *o_l, *b16;
        float* b;
        HIP_OK(hipMalloc(&a, na * 2)); HIP_OK(hipMemcpy(a, ha.data(), na * 2, hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&w, nw * 2)); HIP_OK(hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&r, no * 2)); HIP_OK(hipMemcpy(r, hr.data(), no * 2, hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&b, N * 4)); HIP_OK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&b16, N * 2)); HIP_OK(hipMemcpy(b16, hb16.data(), N * 2, hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&o_e, no * 2)); HIP_OK(hipMalloc(&o_l, no * 2));
        const double flop = 2.0 * M * N * K;

        // ---- engine ----------------------------------------------------------------------------------------------------------
        sdmi_conv_desc d;
        memset(&d, 0, sizeof d);
        const int rows_per_img = sh.M / 16;
        d.a0 = a; d.w = w; d.bias = b; d.resid = sh.resid ? r : nullptr; d.out = o_e;
        d.c0 = sh.K; d.lda0 = sh.K;
        d.B = 16; d.Hi = rows_per_img; d.Wi = 1; d.Ho = rows_per_img; d.Wo = 1;
        d.taps = 1; d.stride = 1; d.pad = 0; d.N = sh.N; d.n_real = sh.N;
        d.ldo = sh.N; d.ldr = sh.N; d.alpha = 1.0f; d.batch = 1;
        const int64_t skb = sdmi_conv_splitk_workspace_bytes(sh.M, sh.N, sh.K, 1);
        void* skw = nullptr;
        if (skb > 0) { HIP_OK(hipMalloc(&skw, skb)); d.splitk_workspace = skw; d.splitk_workspace_bytes = skb; }
        SDMI_OK(sdmi_conv_gemm(&d, stream));
        HIP_OK(hipStreamSynchronize(stream));
        float ms_e = 0.f;
        SDMI_OK(sdmi_bench_conv_gemm(&d, iters, &ms_e, stream));

        // ---- hipBLASLt: row-major D[M][N] = A[M][K] W[N][K]^T  ==  column-major D^T (N x M) = W (op T: N x K) A^T (K x M) --------
        hipblasLtMatmulDesc_t md;
        LT_OK(hipblasLtMatmulDescCreate(&md, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        const hipblasOperation_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
        LT_OK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof opT));
        LT_OK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof opN));
        const hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
        LT_OK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof epi));
        const void* bias_ptr = b16;
        LT_OK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias_ptr, sizeof bias_ptr));
        const hipDataType bias_t = HIP_R_16F;
        LT_OK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bias_t, sizeof bias_t));
        hipblasLtMatrixLayout_t la, lb, lc;
        LT_OK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16F, K, N, K));      // W as stored: K x N column-major (op T -> N x K)
        LT_OK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16F, K, M, K));      // A as stored: K x M column-major
        LT_OK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_16F, N, M, N));      // D^T: N x M column-major = D row-major
        hipblasLtMatmulPreference_t pref;
        LT_OK(hipblasLtMatmulPreferenceCreate(&pref));
        LT_OK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof ws_bytes));
        hipblasLtMatmulHeuristicResult_t res[NALGO];
        int nres = 0;
        hipblasStatus_t hs = hipblasLtMatmulAlgoGetHeuristic(lt, md, la, lb, lc, lc, pref, NALGO, res, &nres);
        const float alpha = 1.0f, beta = sh.resid ? 1.0f : 0.0f;
        float best = 1e30f;
        int timed = 0;
        if (hs == HIPBLAS_STATUS_SUCCESS) {
            for (int i = 0; i < nres; ++i) {
                if (res[i].state != HIPBLAS_STATUS_SUCCESS || res[i].workspaceSize > ws_bytes) continue;
                auto launch = [&]() {
                    return hipblasLtMatmul(lt, md, &alpha, w, la, a, lb, &beta, sh.resid ? (const void*)r : (const void*)o_l, lc, o_l, lc, &res[i].algo, ws, ws_bytes, stream);
                };
                if (launch() != HIPBLAS_STATUS_SUCCESS) continue;
                HIP_OK(hipStreamSynchronize(stream));
                HIP_OK(hipEventRecord(e0, stream));
                for (int it = 0; it < iters; ++it) launch();
                HIP_OK(hipEventRecord(e1, stream));
                HIP_OK(hipEventSynchronize(e1));
                float ms = 0.f;
                HIP_OK(hipEventElapsedTime(&ms, e0, e1));
                ms /= iters;
                ++timed;
                if (ms < best) best = ms;
            }
        }
        // compare (the last timed algorithm's output is in o_l; every algorithm computes the same product)
        double rel = -1.0;
        if (timed) {
            std::vector<half_t> ge(no), gl(no);
            HIP_OK(hipMemcpy(ge.data(), o_e, no * 2, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(gl.data(), o_l, no * 2, hipMemcpyDeviceToHost));
            double num = 0, den = 0;
            for (long i = 0; i < no; ++i) { const double x = (double)(float)ge[i], y = (double)(float)gl[i]; num += (x - y) * (x - y); den += x * x; }
            rel = std::sqrt(num / (den > 0 ? den : 1));
        }
        printf("%-52s %10.1f %10.1f %8s ", sh.name, ms_e * 1e3, flop / (ms_e * 1e-3) / 1e12, "");
        if (timed) printf("%10.1f %10.1f %6d %9.2e\n", best * 1e3, flop / (best * 1e-3) / 1e12, timed, rel);
        else printf("%10s %10s %6d %9s   (heuristic status %d, %d results)\n", "-", "-", 0, "-", (int)hs, nres);
        fflush(stdout);
        hipblasLtMatmulPreferenceDestroy(pref);
        hipblasLtMatrixLayoutDestroy(la); hipblasLtMatrixLayoutDestroy(lb); hipblasLtMatrixLayoutDestroy(lc);
        hipblasLtMatmulDescDestroy(md);
        if (skw) HIP_OK(hipFree(skw));
        HIP_OK(hipFree(a)); HIP_OK(hipFree(w)); HIP_OK(hipFree(r)); HIP_OK(hipFree(b)); HIP_OK(hipFree(b16)); HIP_OK(hipFree(o_e)); HIP_OK(hipFree(o_l));
    }
    hipblasLtDestroy(lt);
    return 0;
}
