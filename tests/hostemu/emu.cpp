// emu.cpp — the runtime of the HOST-EMULATED libsdmi (TEST INFRASTRUCTURE; tests/hostemu/build.py links it with the library's own
// capi.cpp, engine.cpp and the four kernel files compiled as plain C++ against tests/hostemu/hip/hip_runtime.h):
//   * the block runner: every GPU thread of the running block is a fiber, switched in user space, round-robin (deterministic); a fiber
//     parked on a barrier is skipped until the barrier's generation moves; a returned thread leaves its barriers; all-parked = deadlock;
//   * what `extern __shared__` arrays resolve to, and a stand-in for prof.cpp that records launch names and host time;
//   * thin emu_* entry points for the kernels that have no C-ABI entry of their own (launch functions called on host buffers by
//     tests/test_cpu_kernel_emulation.py).
// Nothing here is linked into, or imported by, the product.
#include "common.h"
#include "prof.h"
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <map>
#include <vector>

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
thread_local EmuBlock* emu_block = nullptr;
thread_local unsigned emu_tid = 0;
int emu_threaded = 0;

// ---- the block runner: one fiber per GPU thread of the running block, switched in user space ------------------------------------------
// emu_switch(&save_sp, to_sp): push the callee-saved registers, store the stack pointer, load the other one, pop, return
#if defined(__x86_64__)
extern "C" void emu_switch(void** save_sp, void* to_sp);
asm(".text\n.globl emu_switch\n.type emu_switch,@function\nemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size emu_switch, .-emu_switch\n");
#else
#error "tests/hostemu: the fiber switch is written for x86-64"
#endif
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define EMU_ASAN 1
#include <sanitizer/common_interface_defs.h>
#endif
#endif
#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define EMU_TSAN 1
#include <sanitizer/tsan_interface.h>
#endif
#endif
#ifdef EMU_TSAN                                                // SDMI_HOSTEMU_TSAN=1 builds: every GPU thread is a sanitizer fiber; switches establish NO
#define EMU_TSAN_SWITCH(f) __tsan_switch_to_fiber((f), 1)      // ordering (flag 1 = no_sync) — only barriers (emu_barrier) and block start / end do;
#define EMU_TSAN_RESUME(f) __tsan_switch_to_fiber((f), 0)      // scheduler -> thread carries the scheduler's own writes (threadIdx ...), never another thread's
#else
#define EMU_TSAN_SWITCH(f) ((void)0)
#define EMU_TSAN_RESUME(f) ((void)0)
#endif
#ifdef EMU_ASAN                                                // SDMI_HOSTEMU_ASAN=1 builds: tell the sanitizer about every stack switch
#define EMU_ASAN_START(save, bottom, size) __sanitizer_start_switch_fiber((save), (bottom), (size))
#define EMU_ASAN_FINISH(save, bottom, size) __sanitizer_finish_switch_fiber((save), (bottom), (size))
#else
#define EMU_ASAN_START(save, bottom, size) ((void)0)
#define EMU_ASAN_FINISH(save, bottom, size) ((void)0)
#endif
namespace {
constexpr size_t kFiberStack = 512 << 10;
struct EmuRunner {
    EmuBlock blk;
    char* stacks = nullptr;                                    // 1024 fiber stacks, mapped once (pages are touched on use)
    void* sp[1024];
    dim3 tidx[1024];
    bool done[1024];
    void* sched_sp = nullptr;
    void* sched_fake = nullptr;                                // sanitizer bookkeeping (unused otherwise)
    void* fiber_fake[1024];
    void* tsan_fiber[1024] = {};                               // (thread-sanitizer builds)
    void* tsan_sched = nullptr;
    char tsan_begin = 0, tsan_end = 0;                         // sync variables: block start happens-before every thread, every thread before block end
    const void* sched_bottom = nullptr;
    size_t sched_size = 0;
    unsigned cur = 0, live = 0;
    unsigned long progress = 0;
    dim3 b;
    const std::function<void()>* body = nullptr;
};
EmuRunner g_run;
EMU_RUNTIME void emu_thread_exit(EmuRunner& r, unsigned t) {               // a returned thread leaves its wave's and the block's barriers
    r.done[t] = true;
    --r.live;
    ++r.progress;
    for (unsigned id : {t >> 6, 16u}) {
        EmuBlock& k = r.blk;
        if (--k.need[id] > 0 && k.arrived[id] >= k.need[id]) { k.arrived[id] = 0; ++k.gen[id]; }
    }
}
EMU_RUNTIME void emu_fiber_main() {
    EmuRunner& r = g_run;
    EMU_ASAN_FINISH(nullptr, &r.sched_bottom, &r.sched_size);
#ifdef EMU_TSAN
    __tsan_acquire(&r.tsan_begin);
#endif
    (*r.body)();
#ifdef EMU_TSAN
    __tsan_release(&r.tsan_end);
#endif
    emu_thread_exit(r, r.cur);
    EMU_TSAN_SWITCH(r.tsan_sched);
    EMU_ASAN_START(nullptr, r.sched_bottom, r.sched_size);     // (null: this fiber's stack is not returned to)
    emu_switch(&r.sp[r.cur], r.sched_sp);
    std::abort();                                              // a finished fiber is never resumed
}
}  // namespace
EMU_RUNTIME void emu_yield() {
    EmuRunner& r = g_run;
    const unsigned me = r.cur;
    EMU_ASAN_START(&r.fiber_fake[me], r.sched_bottom, r.sched_size);
    EMU_TSAN_SWITCH(r.tsan_sched);
    emu_switch(&r.sp[me], r.sched_sp);
    EMU_ASAN_FINISH(r.fiber_fake[me], &r.sched_bottom, &r.sched_size);
}
EMU_RUNTIME void emu_run_block_threaded(dim3 g, dim3 b, dim3 bi, const std::function<void()>& body) {
    EmuRunner& r = g_run;
    const unsigned nt = b.x * b.y * b.z;
    if (nt > 1024 || emu_block) std::abort();                  // (no nested launches)
    if (!r.stacks) {
        r.stacks = (char*)mmap(nullptr, 1024 * kFiberStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (r.stacks == (char*)MAP_FAILED) std::abort();
    }
    EmuBlock& k = r.blk;
    const unsigned nw = (nt + 63) / 64;
    for (unsigned w = 0; w < 17; ++w) { k.arrived[w] = 0; k.gen[w] = 0; k.need[w] = 0; }
    for (unsigned w = 0; w < nw; ++w) k.need[w] = k.wave_lanes[w] = std::min(64u, nt - w * 64);
    k.need[16] = nt;
    for (unsigned t = 0; t < nt; ++t) {
        void** top = (void**)(r.stacks + (size_t)(t + 1) * kFiberStack);     // 16-byte aligned
        top[-1] = nullptr;                                     // the fiber entry's (never used) return address
        top[-2] = (void*)&emu_fiber_main;                      // popped by emu_switch's ret; rsp is then 8 mod 16, as after a call
        for (int i = 3; i <= 8; ++i) top[-i] = nullptr;        // rbp rbx r12 .. r15
        r.sp[t] = (void*)(top - 8);
        r.done[t] = false;
        k.wait_id[t] = 16; k.wait_gen[t] = ~0u;                // runnable
        k.mfma_parity[t] = 0;
        r.tidx[t] = dim3(t % b.x, (t / b.x) % b.y, t / (b.x * b.y));
    }
    r.b = b; r.body = &body; r.live = nt;
#ifdef EMU_TSAN
    r.tsan_sched = __tsan_get_current_fiber();
    for (unsigned t = 0; t < nt; ++t) if (!r.tsan_fiber[t]) r.tsan_fiber[t] = __tsan_create_fiber(0);
    __tsan_release(&r.tsan_begin);
#endif
    gridDim = g; blockDim = b; blockIdx = bi;
    emu_block = &k;
    while (r.live) {
        const unsigned long before = r.progress;
        for (unsigned t = 0; t < nt; ++t) {
            if (r.done[t] || k.gen[k.wait_id[t]] == k.wait_gen[t]) continue;
            r.cur = t; emu_tid = t;
            threadIdx = r.tidx[t];
            k.wait_gen[t] = ~0u;
            EMU_ASAN_START(&r.sched_fake, r.stacks + (size_t)t * kFiberStack, kFiberStack);
            EMU_TSAN_RESUME(r.tsan_fiber[t]);
            emu_switch(&r.sched_sp, r.sp[t]);
            EMU_ASAN_FINISH(r.sched_fake, nullptr, nullptr);
            ++r.progress;
        }
        if (r.progress == before) {                            // every live thread is parked on a barrier that cannot complete
            std::fprintf(stderr, "hostemu: deadlock in block (%u,%u,%u): %u threads parked\n", bi.x, bi.y, bi.z, r.live);
            std::abort();
        }
    }
    emu_block = nullptr;
#ifdef EMU_TSAN
    __tsan_acquire(&r.tsan_end);
#endif
}

namespace sdmi {
// the per-launch profiler is replaced (prof.cpp is not linked): the launch name tells which tile / kernel family a launch really took
static std::string g_emu_last_launch;
static std::vector<std::pair<std::string, double>> g_emu_launch_log;      // (name, host milliseconds)
static bool g_emu_log_on = false;
static double g_emu_t0 = 0.0;
static double emu_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
bool prof_enabled() { return true; }
void prof_begin() { g_emu_launch_log.clear(); g_emu_log_on = true; }
std::string prof_end() {                                       // {"kernels": [{"name", "launches", "ms" (HOST time of the emulation), ...}]} like prof.cpp, in first-launch order
    g_emu_log_on = false;
    std::vector<std::string> order;
    std::map<std::string, std::pair<int, double>> acc;
    for (const auto& e : g_emu_launch_log) {
        auto& a = acc[e.first];
        if (a.first++ == 0) order.push_back(e.first);
        a.second += e.second;
    }
    std::string out = "{\"kernels\": [";
    for (size_t i = 0; i < order.size(); ++i)
        out += std::string(i ? ", " : "") + "{\"name\": \"" + order[i] + "\", \"launches\": " + std::to_string(acc[order[i]].first) + ", \"ms\": " +
               std::to_string(acc[order[i]].second) + ", \"flops\": 0, \"bytes\": 0}";
    return out + "]}";
}
void prof_mark_start(const char* name, double, double, hipStream_t) {
    g_emu_last_launch = name ? name : "";
    g_emu_t0 = emu_now_ms();
}
void prof_mark_stop(hipStream_t) {
    if (g_emu_log_on) g_emu_launch_log.emplace_back(g_emu_last_launch, emu_now_ms() - g_emu_t0);
}
alignas(16) float sa[96 * 1024 / 4];                   // what `extern __shared__ float sa[]` (small_linear_lds: up to 96 KB of activations) resolves to
alignas(16) char smem[160 * 1024 + 16384];            // what `extern __shared__ char smem[]` of the GEMM / attention kernels resolves to (one workgroup at a time)
alignas(16) float sc[16384];                            // `extern __shared__ float sc[]` of the generic attention kernel (M scores)
}  // namespace sdmi

using namespace sdmi;
#include "../../include/sdmi.h"
extern "C" {
// the implicit-GEMM convolution / linear launch of the C ABI (capi.cpp desc_to_p + launch_gemm), threaded mode only
struct emu_gemm_extras {                                       // the GemmP fields the public descriptor does not carry (engine-internal fusions)
    float* stats_out; int stats_cpg; int stats_nchunk;         // GroupNorm partial sums from the epilogue; stats_nchunk is written back
    const float* ln_stats; const float* ln_s; int ln_np; float ln_inv_c, ln_eps;   // EP_LNFOLD consumer side
    float* lnp_out; int lnp_np;                                // LayerNorm partial row sums, producer side; lnp_np is written back
    float bias_scale;
    uint16_t* out_lo; const uint16_t* resid_lo;                // (hi, lo) stream tensors (EP_HILO): the lo halves of the output / the residual
};
int emu_conv_gemm(const sdmi_conv_desc* d, int cfg, int split, int korder, emu_gemm_extras* x) {
    GemmP p{};
    p.a0 = (const half_t*)d->a0; p.a1 = (const half_t*)d->a1; p.w = (const half_t*)d->w;
    p.bias = (const float*)d->bias; p.rowbias = (const float*)d->rowbias; p.resid = (const half_t*)d->resid; p.out = d->out;
    p.c0 = d->c0; p.c1 = d->a1 ? d->c1 : 0; p.cin = p.c0 + p.c1;
    p.lda0 = d->lda0 ? d->lda0 : d->c0; p.lda1 = d->lda1 ? d->lda1 : d->c1;
    p.Hi = d->Hi; p.Wi = d->Wi; p.Ho = d->Ho; p.Wo = d->Wo;
    p.taps = d->taps; p.stride = d->stride ? d->stride : 1; p.pad = d->pad; p.up = d->up;
    p.M = d->B * d->Ho * d->Wo; p.N = d->N; p.K = p.taps * p.cin;
    p.ldo = d->ldo; p.ldr = d->ldr; p.ldw = p.K; p.ldrb = d->N;
    p.rows_per_batch = d->Ho * d->Wo;
    p.n_real = d->n_real ? d->n_real : d->N;
    p.flags = d->flags;
    p.alpha = d->alpha == 0.f ? 1.f : d->alpha;
    p.bias_scale = 1.f;
    if (d->splitk_workspace) p.splitk_ws = (float*)d->splitk_workspace;
    g_force_gemm_cfg = cfg;                                   // -1: the engine's own choice (tuned table / score model)
    g_force_gemm_split = split;
    g_conv_korder = korder;
    p.a_bs = d->a_bs; p.w_bs = d->w_bs; p.o_bs = d->o_bs; p.r_bs = d->r_bs;
    if (x) {
        p.stats_out = x->stats_out; p.stats_cpg = x->stats_cpg;
        p.ln_stats = x->ln_stats; p.ln_s = x->ln_s; p.ln_np = x->ln_np; p.ln_inv_c = x->ln_inv_c; p.ln_eps = x->ln_eps;
        p.lnp_out = x->lnp_out;
        if (x->bias_scale != 0.f) p.bias_scale = x->bias_scale;
        if (x->out_lo || x->resid_lo) {                       // as engine.cpp run_conv hands them over
            p.flags |= EP_HILO;
            p.ln_stats = reinterpret_cast<const float*>(x->out_lo);
            p.ln_s = reinterpret_cast<const float*>(x->resid_lo);
        }
    }
    int nchunk = 0, np = 0;
    const int rc = launch_gemm(p, d->batch > 0 ? d->batch : 1, d->force_generic == 1, d->force_generic != 2, nullptr, &nchunk, &np);
    if (x) { x->stats_nchunk = nchunk; x->lnp_np = np; }
    return rc;
}
// flash attention over Q [B, N, ldq], K [B, M, ldk], V^T [B, H*D, vt_ld] (include/sdmi.h sdmi_attention_vt); occ / kvt: the kernel-form knobs
int emu_attention(const uint16_t* q, const uint16_t* k, const uint16_t* vt, uint16_t* out, int B, int H, int N, int M, int D, int ldq, int ldk,
                  int vt_ld, int ldo, float scale, int causal, int generic, int occ, int kvt) {
    AttnP p{};
    p.q = (const half_t*)q; p.k = (const half_t*)k; p.vt = (const half_t*)vt; p.out = (half_t*)out;
    p.B = B; p.H = H; p.N = N; p.M = M; p.D = D; p.ldq = ldq; p.ldk = ldk; p.vt_ld = vt_ld; p.ldo = ldo;
    p.scale_log2 = scale * 1.44269504088896340736f;
    p.causal = causal;
    g_attn_occ = occ; g_attn_kvt = kvt;
    return launch_attention(p, generic != 0, nullptr);
}
int emu_transpose_v(const uint16_t* v, uint16_t* vt, int B, int H, int M, int D, int ldv, int Mpad) {
    return launch_transpose_v((const half_t*)v, (half_t*)vt, B, H, M, D, ldv, Mpad, nullptr);
}
int64_t emu_splitk_ws_bytes(int M, int N, int K) { return (int64_t)gemm_splitk_ws_bytes(M, N, K, 1); }
// control for the sanitizer builds: neighbours exchange through LDS with / without the barrier that orders them
__global__ void emu_race_control_kernel(int* out, int with_barrier, int salt) {
    __shared__ int s[64];
    s[threadIdx.x] = (int)threadIdx.x * 3 + salt;
    if (with_barrier) __syncthreads();
    out[threadIdx.x] = s[(threadIdx.x + 1) & 63];
}
int emu_race_control(int* out, int with_barrier, int salt) {
    hipLaunchKernelGGL(emu_race_control_kernel, dim3(1), dim3(64), 0, nullptr, out, with_barrier, salt);
    return 0;
}
const char* emu_last_error() { return get_error(); }
const char* emu_last_launch() { return g_emu_last_launch.c_str(); }
void emu_set_threaded(int on) { emu_threaded = on; }
// ---- kernels that need the threaded mode ------------------------------------------------------------------------------------------
int64_t emu_groupnorm_ws_bytes(int B, int HW, int groups) { return groupnorm_ws_bytes(B, HW, groups); }
int emu_groupnorm(const uint16_t* x0, const uint16_t* x1, int c0, int c1, const float* gamma, const float* beta, uint16_t* out, int B, int HW, int groups,
                  float eps, int silu, float* ws) {
    return launch_groupnorm((const half_t*)x0, (const half_t*)x1, c0, c1, gamma, beta, (half_t*)out, B, HW, groups, eps, silu != 0, ws, nullptr);
}
// (hi, lo) input pairs (engine option "residual_fp32"); pre_nchunk > 0: `ws` already holds the producing GEMM's partial sums
int emu_groupnorm_hilo(const uint16_t* x0, const uint16_t* x1, int c0, int c1, const float* gamma, const float* beta, uint16_t* out, int B, int HW,
                       int groups, float eps, int silu, float* ws, int pre_nchunk, const uint16_t* x0_lo, const uint16_t* x1_lo) {
    return launch_groupnorm((const half_t*)x0, (const half_t*)x1, c0, c1, gamma, beta, (half_t*)out, B, HW, groups, eps, silu != 0, ws, nullptr,
                            pre_nchunk, (const half_t*)x0_lo, (const half_t*)x1_lo);
}
int emu_nchw_to_nhwc_lo(const float* x, uint16_t* out, int B, int C, int HW, int cpad, int lo_ch) {
    return launch_nchw_to_nhwc(x, 1, (half_t*)out, B, C, HW, cpad, 1.0f, nullptr, nullptr, nullptr, lo_ch != 0);
}
int emu_layernorm(const uint16_t* x, const float* gamma, const float* beta, uint16_t* out, int64_t rows, int C, float eps) {
    return launch_layernorm((const half_t*)x, gamma, beta, (half_t*)out, rows, C, eps, nullptr);
}
int emu_ln_rowstats(const uint16_t* x, float* stats, int64_t rows, int C, float eps) { return launch_ln_rowstats((const half_t*)x, stats, rows, C, eps, nullptr); }
int emu_softmax_rows(const float* in, uint16_t* out, int64_t rows, int cols, int ldi, int ldo) {
    return launch_softmax_rows(in, (half_t*)out, rows, cols, ldi, ldo, nullptr);
}
int emu_dpm_error(const float* lo, const float* hi, const float* prev, float atol, float rtol, float* partial256, int64_t n) {
    return launch_dpm_error(lo, hi, prev, atol, rtol, partial256, n, nullptr);
}
int emu_small_linear(const float* a, const uint16_t* w, const float* bias, const float* add, float* out, int B, int N, int K, int lda, int ldo,
                     int silu_in, int silu_out) {
    return launch_small_linear(a, (const half_t*)w, bias, add, out, B, N, K, lda, ldo, silu_in != 0, silu_out != 0, nullptr);
}
int emu_ctx_compare(const void* src, int dtype, const uint16_t* cached, int B, int L, int Lpad, int C, int* gate) {
    return launch_ctx_compare(src, dtype, (const half_t*)cached, B, L, Lpad, C, gate, nullptr);
}
int emu_ctx_update_gated(const void* src, int dtype, uint16_t* cached, int B, int L, int Lpad, int C, const int* gate) {
    return launch_ctx_update_gated(src, dtype, (half_t*)cached, B, L, Lpad, C, gate, nullptr);
}
int emu_slerp(float* out, const float* low, const float* high, float val, int C, int H, int W, float* scratch) {
    return launch_slerp(out, low, high, val, C, H, W, scratch, nullptr);
}
int emu_philox(float* out, int64_t n, uint64_t seed, uint32_t offset) { return launch_philox(out, n, seed, offset, nullptr); }
int emu_image_to_u8(const float* img, uint8_t* out, int B, int C, int H, int W) { return launch_image_to_u8(img, out, B, C, H, W, nullptr); }
int emu_latent_resize(const float* in, float* out, int planes, int hi, int wi, int ho, int wo, int mode) {
    return launch_latent_resize(in, out, planes, hi, wi, ho, wo, mode, nullptr);
}
int emu_cfg_prepare(const float* x, const float* c_in, void* xin, int out_dtype, int B, int reps, int64_t chw) {
    return launch_cfg_prepare(x, c_in, xin, out_dtype, B, reps, chw, nullptr);
}
int emu_cfg_prepare_concat(const float* x, const float* c_in, const float* cond, void* xin, int out_dtype, int B, int reps, int C, int Cc, int64_t hw,
                           uint32_t zero_reps) {
    return launch_cfg_prepare_concat(x, c_in, cond, xin, out_dtype, B, reps, C, Cc, hw, zero_reps, nullptr);
}
int emu_cfg_combine(const float* x, const float* eps, const float* c_out, float cond_scale, int mode, const float* mask, const float* nmask,
                    const float* init, float* den, int B, int64_t chw) {
    return launch_cfg_combine(x, eps, c_out, cond_scale, mode, mask, nmask, init, den, B, chw, nullptr);
}
int emu_cfg_combine_affine(const float* x, const float* out, const float* c_out, const float* c_skip, float cond_scale, const float* mask,
                           const float* nmask, const float* init, float* den, int B, int64_t chw) {
    return launch_cfg_combine_affine(x, out, c_out, c_skip, cond_scale, mask, nmask, init, den, B, chw, nullptr);
}
int emu_euler_step(float* x, const float* den, const float* noise, float sigma, float sigma_down, float sigma_up, float s_noise, int64_t n) {
    return launch_euler_step(x, den, noise, sigma, sigma_down, sigma_up, s_noise, n, nullptr);
}
int emu_dpmpp2m_step(float* x, const float* den, const float* old, float ratio, float em1, float c1, float c2, int64_t n) {
    return launch_dpmpp2m_step(x, den, old, ratio, em1, c1, c2, n, nullptr);
}
int emu_ddim_step(float* x, const float* e, const float* noise, float* pred_x0, float a_t, float a_prev, float sigma_t, float somat, int64_t n) {
    return launch_ddim_step(x, e, noise, pred_x0, a_t, a_prev, sigma_t, somat, n, nullptr);
}
int emu_axpby(float* y, const float* x, float a, const float* z, float b, int64_t n) { return launch_axpby(y, x, a, z, b, n, nullptr); }
int emu_lincomb(float* out, const float* const* terms, const float* coefs, int n_terms, int64_t n) {
    return launch_lincomb(out, terms, coefs, n_terms, n, nullptr);
}
int emu_mask_blend(float* x, const float* init, const float* mask, const float* nmask, int64_t n) { return launch_mask_blend(x, init, mask, nmask, n, nullptr); }
int emu_timestep_embedding(const float* t, float* out, int B, int dim) { return launch_timestep_embedding(t, 1, out, B, dim, nullptr); }
int emu_pack_conv_weight(const float* w, uint16_t* out_f16_bits, int O, int I, int kh, int kw, int O_pad, int I_pad, int geglu) {
    return launch_pack_conv_weight(w, 1, (half_t*)out_f16_bits, O, I, kh, kw, O_pad, I_pad, geglu, nullptr);
}
int emu_pack_bias(const float* b, float* out, int O, int O_pad, int geglu) { return launch_pack_bias(b, 1, out, O, O_pad, geglu, nullptr); }
int emu_nchw_to_nhwc(const float* x, uint16_t* out_f16_bits, int B, int C, int HW, int cpad, float scale) {
    return launch_nchw_to_nhwc(x, 1, (half_t*)out_f16_bits, B, C, HW, cpad, scale, nullptr, nullptr, nullptr);
}
int emu_weight_kron(float* out, const float* w, const float* w1, const float* w2, int r1, int c1, int r2, int c2, int k, float scale) {
    return launch_weight_kron(out, w, w1, w2, r1, c1, r2, c2, k, scale, nullptr);
}
int emu_act_f16(uint16_t* x, int64_t n, int kind) { return launch_act_f16((half_t*)x, n, kind, nullptr); }
int emu_axpy_f16(uint16_t* y, const uint16_t* x, const uint16_t* h, float a, int64_t n) {
    return launch_axpy_f16((half_t*)y, (const half_t*)x, (const half_t*)h, a, n, nullptr);
}
int emu_lora_merge(float* out, const void* w, int w_dtype, const void* up, int up_dtype, const void* down, int down_dtype, int rows, int cols, int rank,
                   float scale) {
    return launch_lora_merge(out, w, w_dtype, up, up_dtype, down, down_dtype, rows, cols, rank, scale, nullptr);
}
int emu_weight_hadamard(float* out, const float* w, const float* a, const float* b, float scale, int64_t n) {
    return launch_weight_hadamard(out, w, a, b, scale, n, nullptr);
}
int emu_weight_dora(float* out, const float* w, const float* delta, const float* dora_scale, int rows, int cin, int k, float mult) {
    return launch_weight_dora(out, w, delta, dora_scale, rows, cin, k, mult, nullptr);
}
int emu_weight_ia3(float* out, const float* w, const float* v, int rows, int cols, int on_input, float scale) {
    return launch_weight_ia3(out, w, v, rows, cols, on_input, scale, nullptr);
}
}
