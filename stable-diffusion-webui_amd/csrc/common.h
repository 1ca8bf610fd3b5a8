// common.h — shared types and helpers for the gfx950 kernels of libsdmi.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

namespace sdmi {

// GEGLU's gate: val * gelu(g), gelu(g) = g * Phi(g) (ldm's GEGLU uses F.gelu's default, the exact erf form).  Phi is evaluated as a
// logistic of an odd quintic,  Phi(g) ~ 1 / (1 + exp(-g (c0 + c1 g^2 + c2 g^4))),  fitted (minimax over [-8, 8], polynomial argument
// clamped to +-10 where the logistic has long saturated) to |g Phi(g) - gelu(g)| <= 2.6e-5 and a relative error <= 2e-4 near zero —
// below the fp16 rounding of the product it feeds (4.9e-4 relative) (tests/test_cpu_kernel_emulation.py pins the bound).  10 VALU operations,
// two of them transcendental, against 18 / two for the A&S 7.1.26 erf it replaces (gemm.hip gelu_erf, kept for the plain GELU
// epilogue): the GEGLU epilogues are VALU-issue bound next to the MFMAs (profiles/r05_*), so the count is the cost.
#ifndef SDMI_GELU_SIG
#define SDMI_GELU_SIG 1
#endif
__device__ __forceinline__ float geglu_gate(float val, float g) {
    const float gc = fminf(fmaxf(g, -10.0f), 10.0f);
    const float g2 = gc * gc;
    // -log2(e) * (1.59501577, 7.40112920e-2, -7.03033577e-4)
    float p = fmaf(1.01426306e-3f, g2, -0.106775724f);
    p = fmaf(p, g2, -2.30112135f);
    const float e = __builtin_amdgcn_exp2f(p * gc);          // exp(-g (c0 + c1 g^2 + c2 g^4))
    return (val * g) * __builtin_amdgcn_rcpf(1.0f + e);
}

void set_error(const std::string& msg);
const char* get_error();

#define SDMI_CHECK_HIP(expr)                                                                       \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            ::sdmi::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                  \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

#define SDMI_REQUIRE(cond, msg)                                                                    \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            ::sdmi::set_error(std::string("requirement failed: ") + #cond + " — " + (msg));        \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute (MaxDynamicSharedMemorySize) is a per-DEVICE setting, and one process may drive several devices (one engine +
// one host thread per device: parallel.DevicePool) — launch helpers remember what they set per (call site, device).  need(want) is true
// when `want` exceeds what this call site has set on the CURRENT device so far (a process-wide flag skipped the call on the
// second device and the launch failed with more than 64 KB of dynamic LDS).
struct PerDeviceOnce {
    int level[64] = {};
    bool need(int want = 1) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) d = 0;
        d &= 63;
        if (level[d] >= want) return false;
        level[d] = want;
        return true;
    }
};

// A 256-byte zeroed device buffer per device used as the source of padding lanes in LDS-direct loads.
// Device page of zeros that padded / out-of-image operand lanes are pointed at.  32 KB: the ping-pong GEMM adds a channel
// offset of up to Cin halfs to it instead of re-selecting the pointer per load.
constexpr int kZeroPageHalfs = 16384;
const half_t* zero_page();

// ---- GEMM / implicit conv parameters (kernel argument, POD) -----------------------------------------------
struct GemmP {
    const half_t* a0;
    const half_t* a1;
    const half_t* w;
    const float* bias;
    const float* rowbias;
    const half_t* resid;
    void* out;
    const half_t* zero;
    int c0, c1, cin;
    int lda0, lda1;
    int Hi, Wi, Ho, Wo;
    int taps, stride, pad, up;
    int M, N, K;
    int ldo, ldr;
    int ldw;              // row stride of W (elements), normally K
    int ldrb;             // batch stride of rowbias (elements), normally N
    int rows_per_batch;   // Ho*Wo
    int n_real;
    int n_valid;          // rows of W that exist (columns >= n_valid read zeros); normally N
    int flags;
    float alpha;
    long a_bs, w_bs, o_bs, r_bs;
    // split-K (deterministic): grid.y = splitk slices of the K loop, each writing an fp32 partial slab ws[slice][M][N];
    // splitk_reduce sums the slabs in slice order and applies the epilogue.  splitk <= 1: off.
    float* splitk_ws;
    int splitk;
    int splitk_steps;     // BK-steps per slice
    long long* dbg;       // tuning only: per-wave section timers of the ping-pong kernel (sdmi_debug_set gemm_dbg_lo/hi)
    // GroupNorm statistics of the OUTPUT tensor from this launch's epilogue (drops the consumer's statistics pass): per
    // (image, row chunk = this tile's BM rows, group) the sum and sum of squares of the fp16-rounded outputs, written to
    // stats_out[((b * stats_nchunk + chunk) * (N / stats_cpg) + g) * 2 + {0, 1}] — the layout gn_apply's prologue reduces.
    // Deterministic: registers -> 16-lane shuffle tree -> LDS -> one thread per group, all in fixed order.
    float* stats_out;
    int stats_cpg;        // channels per group of the consumer's GroupNorm
    int stats_nchunk;     // rows_per_batch / BM, set by launch_gemm when the fusion applies (0 = not produced)
    int korder;           // K walk of a 3x3 conv: 0 = tap-major (tap outer, channels inner: round 1), 1 = channel-block-major (64 channels
                          // outer, the 9 taps inner): the 9 shifted reads of one channel block follow each other, so they are served by the
                          // XCD's L2 instead of nine passes over the whole tile set's rows (PMC round 1: 2.3-3x the algorithmic fetch).
                          // Weights stay [N][tap][Cin]; only the order of the K tiles (and so the fp32 summation order) changes.
    int tile_order;       // 0: consecutive tile ids walk N first (they share the gathered A panel: the large-M layers); 1: M first
                          // (they share the WEIGHT panel: small-M / deep-K layers, where the weights are the HBM traffic and an
                          // XCD-local run of tiles must reuse them out of its own L2).  Chosen per launch in launch_gemm.
    float bias_scale;     // bias is added as bias * bias_scale (0 is read as 1): range-extended VAE decode, where the residual stream
                          // is carried at 1/64 scale (alpha scales the accumulator, bias_scale the bias)
    const int* gate;      // optional device flag: the launch is a no-op when *gate == 0 (context re-projection only if the context
                          // changed, decided on the device: no host synchronisation — engine.cpp unet_set_context)
    // EP_LNFOLD: the A operand is the UN-normalised input x of a LayerNorm whose affine part is folded into the weights
    // (w = fp16(W * gamma), bias = beta . W^T + b): the epilogue finishes the normalisation per row,
    //     out[m][n] = rstd[m] * (acc[m][n] - mean[m] * ln_s[n]) + bias[n],   ln_s[n] = sum_k w[n][k]
    // so the normalised tensor never exists in HBM (engine.cpp run_st, option "ln_fold").
    const float* ln_stats;   // [M][2]: (mean, rstd) of the input rows (launch_ln_rowstats) — or, with ln_np > 0, [M][ln_np][2] partial
                             // (sum, sum of squares) pairs written by the producing GEMM's epilogue (lnp_out), finished per row here
    const float* ln_s;       // [N]
    int ln_np;               // 0: ln_stats holds (mean, rstd); > 0: that many partial pairs per row
    float ln_inv_c, ln_eps;  // 1 / C and the LayerNorm eps, used with ln_np > 0
    // producer side: when lnp_out is set and the launch qualifies (16-byte plain epilogue, no split-K), the epilogue also writes the
    // per-row sums of its fp16-ROUNDED outputs over the tile's BN columns to lnp_out[(m * lnp_np + tile_n) * 2 + {0, 1}]
    float* lnp_out;
    int lnp_np;              // set by launch_gemm: N / BN of the chosen tile when the partials are produced, else 0
    // Engine option "residual_fp32" (flag EP_HILO): a tensor of the carried stream (ResBlock / transformer residual sums, the first conv's
    // output) as a (hi, lo) pair of fp16 tensors, x = hi + lo with hi = fp16(x) and lo = fp16(x - hi): ~22 bits of the fp32 sum survive
    // the store, every GEMM that takes the stream as its A operand reads hi alone (the MFMA operand is fp16 either way), norms and
    // residual adds read both.  The two extra pointers travel in the fields of the LayerNorm-fold CONSUMER, which such a launch never is
    // (a folded layer writes q / k / v / the GEGLU hidden tensor and carries no residual) — out_lo in `ln_stats`, resid_lo in `ln_s`:
    // two more kernel-argument pointers cost the 256-row ping-pong instantiations 2-12 SGPR spills (round 5).  Accessors: gemm_out_lo /
    // gemm_resid_lo below.  Round 6: split-K (the reduce pass carries the pair), the GroupNorm-statistics epilogue (sums of hi + lo) and
    // the 16-byte epilogue all take (hi, lo) launches — round 5 routed them through the 8-byte epilogue without any of the three.
};

enum { EP_OUT_F32 = 1, EP_GEGLU = 2, EP_NCHW = 4, EP_BIAS_ROW = 8,
       EP_QUICK_GELU = 16, EP_GELU = 32,               // activation on the biased result (CLIP MLP: x*sigmoid(1.702x) / erf GELU)
       EP_TRANSPOSE = 64,                              // store out^T per image: out[b][n][m - b*rows_per_batch] (row stride ldo): the V
                                                       // projection written as V^T [C][tokens] for the attention kernel — the MFMA
                                                       // operands swap roles so a lane owns 4 consecutive TOKENS of one channel
       EP_WRAP = 128,                                  // 3x3 taps wrap around the image instead of reading zero padding (p.tiling:
                                                       // Conv2d padding_mode = 'circular', modules/sd_hijack.py:311-318)
       EP_DBG_NO_BAR_A = 0x100, EP_DBG_NO_BAR_B = 0x200, EP_DBG_NO_GLDS = 0x400, EP_DBG_NO_VMWAIT = 0x800,
       EP_DBG_NO_DSREAD = 0x1000,      // 0x100..0x1000: tuning experiments only (sdmi_debug_set "gemm_dbgflags"): results are wrong
       EP_LNFOLD = 0x4000,             // see GemmP::ln_stats
       EP_HILO = 0x8000,               // (hi, lo) stream tensors: see GemmP (out_lo in ln_stats, resid_lo in ln_s)
       EP_NARROW = 0x2000              // 8-byte epilogue accesses (set by launch_gemm when the 16-byte form's alignment rules fail, or
                                       // by the "ep_wide" knob): gemm_epilogue's swap16 note
     };

__host__ __device__ inline half_t* gemm_out_lo(const GemmP& p) {
    return (p.flags & EP_HILO) ? reinterpret_cast<half_t*>(const_cast<float*>(p.ln_stats)) : nullptr;
}
__host__ __device__ inline const half_t* gemm_resid_lo(const GemmP& p) { return (p.flags & EP_HILO) ? reinterpret_cast<const half_t*>(p.ln_s) : nullptr; }

// stats_nchunk_out (optional): the number of row chunks per image of the GroupNorm partial sums written to p.stats_out, or 0 when the
// launch could not produce them (split-K, a tile spanning two images, a group straddling column tiles ...)
int launch_gemm(const GemmP& p, int batch, bool force_generic, bool use_glds, hipStream_t s, int* stats_nchunk_out = nullptr,
                int* lnp_np_out = nullptr);      // lnp_np_out: partial pairs per row written to p.lnp_out, 0 when not produced
extern int g_ep_wide;               // 1 (default): 16-byte epilogue accesses where the alignment allows; 0: always 8-byte
extern int g_small_linear_lds;      // 1 (default): small_linear stages the activation block in LDS (elementwise.hip); 0: wave-per-column form
extern int g_gn_small;              // 1 (default): single-launch register-resident GroupNorm for the small levels (norm.hip gn_fused_small_kernel)
extern int g_gn_fuse;               // 1 (default): GroupNorm statistics from the producing GEMM's epilogue where possible; 0: always a stats pass
// debug / tuning knobs (sdmi_debug_set): forced GEMM tile config (-1 = heuristic), attention KV tile (0 = heuristic)
// worst-case fp32 workspace a split-K launch of this shape may use (bytes); 0 when split-K would never be chosen
size_t gemm_splitk_ws_bytes(int M, int N, int K, int batch);
int gemm_set_override(const char* spec);   // "M,N,K,taps,kind:cfg:split;..." (kind 0 plain / 1 GEGLU / 2 transposed); "" clears
extern int g_force_gemm_cfg;
extern int g_shortk_gemm_cfg;
extern int g_shortk_max_k;          // the launches g_shortk_gemm_cfg applies to: taps == 1 and K <= this (default 448)
extern int g_geglu_gemm_cfg;        // tile config forced on the GEGLU (ff.net.0.proj) launches, -1 = heuristic
extern int g_conv_korder;           // 2 (default) row-shared walk where admitted, 0 tap-major, 1 channel-block-major (gemm.hip)
extern int g_conv_korder_default;   // value restored by sdmi_debug_set("conv_korder", -1)
extern int g_tile_order;            // -1 heuristic (default), 0 / 1 force
extern int g_vt_mode;               // 1 (default): V^T through EP_TRANSPOSE on token-major tiles; 0: weights-as-rows GEMM (round 1)
extern long g_gn_band_elems;      // tests: element count per image from which launch_groupnorm normalises in bands of rows (0 = 2^31)
extern int g_gn_apply_blocks;     // tests: workgroups per image of gn_apply_kernel (0 = launch_groupnorm's own choice)
extern int g_gemm_lin;              // 1 (default) = running-pointer K walk for 1x1 / linear launches on the 4-wave LDS-direct tiles
extern int g_gemm_pipe;             // 0 = two-stage kernels only, 3 = ping-pong 256-row tiles, 4 = also 128x320 (default)
extern int g_gemm_pipe_default;     // value restored by sdmi_debug_set("gemm_pipe", -1)
extern int g_force_gemm_split;      // 0 = heuristic, 1 = never split, k > 1 = force k slices where allowed
extern int g_attn_kvt;
extern int g_attn_occ;
extern int g_attn_lds_pad;
extern int g_attn_fold_min_m;
extern int g_attn_tau;
extern unsigned long long g_attn_dbg;   // device pointer (0 = off) for AttnP::dbg
extern int g_gemm_dbgflags;
extern unsigned long long g_gemm_dbg;   // device pointer (0 = off): 5 x int64 per wave of section cycle sums

// ---- attention --------------------------------------------------------------------------------------------
struct AttnP {
    const half_t* q;
    const half_t* k;
    const half_t* vt;
    half_t* out;
    int B, H, N, M, D;
    int ldq, ldk, vt_ld, ldo;
    float scale_log2;      // softmax scale * log2(e)
    int causal;            // 1: key j is visible to query i only if j <= i (CLIP text encoder; requires N == M)
    long long* dbg;        // tuning only (SDMI_ATTN_PARTS builds, attn_occ = 18): per-wave section cycle sums, 8 x int64 per wave
    float tau;             // set by launch_attention from g_attn_tau: slack (log2 units) a score may exceed the exponent base by before
                           // the kernels re-base (0: re-base whenever some query's running maximum moves; -1: the round-1 form also rescales O^T in every tile)
};
int launch_attention(const AttnP& p, bool force_generic, hipStream_t s);
// v [B, M, ldv] (head h at h*D) -> vt [B, H*D, Mpad] (zero padded)
int launch_transpose_v(const half_t* v, half_t* vt, int B, int H, int M, int D, int ldv, int Mpad, hipStream_t s);

// ---- the feed-forward chain of the transformer block (rowchain.hip) ------------------------------------------
bool rowchain_supports(int C);                      // row widths the fused chain is instantiated for (320)
size_t rowchain_ff_pack_bytes(int C, int hidden);
// w1 [2*hidden][C]: value rows then gate rows (permuted = false) or the engine's GEGLU row packing (true); b1 likewise (may be null)
int launch_rowchain_ff_pack(const half_t* w1, const float* b1, const half_t* w2, void* packs, int C, int hidden, bool permuted,
                            hipStream_t s);
// out = x + W2 GEGLU(W1 LN(x) + b1) + b2 (rows % 128 == 0)
int launch_rowchain_ff(const half_t* x, half_t* out, const float* gamma, const float* beta, const void* packs, const float* bias_out,
                       long rows, int C, int hidden, float eps, hipStream_t s);

// ---- norms ------------------------------------------------------------------------------------------------
// pre_nchunk > 0: `ws` already holds partial sums [B][pre_nchunk][groups][2] (written by the producing GEMM): skip the statistics pass
// x0_lo / x1_lo (engine option "residual_fp32"): the lo parts when the inputs are (hi, lo) fp16 pairs of the carried stream
int launch_groupnorm(const half_t* x0, const half_t* x1, int c0, int c1, const float* gamma, const float* beta,
                     half_t* out, int B, int HW, int groups, float eps, bool silu, float* ws, hipStream_t s, int pre_nchunk = 0,
                     const half_t* x0_lo = nullptr, const half_t* x1_lo = nullptr);
int64_t groupnorm_ws_bytes(int B, int HW, int groups);
int launch_layernorm(const half_t* x, const float* gamma, const float* beta, half_t* out, int64_t rows, int C,
                     float eps, hipStream_t s, const half_t* x_lo = nullptr);
// LayerNorm folded into the consuming GEMMs (GemmP::ln_stats): per-row (mean, rstd) of x [rows][C] -> stats [rows][2] ...
int launch_ln_rowstats(const half_t* x, float* stats, int64_t rows, int C, float eps, hipStream_t s);
// ... and the one-off weight fold: wf[n][k] = fp16(w[n][k] * gamma[k]) (k < C, 0 beyond), s[n] = sum_k wf[n][k],
// c[n] = sum_k beta[k] * w[n][k] + (bias ? bias[n] : 0); w / wf are packed [n_rows][K] (GEGLU row order included)
int launch_ln_fold_weights(const half_t* w, const float* gamma, const float* beta, const float* bias, half_t* wf, float* s_out,
                           float* c_out, int n_rows, int K, int C, hipStream_t s);

// ---- elementwise / misc -----------------------------------------------------------------------------------
int launch_philox(float* out, int64_t n, uint64_t seed, uint32_t offset, hipStream_t s);
int launch_cfg_prepare(const float* x, const float* c_in, void* xin, int out_dtype, int B, int reps, int64_t chw, hipStream_t s);
int launch_slerp(float* out, const float* low, const float* high, float val, int C, int H, int W, float* scratch, hipStream_t s);
int launch_weight_hadamard(float* out, const float* w, const float* a, const float* b, float scale, int64_t n, hipStream_t s);
int launch_weight_kron(float* out, const float* w, const float* w1, const float* w2, int r1, int c1, int r2, int c2, int k, float scale,
                       hipStream_t s);
int launch_weight_ia3(float* out, const float* w, const float* v, int rows, int cols, int on_input, float scale, hipStream_t s);
int launch_weight_dora(float* out, const float* w, const float* delta, const float* dora_scale, int rows, int cin, int k, float mult,
                       hipStream_t s);
int launch_cfg_prepare_concat(const float* x, const float* c_in, const float* cond, void* xin, int out_dtype, int B, int reps,
                              int C, int Cc, int64_t hw, unsigned zero_reps, hipStream_t s);
int launch_cfg_combine(const float* x, const float* eps, const float* c_out, float cond_scale, int mode,
                       const float* mask, const float* nmask, const float* init_latent, float* den, int B,
                       int64_t chw, hipStream_t s);
int launch_euler_step(float* x, const float* den, const float* noise, float sigma, float sigma_down, float sigma_up,
                      float s_noise, int64_t n, hipStream_t s);
int launch_dpmpp2m_step(float* x, const float* den, const float* old, float ratio, float em1, float c1, float c2,
                        int64_t n, hipStream_t s);
int launch_ddim_step(float* x, const float* e, const float* noise, float* pred_x0, float a_t, float a_prev,
                     float sigma_t, float somat, int64_t n, hipStream_t s);
int launch_axpby(float* y, const float* x, float a, const float* z, float b, int64_t n, hipStream_t s);
int launch_dpm_error(const float* lo, const float* hi, const float* prev, float atol, float rtol, float* partial256, int64_t n,
                     hipStream_t s);
int launch_lincomb(float* out, const float* const* terms, const float* coefs, int n_terms, int64_t n, hipStream_t s);
int launch_lora_merge(float* out, const void* w, int w_dtype, const void* up, int up_dtype, const void* down, int down_dtype,
                      int rows, int cols, int rank, float scale, hipStream_t s);
int launch_clip_embed(const int* tokens, const void* tok_emb, int tok_dtype, const float* pos_emb, const float* inputs_embeds,
                      half_t* out, int B, int L, int C, int vocab, hipStream_t s);
int launch_clip_pool(const int* tokens, const half_t* hidden, float* pooled, int B, int L, int C, hipStream_t s);
int launch_latent_resize(const float* in, float* out, int planes, int hi, int wi, int ho, int wo, int mode, hipStream_t s);
int launch_cfg_combine_affine(const float* x, const float* out, const float* c_out, const float* c_skip, float cond_scale,
                              const float* mask, const float* nmask, const float* init_latent, float* den, int B, int64_t chw,
                              hipStream_t s);
int launch_act_f16(half_t* x, int64_t n, int kind, hipStream_t s);                 // in place; kinds: hn_act in elementwise.hip
int launch_axpy_f16(half_t* y, const half_t* x, const half_t* h, float a, int64_t n, hipStream_t s);   // y = x + a * h
int launch_silu_f32(const float* x, float* y, int64_t n, hipStream_t s);
int launch_mask_blend(float* x, const float* init, const float* mask, const float* nmask, int64_t n, hipStream_t s);
int launch_image_to_u8(const float* img, uint8_t* out, int B, int C, int H, int W, hipStream_t s);

// NCHW (f16|f32) * scale -> NHWC fp16 with channels zero-padded to cpad; optional 1x1 channel mix (pqc) first.
int launch_nchw_to_nhwc(const void* x, int dtype, half_t* out, int B, int C, int HW, int cpad, float scale,
                        const float* mix_w, const float* mix_b, hipStream_t s, bool lo_ch = false);
// dst (NHWC fp16, optionally a (hi, lo) pair) = src + ctrl (NCHW, f16 | f32): ControlNet residuals into the UNet's skip connections
int launch_add_nchw_residual(const half_t* src, const half_t* src_lo, const void* ctrl, int dtype, half_t* dst, half_t* dst_lo, int B, int C,
                             int HW, hipStream_t s);
// fp32 NCHW -> user dtype NCHW copy
int launch_copy_out(const float* src, void* dst, int dtype, int64_t n, hipStream_t s);
int launch_convert_to_f16(const void* src, int dtype, half_t* dst, int64_t n, hipStream_t s);
int launch_convert_to_f32(const void* src, int dtype, float* dst, int64_t n, hipStream_t s);
// context cache validation on the device: *gate = 1 if any element of fp16(src[b][l][c]) differs from cached[b][l (stride Lpad)][c]
// (gate must be zeroed before), then — when *gate != 0 — the cached copy is overwritten with the new rows
int launch_ctx_compare(const void* src, int dtype, const half_t* cached, int B, int L, int Lpad, int C, int* gate, hipStream_t s);
int launch_ctx_update_gated(const void* src, int dtype, half_t* cached, int B, int L, int Lpad, int C, const int* gate, hipStream_t s);
// sinusoidal timestep embedding (cos first): t [B] (f16|f32) -> out fp32 [B, dim]
int launch_timestep_embedding(const void* t, int dtype, float* out, int B, int dim, hipStream_t s);
// out[b][n] = act_out( sum_k act_in(a[b][k]) * w[n][k] + bias[n] ) (+ add[b][n]); a fp32, w fp16, out fp32
int launch_small_linear(const float* a, const half_t* w, const float* bias, const float* add, float* out, int B,
                        int N, int K, int lda, int ldo, bool silu_in, bool silu_out, hipStream_t s);
// row softmax: in fp32 [rows, cols] -> out fp16 [rows, ldo] (zero-filled up to ldo)
int launch_softmax_rows(const float* in, half_t* out, int64_t rows, int cols, int ldi, int ldo, hipStream_t s);
// weight repack: OIHW (f16|f32) -> [O_pad][kh*kw][I_pad] fp16 (zero padded); geglu bit 0: permute output rows; bit 1: input channels
// [I, 2I) repeat [0, I) (the UNet's conv_in: launch_nchw_to_nhwc's lo channels)
int launch_pack_conv_weight(const void* w, int dtype, half_t* out, int O, int I, int kh, int kw, int O_pad,
                            int I_pad, int geglu, hipStream_t s);
// fp32 vector permute for GEGLU bias
int launch_pack_bias(const void* b, int dtype, float* out, int O, int O_pad, int geglu, hipStream_t s);

}  // namespace sdmi
