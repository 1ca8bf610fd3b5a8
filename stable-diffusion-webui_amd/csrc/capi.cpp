// capi.cpp — extern "C" surface of libsdmi.so (declared in include/sdmi.h).
#include "engine.h"
#include "prof.h"
#include <cstring>

namespace sdmi {
extern int g_vae_attn_rows;
int unet_forward(sdmi_engine* e, const void* x, const void* t, const void* ctx, const void* y, void* out, int io_dtype,
                 int Bn, int h, int w, int L, hipStream_t s);
int vae_decode(sdmi_engine* e, const void* z, int io_dtype, float* out, int B, int h, int w, hipStream_t s);
int vae_encode(sdmi_engine* e, const void* x, int io_dtype, float* out, int B, int H, int W, hipStream_t s);
int engine_load_unet_tensor(sdmi_engine* e, const char* key, const void* data, int dtype, int ndim, const int64_t* shape, int on_device);
int engine_load_vae_tensor(sdmi_engine* e, const char* key, const void* data, int dtype, int ndim, const int64_t* shape, int on_device);
int engine_unet_finalize(sdmi_engine* e);
int engine_clip_configure(sdmi_engine* e, int slot, const sdmi_clip_config* cfg);
int engine_clip_load_tensor(sdmi_engine* e, int slot, const char* key, const void* data, int dtype, int ndim, const int64_t* shape, int on_device);
int engine_clip_finalize(sdmi_engine* e, int slot);
int engine_clip_forward(sdmi_engine* e, int slot, const int* tokens, const float* inputs_embeds, int B, int L, int skip,
                        int apply_final_ln, float* out, float* pooled, hipStream_t s);
int engine_unet_update_weight(sdmi_engine* e, const char* key, const void* data, int dtype, int ndim, const int64_t* shape, int on_device);
int engine_vae_finalize(sdmi_engine* e);
int engine_hypernet_clear(sdmi_engine* e);
int engine_hypernet_begin(sdmi_engine* e, float multiplier);
int engine_hypernet_linear(sdmi_engine* e, int dim, int which, const void* w, const void* b, int dtype, int out_f, int in_f, int on_device);
int engine_hypernet_act(sdmi_engine* e, int dim, int which, int act);
int engine_hypernet_layernorm(sdmi_engine* e, int dim, int which, const void* g, const void* b, int dtype, int n, int on_device);
int engine_unet_update_vector(sdmi_engine* e, const char* key, const void* data, int dtype, int64_t n, int on_device);
int engine_set_context(sdmi_engine* e, const void* ctx, int dtype, int Bn, int L, hipStream_t s, bool conditional = false);
}  // namespace sdmi

using namespace sdmi;

#define API_GUARD_BEGIN try {
#define API_GUARD_END                                            \
    }                                                            \
    catch (const std::exception& ex) {                           \
        set_error(std::string("exception: ") + ex.what());      \
        return 1;                                                \
    }                                                            \
    catch (...) {                                                \
        set_error("unknown exception");                          \
        return 1;                                                \
    }

extern "C" {

int sdmi_version(void) { return SDMI_VERSION; }
const char* sdmi_last_error(void) { return get_error(); }

int sdmi_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
    return std::string(prop.gcnArchName).rfind("gfx950", 0) == 0 ? 1 : 0;
}

int64_t sdmi_attention_workspace_bytes(int B, int H, int M, int D) {
    const int64_t mpad = (M + 63) / 64 * 64;
    return (int64_t)B * H * D * mpad * (int64_t)sizeof(half_t);
}

int sdmi_attention_vt(const void* q, const void* k, const void* vt, void* out, int B, int H, int N, int M, int D, int ldq,
                      int ldk, int vt_ld, int ldo, float scale, int force_generic, void* stream) {
    API_GUARD_BEGIN
    AttnP p{};
    p.q = (const half_t*)q; p.k = (const half_t*)k; p.vt = (const half_t*)vt; p.out = (half_t*)out;
    p.B = B; p.H = H; p.N = N; p.M = M; p.D = D; p.ldq = ldq; p.ldk = ldk; p.vt_ld = vt_ld; p.ldo = ldo;
    p.scale_log2 = scale * 1.4426950408889634f;
    return launch_attention(p, force_generic != 0, (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_attention(const void* q, const void* k, const void* v, void* out, int B, int H, int N, int M, int D, int ldq,
                   int ldk, int ldv, int ldo, float scale, void* workspace, int64_t workspace_bytes, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(workspace && workspace_bytes >= sdmi_attention_workspace_bytes(B, H, M, D), "attention workspace too small");
    const int mpad = (M + 63) / 64 * 64;
    if (launch_transpose_v((const half_t*)v, (half_t*)workspace, B, H, M, D, ldv, mpad, (hipStream_t)stream)) return 1;
    return sdmi_attention_vt(q, k, workspace, out, B, H, N, M, D, ldq, ldk, mpad, ldo, scale, 0, stream);
    API_GUARD_END
}

// single-head attention over a wide head (the VAE AttnBlock: d = C = 512) with materialised scores, one image and one block of
// kWideRows query rows at a time, so the workspace — V^T for the batch plus ONE block's fp32 scores and fp16 probabilities — stays
// bounded (6 * 4096 * M bytes: 100 MB at a 1024^2 decode, 400 MB at 2048^2; the full N x M form was 1.6 GB / 25 GB) like
// the memory-bounded forwards it replaces (modules/sd_hijack_optimizations.py:390-424, 613-676) — the launches of engine.cpp run_vae_attn
static constexpr int kWideRows = 4096;
int64_t sdmi_attention_wide_workspace_bytes(int B, int N, int M, int D) {
    if (B <= 0 || N <= 0 || M <= 0 || D <= 0) return 0;
    const int64_t mpad = (M + 63) / 64 * 64;
    const int64_t rows = N < kWideRows ? N : kWideRows;
    return (int64_t)B * D * mpad * (int64_t)sizeof(half_t) + rows * mpad * (int64_t)(sizeof(float) + sizeof(half_t));
}

int sdmi_attention_wide(const void* q, const void* k, const void* v, void* out, int B, int N, int M, int D, int ldq, int ldk,
                        int ldv, int ldo, float scale, void* workspace, int64_t workspace_bytes, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(q && k && v && out && B > 0 && N > 0 && M > 0, "sdmi_attention_wide: null pointer or empty problem");
    SDMI_REQUIRE(D > 0 && D % 64 == 0 && D <= 1024, "sdmi_attention_wide: D must be a multiple of 64, at most 1024");
    SDMI_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 8 == 0, "sdmi_attention_wide: row strides must be multiples of 8 elements");
    SDMI_REQUIRE(ldq >= D && ldk >= D && ldv >= D && ldo >= D, "sdmi_attention_wide: row strides must be at least D");
    SDMI_REQUIRE(workspace && workspace_bytes >= sdmi_attention_wide_workspace_bytes(B, N, M, D), "attention workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int mpad = (M + 63) / 64 * 64;
    const int rows_max = N < kWideRows ? N : kWideRows;
    half_t* vt = (half_t*)workspace;
    float* S = (float*)(vt + (size_t)B * D * mpad);
    half_t* P = (half_t*)(S + (size_t)rows_max * mpad);
    if (launch_transpose_v((const half_t*)v, vt, B, 1, M, D, ldv, mpad, s)) return 1;
    for (int b = 0; b < B; ++b)
        for (int r0 = 0; r0 < N; r0 += kWideRows) {
            const int rows = N - r0 < kWideRows ? N - r0 : kWideRows;
            GemmP p{};
            p.a0 = (const half_t*)q + ((size_t)b * N + r0) * ldq; p.c0 = D; p.cin = D; p.lda0 = ldq;
            p.w = (const half_t*)k + (size_t)b * M * ldk; p.ldw = ldk;
            p.out = S;
            p.Hi = rows; p.Wi = 1; p.Ho = rows; p.Wo = 1; p.taps = 1; p.stride = 1;
            p.M = rows; p.N = mpad; p.n_valid = M; p.K = D; p.ldo = mpad; p.rows_per_batch = rows; p.n_real = mpad;
            p.flags = EP_OUT_F32;
            p.alpha = scale;
            if (launch_gemm(p, 1, false, true, s)) return 1;
            if (launch_softmax_rows(S, P, rows, M, mpad, mpad, s)) return 1;
            GemmP g{};
            g.a0 = P; g.c0 = mpad; g.cin = mpad; g.lda0 = mpad;
            g.w = vt + (size_t)b * D * mpad; g.ldw = mpad;
            g.out = (half_t*)out + ((size_t)b * N + r0) * ldo;
            g.Hi = rows; g.Wi = 1; g.Ho = rows; g.Wo = 1; g.taps = 1; g.stride = 1;
            g.M = rows; g.N = D; g.K = mpad; g.ldo = ldo; g.rows_per_batch = rows; g.n_real = D;
            g.alpha = 1.f;
            if (launch_gemm(g, 1, false, true, s)) return 1;
        }
    return 0;
    API_GUARD_END
}

static int desc_to_p(const sdmi_conv_desc* d, GemmP* p) {
    SDMI_REQUIRE(d != nullptr, "null descriptor");
    *p = GemmP{};
    p->a0 = (const half_t*)d->a0; p->a1 = (const half_t*)d->a1; p->w = (const half_t*)d->w;
    p->bias = (const float*)d->bias; p->rowbias = (const float*)d->rowbias; p->resid = (const half_t*)d->resid;
    p->out = d->out;
    p->c0 = d->c0; p->c1 = d->a1 ? d->c1 : 0; p->cin = p->c0 + p->c1;
    p->lda0 = d->lda0 ? d->lda0 : d->c0; p->lda1 = d->lda1 ? d->lda1 : d->c1;
    p->Hi = d->Hi; p->Wi = d->Wi; p->Ho = d->Ho; p->Wo = d->Wo;
    p->taps = d->taps; p->stride = d->stride ? d->stride : 1; p->pad = d->pad; p->up = d->up;
    SDMI_REQUIRE(p->taps == 1 || p->taps == 9, "taps must be 1 or 9");
    p->M = d->B * d->Ho * d->Wo; p->N = d->N; p->K = p->taps * p->cin;
    p->ldo = d->ldo; p->ldr = d->ldr; p->ldw = p->K; p->ldrb = d->N;
    p->rows_per_batch = d->Ho * d->Wo;
    p->n_real = d->n_real ? d->n_real : d->N;
    p->flags = d->flags;
    p->alpha = d->alpha == 0.f ? 1.f : d->alpha;
    p->a_bs = d->a_bs; p->w_bs = d->w_bs; p->o_bs = d->o_bs; p->r_bs = d->r_bs;
    const size_t need = gemm_splitk_ws_bytes(p->M, p->N, p->K, d->batch > 0 ? d->batch : 1);
    if (d->splitk_workspace && need && (size_t)d->splitk_workspace_bytes >= need) p->splitk_ws = (float*)d->splitk_workspace;
    return 0;
}

int64_t sdmi_conv_splitk_workspace_bytes(int M, int N, int K, int batch) { return (int64_t)gemm_splitk_ws_bytes(M, N, K, batch); }

int sdmi_conv_gemm(const sdmi_conv_desc* d, void* stream) {
    API_GUARD_BEGIN
    GemmP p;
    if (desc_to_p(d, &p)) return 1;
    const char* env = getenv("SDMI_NO_GLDS");
    return launch_gemm(p, d->batch > 0 ? d->batch : 1, d->force_generic == 1, !(env && env[0] == '1') && d->force_generic != 2,
                       (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_bench_conv_gemm(const sdmi_conv_desc* d, int iters, float* ms_out, void* stream) {
    API_GUARD_BEGIN
    GemmP p;
    if (desc_to_p(d, &p)) return 1;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    SDMI_CHECK_HIP(hipEventCreate(&e0));
    SDMI_CHECK_HIP(hipEventCreate(&e1));
    const bool glds = d->force_generic != 2;
    for (int i = 0; i < 2; ++i)
        if (launch_gemm(p, d->batch > 0 ? d->batch : 1, d->force_generic == 1, glds, s)) return 1;
    SDMI_CHECK_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i)
        if (launch_gemm(p, d->batch > 0 ? d->batch : 1, d->force_generic == 1, glds, s)) return 1;
    SDMI_CHECK_HIP(hipEventRecord(e1, s));
    SDMI_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    SDMI_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / (float)iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
    API_GUARD_END
}

int sdmi_pack_conv_weight(const void* w, int dtype, void* out, int O, int I, int kh, int kw, int O_pad, int I_pad, int geglu,
                          void* stream) {
    API_GUARD_BEGIN
    return launch_pack_conv_weight(w, dtype, (half_t*)out, O, I, kh, kw, O_pad, I_pad, geglu, (hipStream_t)stream);
    API_GUARD_END
}

int64_t sdmi_groupnorm_workspace_bytes(int B, int HW, int groups) { return groupnorm_ws_bytes(B, HW, groups); }

int sdmi_groupnorm(const void* x0, const void* x1, int c0, int c1, const void* gamma, const void* beta, void* out, int B, int HW,
                   int groups, float eps, int silu, void* ws, int64_t ws_bytes, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(ws && ws_bytes >= groupnorm_ws_bytes(B, HW, groups), "groupnorm workspace too small");
    return launch_groupnorm((const half_t*)x0, (const half_t*)x1, c0, x1 ? c1 : 0, (const float*)gamma, (const float*)beta,
                            (half_t*)out, B, HW, groups, eps, silu != 0, (float*)ws, (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_layernorm(const void* x, const void* gamma, const void* beta, void* out, int64_t rows, int C, float eps, void* stream) {
    API_GUARD_BEGIN
    return launch_layernorm((const half_t*)x, (const float*)gamma, (const float*)beta, (half_t*)out, rows, C, eps, (hipStream_t)stream);
    API_GUARD_END
}

int64_t sdmi_rowchain_ff_pack_bytes(int C, int hidden) { return (int64_t)rowchain_ff_pack_bytes(C, hidden); }
int sdmi_rowchain_ff_pack(const void* w1, const void* b1, const void* w2, void* packs, int C, int hidden, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(w1 && w2 && packs, "null argument");
    return launch_rowchain_ff_pack((const half_t*)w1, (const float*)b1, (const half_t*)w2, packs, C, hidden, false, (hipStream_t)stream);
    API_GUARD_END
}
int sdmi_rowchain_ff(const void* x, void* out, const void* g, const void* b, const void* packs, const void* b2, int64_t rows, int C,
                     int hidden, float eps, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(x && out && g && b && packs, "null argument");
    return launch_rowchain_ff((const half_t*)x, (half_t*)out, (const float*)g, (const float*)b, packs, (const float*)b2, (long)rows, C,
                              hidden, eps, (hipStream_t)stream);
    API_GUARD_END
}


int sdmi_philox_randn(void* out, int64_t n, uint64_t seed, uint32_t offset, void* stream) {
    API_GUARD_BEGIN
    return launch_philox((float*)out, n, seed, offset, (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_slerp(void* out, const void* low, const void* high, float val, int C, int H, int W, void* scratch, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(out && low && high && scratch && C > 0 && H > 0 && W > 0, "bad arguments");
    return launch_slerp((float*)out, (const float*)low, (const float*)high, val, C, H, W, (float*)scratch, (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_cfg_prepare_input(const void* x, const void* c_in, void* x_in, int out_dtype, int B, int reps, int64_t chw, void* stream) {
    API_GUARD_BEGIN
    return launch_cfg_prepare((const float*)x, (const float*)c_in, x_in, out_dtype, B, reps, chw, (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_cfg_prepare_concat(const void* x, const void* c_in, const void* c_concat, void* x_in, int out_dtype, int B, int reps,
                            int C, int Cc, int64_t hw, uint32_t zero_reps, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(x && c_concat && x_in, "null pointer");
    SDMI_REQUIRE(B > 0 && reps > 0 && reps <= 32 && C > 0 && Cc > 0 && hw > 0, "bad sizes");
    return launch_cfg_prepare_concat((const float*)x, (const float*)c_in, (const float*)c_concat, x_in, out_dtype, B, reps, C, Cc, hw,
                                     zero_reps, (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_cfg_combine(const void* x, const void* eps, const void* c_out, float cond_scale, int mode, const void* mask,
                     const void* nmask, const void* init_latent, void* den, int B, int64_t chw, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(mode == 1 || (x && c_out), "sigma-space combine needs x and c_out");
    return launch_cfg_combine((const float*)x, (const float*)eps, (const float*)c_out, cond_scale, mode, (const float*)mask,
                              (const float*)nmask, (const float*)init_latent, (float*)den, B, chw, (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_euler_step(void* x, const void* den, const void* noise, float sigma, float sigma_down, float sigma_up, float s_noise,
                    int64_t n, void* stream) {
    API_GUARD_BEGIN
    return launch_euler_step((float*)x, (const float*)den, (const float*)noise, sigma, sigma_down, sigma_up, s_noise, n,
                             (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_dpmpp2m_step(void* x, const void* den, const void* old, float ratio, float em1, float c1, float c2, int64_t n,
                      void* stream) {
    API_GUARD_BEGIN
    return launch_dpmpp2m_step((float*)x, (const float*)den, (const float*)old, ratio, em1, c1, c2, n, (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_ddim_step(void* x, const void* e_t, const void* noise, void* pred_x0, float a_t, float a_prev, float sigma_t,
                   float sqrt_one_minus_at, int64_t n, void* stream) {
    API_GUARD_BEGIN
    return launch_ddim_step((float*)x, (const float*)e_t, (const float*)noise, (float*)pred_x0, a_t, a_prev, sigma_t,
                            sqrt_one_minus_at, n, (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_dpm_error_partials(const void* x_low, const void* x_high, const void* x_prev, float atol, float rtol, void* partial256_f32,
                            int64_t n, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(x_low && x_high && x_prev && partial256_f32 && n > 0, "null argument");
    return launch_dpm_error((const float*)x_low, (const float*)x_high, (const float*)x_prev, atol, rtol, (float*)partial256_f32, n,
                            (hipStream_t)stream);
    API_GUARD_END
}
int sdmi_lincomb(void* out, const void* const* terms, const float* coefs, int n_terms, int64_t n, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(out && terms && coefs, "null argument");
    return launch_lincomb((float*)out, (const float* const*)terms, coefs, n_terms, n, (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_latent_resize(const void* in, void* out, int planes, int hi, int wi, int ho, int wo, int mode, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(in && out, "null argument");
    return launch_latent_resize((const float*)in, (float*)out, planes, hi, wi, ho, wo, mode, (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_cfg_combine_affine(const void* x, const void* out, const void* c_out, const void* c_skip, float cond_scale,
                            const void* mask, const void* nmask, const void* init_latent, void* den, int B, int64_t chw, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(x && out && c_out && c_skip && den, "null argument");
    return launch_cfg_combine_affine((const float*)x, (const float*)out, (const float*)c_out, (const float*)c_skip, cond_scale,
                                     (const float*)mask, (const float*)nmask, (const float*)init_latent, (float*)den, B, chw,
                                     (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_mask_blend(void* x, const void* init, const void* mask, const void* nmask, int64_t n, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(x && init && mask && nmask, "null argument");
    return launch_mask_blend((float*)x, (const float*)init, (const float*)mask, (const float*)nmask, n, (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_axpby(void* y, const void* x, float a, const void* z, float b, int64_t n, void* stream) {
    API_GUARD_BEGIN
    return launch_axpby((float*)y, (const float*)x, a, (const float*)z, b, n, (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_image_to_u8(const void* img, void* out, int B, int C, int H, int W, void* stream) {
    API_GUARD_BEGIN
    return launch_image_to_u8((const float*)img, (uint8_t*)out, B, C, H, W, (hipStream_t)stream);
    API_GUARD_END
}

sdmi_engine* sdmi_engine_create(int device) {
    try {
        if (hipSetDevice(device) != hipSuccess) {
            set_error("hipSetDevice failed");
            return nullptr;
        }
        sdmi_engine* e = new sdmi_engine();
        e->device = device;
        const char* g = getenv("SDMI_NO_GLDS");
        if (g && g[0] == '1') e->use_glds = false;
        const char* f = getenv("SDMI_FORCE_GENERIC");
        if (f && f[0] == '1') e->force_generic = true;
        return e;
    } catch (...) {
        set_error("engine allocation failed");
        return nullptr;
    }
}
void sdmi_engine_destroy(sdmi_engine* e) { delete e; }

int sdmi_unet_configure(sdmi_engine* e, const sdmi_unet_config* cfg) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(e && cfg, "null argument");
    SDMI_REQUIRE(cfg->num_levels >= 1 && cfg->num_levels <= 8, "num_levels out of range");
    SDMI_REQUIRE(cfg->model_channels % 64 == 0, "model_channels must be a multiple of 64");
    SDMI_REQUIRE(cfg->context_dim % 64 == 0, "context_dim must be a multiple of 64");
    SDMI_REQUIRE(cfg->in_channels <= 16 && cfg->out_channels <= 64, "in_channels <= 16, out_channels <= 64");
    e->unet.cfg = *cfg;
    e->unet.ready = false;
    return 0;
    API_GUARD_END
}
int sdmi_unet_load_tensor(sdmi_engine* e, const char* key, const void* data, int dtype, int ndim, const int64_t* shape,
                          int on_device) {
    API_GUARD_BEGIN
    return engine_load_unet_tensor(e, key, data, dtype, ndim, shape, on_device);
    API_GUARD_END
}
int sdmi_unet_finalize(sdmi_engine* e) {
    API_GUARD_BEGIN
    return engine_unet_finalize(e);
    API_GUARD_END
}
int sdmi_unet_update_weight(sdmi_engine* e, const char* key, const void* data, int dtype, int ndim, const int64_t* shape,
                            int on_device) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(e != nullptr, "null engine");
    return engine_unet_update_weight(e, key, data, dtype, ndim, shape, on_device);
    API_GUARD_END
}
int sdmi_lora_merge(void* out, const void* w, int w_dtype, const void* up, int up_dtype, const void* down, int down_dtype,
                    int rows, int cols, int rank, float scale, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(out && w && up && down && rows > 0 && cols > 0 && rank > 0, "bad lora_merge arguments");
    return launch_lora_merge((float*)out, w, w_dtype, up, up_dtype, down, down_dtype, rows, cols, rank, scale, (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_weight_hadamard(void* out, const void* w, const void* a, const void* b, float scale, int64_t n, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(out && w && a && b && n > 0, "bad arguments");
    return launch_weight_hadamard((float*)out, (const float*)w, (const float*)a, (const float*)b, scale, n, (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_weight_kron(void* out, const void* w, const void* w1, const void* w2, int r1, int c1, int r2, int c2, int k, float scale,
                     void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(out && w && w1 && w2 && r1 > 0 && c1 > 0 && r2 > 0 && c2 > 0 && k > 0, "bad arguments");
    return launch_weight_kron((float*)out, (const float*)w, (const float*)w1, (const float*)w2, r1, c1, r2, c2, k, scale, (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_weight_ia3(void* out, const void* w, const void* v, int rows, int cols, int on_input, float scale, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(out && w && v && rows > 0 && cols > 0, "bad arguments");
    return launch_weight_ia3((float*)out, (const float*)w, (const float*)v, rows, cols, on_input, scale, (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_weight_dora(void* out, const void* w, const void* delta, const void* dora_scale, int rows, int cin, int k, float mult,
                     void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(out && w && delta && dora_scale && rows > 0 && cin > 0 && k > 0, "bad arguments");
    return launch_weight_dora((float*)out, (const float*)w, (const float*)delta, (const float*)dora_scale, rows, cin, k, mult,
                              (hipStream_t)stream);
    API_GUARD_END
}
int sdmi_clip_configure(sdmi_engine* e, int slot, const sdmi_clip_config* cfg) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(e && cfg, "null argument");
    return engine_clip_configure(e, slot, cfg);
    API_GUARD_END
}
int sdmi_clip_load_tensor(sdmi_engine* e, int slot, const char* key, const void* data, int dtype, int ndim, const int64_t* shape,
                          int on_device) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(e != nullptr, "null engine");
    return engine_clip_load_tensor(e, slot, key, data, dtype, ndim, shape, on_device);
    API_GUARD_END
}
int sdmi_clip_finalize(sdmi_engine* e, int slot) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(e != nullptr, "null engine");
    return engine_clip_finalize(e, slot);
    API_GUARD_END
}
int sdmi_clip_forward(sdmi_engine* e, int slot, const void* tokens, const void* inputs_embeds, int B, int L, int skip,
                      int apply_final_ln, void* out, void* pooled, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(e != nullptr, "null engine");
    return engine_clip_forward(e, slot, (const int*)tokens, (const float*)inputs_embeds, B, L, skip, apply_final_ln, (float*)out,
                               (float*)pooled, (hipStream_t)stream);
    API_GUARD_END
}
int sdmi_unet_hypernet_clear(sdmi_engine* e) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(e != nullptr, "null engine");
    return engine_hypernet_clear(e);
    API_GUARD_END
}
int sdmi_unet_hypernet_begin(sdmi_engine* e, float multiplier) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(e != nullptr, "null engine");
    return engine_hypernet_begin(e, multiplier);
    API_GUARD_END
}
int sdmi_unet_hypernet_linear(sdmi_engine* e, int dim, int which, const void* w, const void* b, int dtype, int out_features, int in_features,
                              int on_device) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(e != nullptr, "null engine");
    return engine_hypernet_linear(e, dim, which, w, b, dtype, out_features, in_features, on_device);
    API_GUARD_END
}
int sdmi_unet_hypernet_act(sdmi_engine* e, int dim, int which, int act) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(e != nullptr, "null engine");
    return engine_hypernet_act(e, dim, which, act);
    API_GUARD_END
}
int sdmi_unet_hypernet_layernorm(sdmi_engine* e, int dim, int which, const void* gamma, const void* beta, int dtype, int n, int on_device) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(e != nullptr, "null engine");
    return engine_hypernet_layernorm(e, dim, which, gamma, beta, dtype, n, on_device);
    API_GUARD_END
}
int sdmi_unet_update_vector(sdmi_engine* e, const char* key, const void* data, int dtype, int64_t n, int on_device) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(e != nullptr, "null engine");
    return engine_unet_update_vector(e, key, data, dtype, n, on_device);
    API_GUARD_END
}
int sdmi_vae_configure(sdmi_engine* e, const sdmi_vae_config* cfg) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(e && cfg, "null argument");
    SDMI_REQUIRE(cfg->num_levels >= 1 && cfg->num_levels <= 8 && cfg->ch % 64 == 0, "vae: ch % 64 == 0, 1..8 levels");
    SDMI_REQUIRE(cfg->z_channels <= 16 && cfg->in_channels <= 16, "vae: z_channels, in_channels <= 16");
    e->vae.cfg = *cfg;
    e->vae.ready = false;
    return 0;
    API_GUARD_END
}
int sdmi_vae_load_tensor(sdmi_engine* e, const char* key, const void* data, int dtype, int ndim, const int64_t* shape,
                         int on_device) {
    API_GUARD_BEGIN
    return engine_load_vae_tensor(e, key, data, dtype, ndim, shape, on_device);
    API_GUARD_END
}
int sdmi_vae_finalize(sdmi_engine* e) {
    API_GUARD_BEGIN
    return engine_vae_finalize(e);
    API_GUARD_END
}

int sdmi_unet_set_context(sdmi_engine* e, const void* context, int io_dtype, int Bn, int L, void* stream) {
    API_GUARD_BEGIN
    return engine_set_context(e, context, io_dtype, Bn, L, (hipStream_t)stream);
    API_GUARD_END
}
int sdmi_unet_set_context_cached(sdmi_engine* e, const void* context, int io_dtype, int Bn, int L, void* stream) {
    API_GUARD_BEGIN
    return engine_set_context(e, context, io_dtype, Bn, L, (hipStream_t)stream, true);
    API_GUARD_END
}
int sdmi_unet_forward(sdmi_engine* e, const void* x, const void* timesteps, const void* context, const void* y, void* out,
                      int io_dtype, int Bn, int h, int w, int L, void* stream) {
    API_GUARD_BEGIN
    return unet_forward(e, x, timesteps, context, y, out, io_dtype, Bn, h, w, L, (hipStream_t)stream);
    API_GUARD_END
}
int sdmi_unet_set_control(sdmi_engine* e, const void* const* tensors, const int64_t* numel, int n, int only_mid_control) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(e != nullptr && n >= 0 && (n == 0 || (tensors && numel)), "null argument");
    e->control.assign(tensors, tensors + n);
    e->control_numel.assign(numel, numel + n);
    e->only_mid_control = only_mid_control != 0;
    for (int i = 0; i < n; ++i) SDMI_REQUIRE(tensors[i] != nullptr, "null control tensor");
    return 0;
    API_GUARD_END
}
int sdmi_unet_forward_ex(sdmi_engine* e, const void* x, const void* timesteps, const void* context, const void* y, void* out,
                         int io_dtype, int Bn, int h, int w, int L, int call_flags, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(e != nullptr, "null engine");
    // the promises hold for this call only: whatever the (legacy) sticky options say is put back afterwards
    struct Restore {
        sdmi_engine* e; bool uni, pairs, derive;
        ~Restore() { e->uniform_t = uni; e->cfg_pairs = pairs; e->auto_promises = derive; }
    } restore{e, e->uniform_t, e->cfg_pairs, e->auto_promises};
    e->uniform_t = (call_flags & SDMI_CALL_UNIFORM_T) != 0;
    e->cfg_pairs = (call_flags & SDMI_CALL_CFG_PAIRS) != 0;
    e->auto_promises = (call_flags & SDMI_CALL_DERIVE) != 0;
    return unet_forward(e, x, timesteps, context, y, out, io_dtype, Bn, h, w, L, (hipStream_t)stream);
    API_GUARD_END
}
int sdmi_vae_decode(sdmi_engine* e, const void* z, int io_dtype, void* out, int B, int h, int w, void* stream) {
    API_GUARD_BEGIN
    return vae_decode(e, z, io_dtype, (float*)out, B, h, w, (hipStream_t)stream);
    API_GUARD_END
}
int sdmi_vae_encode(sdmi_engine* e, const void* x, int io_dtype, void* out, int B, int H, int W, void* stream) {
    API_GUARD_BEGIN
    return vae_encode(e, x, io_dtype, (float*)out, B, H, W, (hipStream_t)stream);
    API_GUARD_END
}

int sdmi_debug_set(const char* name, int value) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(name != nullptr, "null name");
    const std::string n(name);
    if (n == "gemm_cfg") g_force_gemm_cfg = value;
    else if (n == "gemm_shortk_cfg") g_shortk_gemm_cfg = value;
    else if (n == "vae_attn_rows") g_vae_attn_rows = value;
    else if (n == "gemm_shortk_maxk") g_shortk_max_k = value;
    else if (n == "gemm_geglu_cfg") g_geglu_gemm_cfg = value;
    else if (n == "vt_mode") g_vt_mode = value;
    else if (n == "tile_order") g_tile_order = value;
    else if (n == "conv_korder") g_conv_korder = value < 0 ? g_conv_korder_default : value;
    else if (n == "small_linear_lds") g_small_linear_lds = value;
    else if (n == "gn_fuse") g_gn_fuse = value;
    else if (n == "gn_small") g_gn_small = value;
    else if (n == "ep_wide") g_ep_wide = value;
    else if (n == "attn_kvt") g_attn_kvt = value;
    else if (n == "attn_occ") g_attn_occ = value;
    else if (n == "attn_lds_pad") g_attn_lds_pad = value;
    else if (n == "attn_fold_min_m") g_attn_fold_min_m = value;
    else if (n == "attn_tau") g_attn_tau = value;
    else if (n == "attn_dbg_lo") g_attn_dbg = (g_attn_dbg & 0xFFFFFFFF00000000ull) | (unsigned)value;
    else if (n == "attn_dbg_hi") g_attn_dbg = (g_attn_dbg & 0xFFFFFFFFull) | ((unsigned long long)(unsigned)value << 32);
    else if (n == "gemm_split") g_force_gemm_split = value;
    else if (n == "gemm_pipe") g_gemm_pipe = value < 0 ? g_gemm_pipe_default : value;
    else if (n == "gemm_lin") g_gemm_lin = value < 0 ? 1 : value;
    else if (n == "gn_apply_blocks") g_gn_apply_blocks = value < 0 ? 0 : value;
    else if (n == "gn_band_elems") g_gn_band_elems = value < 0 ? 0 : value;
    else if (n == "gemm_dbgflags") g_gemm_dbgflags = value & 0x1F00;
    else if (n == "gemm_dbg_lo") g_gemm_dbg = (g_gemm_dbg & 0xFFFFFFFF00000000ull) | (unsigned)value;
    else if (n == "gemm_dbg_hi") g_gemm_dbg = (g_gemm_dbg & 0xFFFFFFFFull) | ((unsigned long long)(unsigned)value << 32);
    else { set_error("unknown debug knob " + n); return 1; }
    return 0;
    API_GUARD_END
}

int sdmi_debug_set_str(const char* name, const char* value) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(name != nullptr, "null name");
    if (std::string(name) == "gemm_override") {
        SDMI_REQUIRE(gemm_set_override(value) == 0, "gemm_override: expected M,N,K,taps,kind:cfg:split;...");
        return 0;
    }
    set_error(std::string("unknown debug knob ") + name);
    return 1;
    API_GUARD_END
}

int sdmi_profile_begin(void) {
    prof_begin();
    return 0;
}
int sdmi_profile_end(char* json_out, int capacity) {
    const std::string s = prof_end();
    if (!json_out || capacity <= (int)s.size()) {
        set_error("profile buffer too small");
        return 1;
    }
    std::memcpy(json_out, s.c_str(), s.size() + 1);
    return 0;
}

int64_t sdmi_engine_arena_bytes(sdmi_engine* e) { return e ? (int64_t)e->arena.cap : 0; }

int sdmi_engine_set_option(sdmi_engine* e, const char* name, int value) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(e && name, "null argument");
    const std::string n(name);
    if (n == "force_generic") e->force_generic = value != 0;
    else if (n == "glds") e->use_glds = value != 0;
    else if (n == "trace") { e->trace = value != 0; e->taps.clear(); }
    else if (n == "tiling") e->tiling = value != 0;
    else if (n == "uniform_t") e->uniform_t = value != 0;
    else if (n == "cfg_pairs") e->cfg_pairs = value != 0;
    else if (n == "auto_promises") e->auto_promises = value != 0;
    else if (n == "ln_fold") e->ln_fold = value;
    else if (n == "fuse_rows") e->fuse_rows = value;
    else if (n == "residual_fp32") e->residual_fp32 = value != 0;
    else if (n == "arena_reuse") e->arena_reuse = value;
    else if (n == "streams") { if (value < 1 || value > 8) { sdmi::set_error("streams must be 1..8"); return 1; } e->n_streams = value; }
    else if (n == "vae_range_extend") e->vae_stream_scale = value ? 1.0f / 64.0f : 1.0f;
    else { set_error("unknown option " + n); return 1; }
    return 0;
    API_GUARD_END
}

int sdmi_engine_tap_count(sdmi_engine* e) { return e ? (int)e->taps.size() : 0; }
int sdmi_engine_tap_info(sdmi_engine* e, int index, char* name_out, int capacity, int64_t* dims_bhwc) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(e && name_out && dims_bhwc && index >= 0 && index < (int)e->taps.size(), "bad tap index / null argument");
    const auto& t = e->taps[index];
    SDMI_REQUIRE((int)t.name.size() < capacity, "tap name buffer too small");
    std::memcpy(name_out, t.name.c_str(), t.name.size() + 1);
    dims_bhwc[0] = t.B; dims_bhwc[1] = t.H; dims_bhwc[2] = t.W; dims_bhwc[3] = t.C;
    return 0;
    API_GUARD_END
}
int sdmi_engine_tap_read(sdmi_engine* e, int index, void* out_f16_nhwc, void* stream) {
    API_GUARD_BEGIN
    SDMI_REQUIRE(e && out_f16_nhwc && index >= 0 && index < (int)e->taps.size(), "bad tap index / null argument");
    const auto& t = e->taps[index];
    SDMI_CHECK_HIP(hipMemcpyAsync(out_f16_nhwc, t.ptr, (size_t)t.B * t.H * t.W * t.C * sizeof(half_t), hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream));
    return 0;
    API_GUARD_END
}

}  // extern "C"
