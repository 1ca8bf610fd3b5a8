/* sdmi.h — C ABI of the MI355X-native Stable Diffusion hot-path engine (libsdmi.so).
 *
 * Drop-in boundary for ONE path of AUTOMATIC1111/stable-diffusion-webui: the txt2img/img2img UNet denoising
 * loop + k-diffusion/DDIM sampler arithmetic + VAE decode.  The reference has no FFI of its
 * own (it is pure Python over torch); each entry point below states the reference interface it replaces
 * (paths relative to /root/reference).  The Python/ctypes binding a webui maintainer adds is in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the engine's GPU unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous on it;
 *   - return value: 0 = ok, non-zero = error; sdmi_last_error() returns a thread-local message;
 *   - fp16 = IEEE binary16 ("half"); all accumulation is fp32; activations inside the engine are NHWC fp16;
 *   - no torch / python types appear here.
 */
#ifndef SDMI_H
#define SDMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDMI_VERSION 100

enum { SDMI_F16 = 0, SDMI_F32 = 1 };

typedef struct sdmi_engine sdmi_engine;

int sdmi_version(void);
const char* sdmi_last_error(void);
/* 1 if a gfx950 device is visible to HIP, else 0 (never throws). */
int sdmi_device_ok(void);

/* ------------------------------------------------------------------------------------------------------------
 * Op-level entry points (boundary B2 "SdOptimization", and the units the parity tests exercise one by one).
 * ---------------------------------------------------------------------------------------------------------- */

/* Fused QK^T * scale -> softmax -> PV, never materialising the score matrix.
 * Replaces: the attention math inside every CrossAttention.forward the webui can select —
 *   modules/sd_hijack_optimizations.py:221-281 (split_cross_attention_forward, GPU default),
 *   :508-546 (sdp), :480-503 (xformers), modules/hypernetworks/hypernetwork.py:382-407 (baseline),
 *   algorithm twin modules/sub_quadratic_attention.py:141-215.
 * q   [B, N, ldq]  head h at columns [h*D, (h+1)*D)            (fp16)
 * k   [B, M, ldk]  same head layout                             (fp16)
 * v   [B, M, ldv]  same head layout (row-major; transposed internally into `workspace`)
 * out [B, N, ldo]                                                (fp16)
 * workspace: at least sdmi_attention_workspace_bytes(B,H,M,D) bytes of device memory.
 * D in {40, 64, 80, 128, 160} runs the MFMA kernel; any other D <= 512 runs the generic HIP kernel. */
int64_t sdmi_attention_workspace_bytes(int B, int H, int M, int D);
int sdmi_attention(const void* q, const void* k, const void* v, void* out,
                   int B, int H, int N, int M, int D,
                   int ldq, int ldk, int ldv, int ldo, float scale,
                   void* workspace, int64_t workspace_bytes, void* stream);

/* Single-head attention over a wide head: the VAE mid-block AttnBlock (d = C = 512, N = M = h*w) that every in-tree optimizer
 * also replaces — modules/sd_hijack_optimizations.py:554-610 (cross_attention_attnblock_forward), :613-634 (xformers), :637-655
 * (sdp), :658-676 (sub-quadratic).  Scores are materialised in fp32 one image at a time (GEMM, row softmax, GEMM: the engine's own
 * VAE attention).  q / k / v / out [B, N|M, ld*] fp16, D a multiple of 64, row strides multiples of 8 elements. */
int64_t sdmi_attention_wide_workspace_bytes(int B, int N, int M, int D);
int sdmi_attention_wide(const void* q, const void* k, const void* v, void* out,
                        int B, int N, int M, int D,
                        int ldq, int ldk, int ldv, int ldo, float scale,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* Same, with V already transposed by the producer: vt [B, H*D, Mpad] (row = h*D+d, Mpad = vt_ld >= M rounded up to 64,
 * padding columns must be finite).  This is what the engine's UNet uses (its V projection writes V^T directly). */
int sdmi_attention_vt(const void* q, const void* k, const void* vt, void* out,
                      int B, int H, int N, int M, int D,
                      int ldq, int ldk, int vt_ld, int ldo, float scale, int force_generic, void* stream);

/* Implicit-GEMM convolution / linear layer on NHWC fp16 activations (MFMA, LDS-staged):
 *   out[m, n] = epilogue( alpha * sum_{tap,c} A(m, tap, c) * W[n, tap*Cin + c] )
 * Replaces the torch ops issued by ldm's ResBlock / Downsample / Upsample / SpatialTransformer / FeedForward modules
 * (third-party; layer names pinned at extensions-builtin/Lora/networks.py:43-98; op inventory SURVEY.md 2.3 K1,K4,K7,K8).
 * All fields are plain values / device pointers. */
typedef struct sdmi_conv_desc {
    const void* a0;        /* source 0, NHWC fp16 [B,Hi,Wi,c0] (pixel stride lda0 elements) */
    const void* a1;        /* optional source 1 (channel-concatenated after source 0), or NULL */
    const void* w;         /* packed weights fp16 [N][taps*(c0+c1)], k = tap*Cin + c, tap = ky*3+kx */
    const void* bias;      /* fp32 [N] (or [M] when SDMI_EP_BIAS_ROW), or NULL */
    const void* rowbias;   /* fp32 [B][N] added to every pixel of image b (ResBlock emb add), or NULL */
    const void* resid;     /* fp16 [M][ldr] added before the store, or NULL */
    void* out;             /* fp16 or fp32, row-major [M][ldo] or NCHW fp32 */
    int32_t c0, c1, lda0, lda1;
    int32_t B, Hi, Wi, Ho, Wo;
    int32_t taps;          /* 1 or 9 */
    int32_t stride;        /* 1 or 2 */
    int32_t pad;           /* 1: symmetric "padding=1"; 0: none (VAE-encoder downsample pads right/bottom only) */
    int32_t up;            /* 1: nearest x2 upsample of the source fused into the gather */
    int32_t N;             /* output channels as packed (multiple of 64 for the MFMA kernel) */
    int32_t n_real;        /* output channels actually stored (NCHW mode), else = N */
    int32_t ldo, ldr;
    int32_t flags;         /* SDMI_EP_* */
    float alpha;
    int32_t batch;         /* grid.z batched GEMM count (>=1) with the strides below (elements) */
    int64_t a_bs, w_bs, o_bs, r_bs;
    int32_t force_generic; /* 1: run the simple non-MFMA HIP kernel (debug / unsupported shapes) */
    int32_t reserved;
    void* splitk_workspace;          /* optional fp32 scratch enabling deterministic split-K on small-M / large-K shapes */
    int64_t splitk_workspace_bytes;  /* >= sdmi_conv_splitk_workspace_bytes(M, N, K, batch) or the workspace is ignored */
} sdmi_conv_desc;

enum {
    SDMI_EP_OUT_F32  = 1,   /* store fp32 instead of fp16 */
    SDMI_EP_GEGLU    = 2,   /* W rows packed in (32 value | 32 gate) groups; out has N/2 columns: a*gelu(g) */
    SDMI_EP_NCHW     = 4,   /* store fp32 NCHW [B][n_real][Ho][Wo] */
    SDMI_EP_BIAS_ROW = 8,   /* bias indexed by output row m instead of column n */
    SDMI_EP_WRAP = 128,     /* 3x3 taps wrap around the image (Conv2d padding_mode = 'circular') instead of reading zero padding */
    SDMI_EP_TRANSPOSE = 64  /* store out^T per image: out[b][n][m - b*Ho*Wo], row stride ldo (fp16; Ho*Wo % 4 == 0; no residual /
                               rowbias): how the V projection is handed to the attention kernel as V^T [C][tokens] */
};
int sdmi_conv_gemm(const sdmi_conv_desc* d, void* stream);
int64_t sdmi_conv_splitk_workspace_bytes(int M, int N, int K, int batch);

/* Repack helpers (device side, run once at load): OIHW fp16/fp32 conv weight -> [O][ky*3+kx][I(+pad)] fp16. */
int sdmi_pack_conv_weight(const void* w_oihw, int dtype, void* out_f16, int O, int I, int kh, int kw,
                          int O_pad, int I_pad, int geglu, void* stream);

/* GroupNorm(32 groups, fp32 statistics) [+ SiLU] over NHWC fp16, optionally reading two channel-concatenated sources
 * and writing one tensor.  Replaces ldm GroupNorm32 + SiLU (fp32 contract: modules/devices.py:284-295; SiLU forced at
 * modules/sd_hijack.py:69) and the torch.cat of skip connections (modules/sd_hijack_unet.py:10-33). */
int sdmi_groupnorm(const void* x0, const void* x1, int c0, int c1, const void* gamma_f32, const void* beta_f32,
                   void* out_f16, int B, int HW, int groups, float eps, int silu,
                   void* workspace_f32, int64_t workspace_bytes, void* stream);
int64_t sdmi_groupnorm_workspace_bytes(int B, int HW, int groups);

/* LayerNorm over the last dim of [rows, C] fp16 (fp32 statistics, eps 1e-5). */
int sdmi_layernorm(const void* x, const void* gamma_f32, const void* beta_f32, void* out_f16,
                   int64_t rows, int C, float eps, void* stream);

/* The feed-forward chain of a BasicTransformerBlock as one launch (csrc/rowchain.hip; row width C = 320, rows % 128 == 0).  It
 * replaces, for the ff third of ldm's BasicTransformerBlock._forward as the webui runs it (modules/sd_hijack_unet.py:83-102), the launch
 * sequence LayerNorm -> GEGLU proj -> Linear -> + x.  (Round 5 also exported the cross-attention chain, sdmi_rowchain_xattn*: measured
 * slower than the launches it replaced and removed in round 6 — VERDICT r5 item 7.)
 *   sdmi_rowchain_ff_pack     w1 [2*hidden][C] fp16 (torch order: value rows, then gate rows), b1 [2*hidden] fp32 or null,
 *                             w2 [C][hidden] fp16 -> the packed operand stream (sdmi_rowchain_ff_pack_bytes)
 *   sdmi_rowchain_ff          out = x + (w2 GEGLU(w1 LN(x) + b1) + b2) */
int64_t sdmi_rowchain_ff_pack_bytes(int C, int hidden);
int sdmi_rowchain_ff_pack(const void* w1_f16, const void* b1_f32_or_null, const void* w2_f16, void* packs, int C, int hidden,
                          void* stream);
int sdmi_rowchain_ff(const void* x_f16, void* out_f16, const void* ln_gamma_f32, const void* ln_beta_f32, const void* packs,
                     const void* b2_f32_or_null, int64_t rows, int C, int hidden, float eps, void* stream);


/* Philox4x32-10 + Box-Muller normal draws, bit-compatible with the reference's "NV" noise source
 * (modules/rng_philox.py:32-102): out[i] = randn(counter=[offset,0,i,0], key=seed). */
int sdmi_philox_randn(void* out_f32, int64_t n, uint64_t seed, uint32_t offset, void* stream);

/* Variation seeds: slerp(val, low, high) of modules/rng.py:85-96 on ONE image's noise tensors [C][H][W] (fp32) — the angle per
 * (c, w) column along H, linear fallback low*val + high*(1-val) when the mean cosine exceeds 0.9995 (sic: the reference's weights).
 * scratch: C*W floats.  Called from ImageRNG.first (modules/rng.py:120-127) when subseed_strength != 0. */
int sdmi_slerp(void* out_f32, const void* low_f32, const void* high_f32, float val, int C, int H, int W, void* scratch_f32,
               void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Sampler arithmetic (boundary B3): the elementwise work of CFGDenoiser + CompVisDenoiser + the k-diffusion /
 * DDIM update, fused.  x is the fp32 sampler state [B,4,h,w] (NCHW, as the webui keeps it).
 * ---------------------------------------------------------------------------------------------------------- */

/* x_in[r*B + b] = x[b] * c_in[b]  for r in 0..reps-1, written NCHW [reps*B, C, h, w] in out_dtype (SDMI_F16/F32).
 * Replaces modules/sd_samplers_cfg_denoiser.py:203-205 (x_in = cat(cond copies, uncond)) + k-diffusion
 * CompVisDenoiser's `input * c_in` + the cast to dtype_unet (modules/sd_hijack_unet.py:50). c_in_f32 may be NULL (=1). */
int sdmi_cfg_prepare_input(const void* x_f32, const void* c_in_f32, void* x_in, int out_dtype,
                           int B, int reps, int64_t chw, void* stream);

/* The same for checkpoints whose UNet input is cat([x, c_concat], dim=1) — inpainting (9 channels: x | mask | masked-image
 * latent) and InstructPix2Pix (8: x | image latent); ldm's DiffusionWrapper does the cat for conditioning_key "hybrid" /
 * "concat", fed by make_condition_dict at modules/sd_samplers_cfg_denoiser.py:193-209:
 *   x_in[r*B + b, 0:C] = x[b] * c_in[b],   x_in[r*B + b, C:C+Cc] = c_concat[b]   (zeros when bit r of zero_reps is set:
 *   the third, image-unconditional group of the edit model, :209).  x [B,C,h,w], c_concat [B,Cc,h,w] fp32, hw = h*w. */
int sdmi_cfg_prepare_concat(const void* x_f32, const void* c_in_f32, const void* c_concat_f32, void* x_in, int out_dtype,
                            int B, int reps, int C, int Cc, int64_t hw, uint32_t zero_reps, void* stream);

/* denoised[b] = u + (c - u) * cond_scale  with  c = x + eps_c * c_out[b], u = x + eps_u * c_out[b]
 * (eps = [cond(B) | uncond(B)], fp32 NCHW).  Replaces CompVisDenoiser's `input + eps * c_out` and
 * CFGDenoiser.combine_denoised (modules/sd_samplers_cfg_denoiser.py:74-82) for one cond of weight 1 per image.
 * mode 0: sigma-space (above);  mode 1: timestep-space (DDIM): out = eps_u + (eps_c - eps_u) * cond_scale.
 * Optional mask blend (img2img, :174-187): out = out*nmask + init_latent*mask when mask != NULL. */
int sdmi_cfg_combine(const void* x_f32, const void* eps_f32, const void* c_out_f32, float cond_scale, int mode,
                     const void* mask_f32, const void* nmask_f32, const void* init_latent_f32,
                     void* denoised_f32, int B, int64_t chw, void* stream);

/* CFG combine for v-prediction checkpoints (SD 2.x 768-v; `parameterization == "v"`, modules/sd_models_config.py:86-94):
 * each half of the UNet output goes through d = out * c_out[b] + x * c_skip[b] first — k-diffusion's CompVisVDenoiser in
 * sigma space (denoised; c_skip = 1/(s^2+1), c_out = -s/sqrt(s^2+1), selected at modules/sd_samplers_kdiffusion.py:60-62)
 * or CompVisTimestepsVDenoiser.predict_eps_from_z_and_v in timestep space (eps; c_out = sqrt(a_t), c_skip = sqrt(1 - a_t),
 * modules/sd_samplers_timesteps.py:33-45) — then u + (c - u) * cond_scale and the optional mask blend as above. */
int sdmi_cfg_combine_affine(const void* x_f32, const void* out_f32, const void* c_out_f32, const void* c_skip_f32,
                            float cond_scale, const void* mask_f32, const void* nmask_f32, const void* init_latent_f32,
                            void* denoised_f32, int B, int64_t chw, void* stream);

/* k-diffusion sample_euler_ancestral / sample_euler update (pinned k-diffusion@ab527a9; to_d per
 * modules/sd_schedulers.py:10-15):  d=(x-den)/sigma; x+=d*(sigma_down-sigma); x+=noise*s_noise*sigma_up (if noise). */
int sdmi_euler_step(void* x_f32, const void* denoised_f32, const void* noise_f32_or_null,
                    float sigma, float sigma_down, float sigma_up, float s_noise, int64_t n, void* stream);

/* k-diffusion sample_dpmpp_2m update: x = ratio*x - em1*(c1*den - c2*old_den)  (c2 = 0 on first/last step). */
int sdmi_dpmpp2m_step(void* x_f32, const void* denoised_f32, const void* old_denoised_f32_or_null,
                      float ratio, float em1, float c1, float c2, int64_t n, void* stream);

/* DDIM update (modules/sd_samplers_timesteps_impl.py:30-36):
 *   pred_x0=(x-sqrt_one_minus_at*e)/sqrt(a_t); x = sqrt(a_prev)*pred_x0 + sqrt(1-a_prev-sigma_t^2)*e + sigma_t*noise. */
int sdmi_ddim_step(void* x_f32, const void* e_t_f32, const void* noise_f32_or_null, void* pred_x0_f32_or_null,
                   float a_t, float a_prev, float sigma_t, float sqrt_one_minus_at, int64_t n, void* stream);

/* y = a*x + b*z elementwise fp32 (x*sigmas[0]; init_latent + noise*sigma: modules/sd_samplers_kdiffusion.py:143,199). */
int sdmi_axpby(void* y_f32, const void* x_f32, float a, const void* z_f32_or_null, float b, int64_t n, void* stream);

/* out = sum_k coefs[k] * terms[k] (1..6 fp32 tensors of n elements, accumulated left to right; out may alias a term).
 * The update rules of the k-diffusion samplers that have no dedicated kernel (Heun, DPM2, DPM2 a, LMS, DPM++ 2S a:
 * names at modules/sd_samplers_kdiffusion.py:11-27) and of PLMS (modules/sd_samplers_timesteps_impl.py:85-137) are
 * linear combinations of x, denoiser outputs and noise; `terms` / `coefs` are HOST arrays. */
int sdmi_lincomb(void* out_f32, const void* const* terms_f32, const float* coefs, int n_terms, int64_t n, void* stream);

/* DPM adaptive (k-diffusion DPMSolver.dpm_solver_adaptive; table row modules/sd_samplers_kdiffusion.py:25): the squared error of the
 * embedded solver pair under the mixed tolerance, sum_i ((lo_i - hi_i) / max(atol, rtol * max(|lo_i|, |prev_i|)))^2, as 256
 * per-block partial sums (fp32, fixed summation order: reproducible); the host adds them and takes sqrt(sum / n). */
int sdmi_dpm_error_partials(const void* x_low_f32, const void* x_high_f32, const void* x_prev_f32, float atol, float rtol,
                            void* partial256_f32, int64_t n, void* stream);

/* Latent upscale of the hires-fix pass: torch.nn.functional.interpolate(samples, size=(ho, wo), mode, antialias=False) on
 * `planes` = B*C fp32 planes of hi x wi (modules/processing.py:1392 with the "Latent*" upscalers of modules/shared.py:54-62).
 * mode: 0 "nearest", 1 "nearest-exact", 2 "bilinear", 3 "bicubic" (align_corners = False, ATen index arithmetic); 4 / 5: bilinear /
 * bicubic with antialias = True ("Latent (antialiased)", "Latent (bicubic antialiased)": ATen's separable area filter — windows that
 * widen with the scale when shrinking, weights renormalised inside the image, Keys a = -0.5 for the cubic). */
int sdmi_latent_resize(const void* in_f32, void* out_f32, int planes, int hi, int wi, int ho, int wo, int mode, void* stream);

/* x = init*mask + nmask*x, all fp32 tensors of n elements: the inpainting blend CFGDenoiser applies before / after
 * denoising (modules/sd_samplers_cfg_denoiser.py:206-209, 279-280). */
int sdmi_mask_blend(void* x_f32, const void* init_f32, const void* mask_f32, const void* nmask_f32, int64_t n, void* stream);

/* clamp((x+1)/2,0,1)*255 truncated to uint8, NCHW fp32 -> NHWC uint8 (modules/processing.py:1004-1005,1034-1035). */
int sdmi_image_to_u8(const void* img_f32_nchw, void* out_u8_nhwc, int B, int C, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Engine (boundaries B1 "SdUnet" and B4 VAE): whole-UNet forward and whole-VAE decode on packed weights.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct sdmi_unet_config {
    int32_t in_channels, out_channels, model_channels;
    int32_t num_levels;
    int32_t channel_mult[8];
    int32_t num_res_blocks;
    int32_t attn_level[8];          /* 1 if that level has SpatialTransformers */
    int32_t transformer_depth[8];
    int32_t num_heads;              /* used when num_head_channels == -1 */
    int32_t num_head_channels;
    int32_t context_dim;
    int32_t adm_in_channels;        /* 0 = none */
    int32_t reserved[4];
} sdmi_unet_config;

typedef struct sdmi_vae_config {
    int32_t ch, num_levels;
    int32_t ch_mult[8];
    int32_t num_res_blocks, in_channels, out_ch, z_channels;
    float scale_factor;
    int32_t reserved[4];
} sdmi_vae_config;

sdmi_engine* sdmi_engine_create(int device);
void sdmi_engine_destroy(sdmi_engine* e);

/* Weight hand-over, one tensor at a time, by checkpoint key WITHOUT the "model.diffusion_model." / "first_stage_model."
 * prefix.  Replaces the torch module tree the reference loader fills (modules/sd_models.py:410-538 load_model_weights):
 * the caller keeps using read_state_dict (:312-329) and streams the dict here.  `data` may be a device or host pointer
 * (on_device flag); fp16 or fp32; shape as in the checkpoint (OIHW conv / [out,in] linear). */
int sdmi_unet_configure(sdmi_engine* e, const sdmi_unet_config* cfg);
int sdmi_unet_load_tensor(sdmi_engine* e, const char* key, const void* data, int dtype, int ndim,
                          const int64_t* shape, int on_device);
int sdmi_unet_finalize(sdmi_engine* e);            /* packs layouts; errors if a required key is missing */

/* Replace ONE conv / linear weight ("<layer>.weight", same shape as loaded) of the finalized UNet: re-packed in place.
 * This is where the weight rewrite of extensions-builtin/Lora/networks.py:411-480 (network_apply_weights: restore the
 * backup, add every loaded network's delta, copy into the layer) lands for a UNet that no longer owns torch modules.
 * Call sdmi_unet_set_context again afterwards (cached cross-attention projections depend on attn2.to_k / to_v). */
int sdmi_unet_update_weight(sdmi_engine* e, const char* key, const void* data, int dtype, int ndim,
                            const int64_t* shape, int on_device);

/* Hypernetworks (modules/hypernetworks/hypernetwork.py): per feature width (320 / 640 / 768 / 1280 ...) a pair of small MLPs that
 * transform the attention context before to_k / to_v — context_k = x + multiplier * MLP_k(x) (HypernetworkModule.forward :104-105,
 * apply_hypernetworks :358-379, called from every CrossAttention.forward: sd_hijack_optimizations.py:231, hypernetwork.py:388).  The
 * engine applies them inside its attention layers (self-attention: on the normalised tokens; cross-attention: on the text context
 * when its K / V are projected).  Hand-over: clear, then per loaded hypernetwork `begin(multiplier)` followed, for every width `dim`
 * and `which` (0 = K module, 1 = V module), by its op sequence in order: linear ([out, in] weight + bias, nn.Linear layout),
 * activation (1 relu, 2 leakyrelu, 3 elu, 4 hardswish = the webui's "swish", 5 tanh, 6 sigmoid, 7 silu, 8 gelu, 9 mish, 10 relu6,
 * 11 selu, 12 softplus, 13 softsign, 14 hardtanh, 15 hardsigmoid), layer norm.  Dropout layers are the identity at inference. */
int sdmi_unet_hypernet_clear(sdmi_engine* e);
int sdmi_unet_hypernet_begin(sdmi_engine* e, float multiplier);
int sdmi_unet_hypernet_linear(sdmi_engine* e, int dim, int which, const void* w, const void* b, int dtype, int out_features,
                              int in_features, int on_device);
int sdmi_unet_hypernet_act(sdmi_engine* e, int dim, int which, int act);
int sdmi_unet_hypernet_layernorm(sdmi_engine* e, int dim, int which, const void* gamma, const void* beta, int dtype, int n, int on_device);

/* The same for a 1-D parameter: "<layer>.bias" of a conv / linear layer, "<norm>.weight" / "<norm>.bias" of a GroupNorm / LayerNorm.
 * Where the bias deltas (`ex_bias`: network_full.py diff_b) and the LyCORIS norm modules (network_norm.py w_norm / b_norm) of
 * extensions-builtin/Lora/networks.py:411-480 land. */
int sdmi_unet_update_vector(sdmi_engine* e, const char* key, const void* data, int dtype, int64_t n, int on_device);

/* out[rows*cols] fp32 = W + scale * (up[rows][rank] @ down[rank][cols]); W / up / down fp16 or fp32 device tensors.
 * The LoRA delta of extensions-builtin/Lora/network_lora.py:65-80 with lyco_helpers.rebuild_conventional (:9-15) and
 * network.py:196-216 finalize_updown (scale = alpha / rank * multiplier) folded into the weight in one pass. */
int sdmi_lora_merge(void* out_f32, const void* w, int w_dtype, const void* up, int up_dtype, const void* down, int down_dtype,
                    int rows, int cols, int rank, float scale, void* stream);

/* The other LyCORIS module types of extensions-builtin/Lora (dispatch order networks.py:26-36), all on fp32 device buffers with the
 * layer's current weight W viewed [rows][cols], cols = Cin*kh*kw; the small factor products they start from are plain
 * sdmi_lora_merge calls over a zero W.
 *   hadamard  LoHa, network_hada.py:28-55:   out = W + scale * a * b                      (a, b = the two rebuilt products)
 *   kron      LoKr, network_lokr.py:19-23:   out = W + scale * kron(w1[r1][c1], w2[r2][c2][k])   (k = kh*kw taps of w2)
 *   ia3       IA3,  network_ia3.py:18-30:    out = W + scale * W * v[col if on_input else row]
 *   dora      DoRA, network.py:175-194:      out = W + mult * ((W + delta) * dora_scale[j] / ||(W + delta)[:, j]|| - W), the
 *             norm per input channel j over (out, kh, kw); delta = the module's updown * calc_scale (fp32 [rows][cin][k]). */
int sdmi_weight_hadamard(void* out_f32, const void* w_f32, const void* a_f32, const void* b_f32, float scale, int64_t n, void* stream);
int sdmi_weight_kron(void* out_f32, const void* w_f32, const void* w1_f32, const void* w2_f32, int r1, int c1, int r2, int c2, int k,
                     float scale, void* stream);
int sdmi_weight_ia3(void* out_f32, const void* w_f32, const void* v_f32, int rows, int cols, int on_input, float scale, void* stream);
int sdmi_weight_dora(void* out_f32, const void* w_f32, const void* delta_f32, const void* dora_scale_f32, int rows, int cin, int k,
                     float mult, void* stream);

int sdmi_vae_configure(sdmi_engine* e, const sdmi_vae_config* cfg);
int sdmi_vae_load_tensor(sdmi_engine* e, const char* key, const void* data, int dtype, int ndim,
                         const int64_t* shape, int on_device);
int sdmi_vae_finalize(sdmi_engine* e);

/* eps = UNet(x, timesteps, context[, y]).  Replaces SdUnet.forward (modules/sd_unet.py:75-83), called from the patched
 * UNetModel.forward (modules/sd_unet.py:86-93, modules/sd_hijack.py:41-45).
 * x [Bn, Cin, h, w] NCHW (io_dtype), timesteps [Bn] (io_dtype), context [Bn, L, context_dim] (io_dtype) or NULL to reuse
 * the K/V projections cached by the previous call / sdmi_unet_set_context, y [Bn, adm] or NULL, out [Bn, Cout, h, w]. */
int sdmi_unet_set_context(sdmi_engine* e, const void* context, int io_dtype, int Bn, int L, void* stream);
/* The same for a caller that cannot tell whether `context` changed since the previous call — the webui hands SdUnet.forward a
 * freshly catenated cond | uncond tensor on every sampling step (modules/sd_samplers_cfg_denoiser.py:246; it only really changes
 * when a prompt-editing schedule switches).  The decision is taken ON THE DEVICE, without a host synchronisation: a compare kernel
 * raises a flag if any fp16-converted element differs from the cached copy, and the copy and all K / V^T projection launches are
 * predicated on that flag (empty launches otherwise).  Follow with sdmi_unet_forward(..., context = NULL, ...). */
int sdmi_unet_set_context_cached(sdmi_engine* e, const void* context, int io_dtype, int Bn, int L, void* stream);
int sdmi_unet_forward(sdmi_engine* e, const void* x, const void* timesteps, const void* context, const void* y,
                      void* out, int io_dtype, int Bn, int h, int w, int L, void* stream);
/* The same with the caller's PROMISES about this call's data as an argument of the call (round 6; ADVICE r4 / VERDICT r5: as sticky
 * engine options — sdmi_engine_set_option "uniform_t" / "cfg_pairs", still accepted by sdmi_unet_forward — a stale promise silently
 * overwrote the second half of a later batch).  call_flags, valid for THIS call only:
 *   SDMI_CALL_UNIFORM_T   every row sits at the same timestep (the samplers' CFG batch, modules/sd_samplers_cfg_denoiser.py:230-246):
 *                         the timestep-embedding path runs for one row; same bits
 *   SDMI_CALL_CFG_PAIRS   rows [Bn/2, Bn) repeat the latent and timestep of rows [0, Bn/2) (the [cond | uncond] batch): the layers in
 *                         front of the first cross-attention run for one half
 *   SDMI_CALL_DERIVE      neither is known (the stock CFG denoiser behind SdUnet.forward, modules/sd_unet.py:86-93): the engine derives
 *                         both from x and timesteps — one synchronising device -> host compare */
enum { SDMI_CALL_UNIFORM_T = 1, SDMI_CALL_CFG_PAIRS = 2, SDMI_CALL_DERIVE = 4 };
/* Extra UNet inputs of SdUnet.forward(x, timesteps, context, *args, **kwargs) (modules/sd_unet.py:76-77, 87-91 pass them through): the
 * ControlNet residuals `control` of ldm's ControlledUnetModel.forward (cldm.py) for the NEXT sdmi_unet_forward[_ex] call only — n =
 * (number of input blocks) + 1 device tensors in the io dtype of that call, NCHW, tensor i shaped like input block i's output
 * [Bn, C_i, h_i, w_i] and the last like the middle block's; numel[i] is checked against that shape.  They are added to the skip
 * connections as the output blocks read them and to the middle block's output; only_mid_control: the last one alone.  n = 0 clears. */
int sdmi_unet_set_control(sdmi_engine* e, const void* const* tensors, const int64_t* numel, int n, int only_mid_control);
int sdmi_unet_forward_ex(sdmi_engine* e, const void* x, const void* timesteps, const void* context, const void* y,
                         void* out, int io_dtype, int Bn, int h, int w, int L, int call_flags, void* stream);

/* image = decoder(post_quant_conv(z / scale_factor)), all B latents in one batched pass.
 * Replaces decode_latent_batch's per-image loop (modules/processing.py:625-672) -> decode_first_stage
 * (modules/sd_samplers_common.py:73-76) -> LatentDiffusion.decode_first_stage / AutoencoderKL.decode (third-party).
 * z [B, 4, h, w] NCHW (io_dtype); out fp32 NCHW [B, 3, 8h, 8w] in [-1, 1] (what the caller .float()s anyway). */
int sdmi_vae_decode(sdmi_engine* e, const void* z, int io_dtype, void* out_f32, int B, int h, int w, void* stream);

/* latent moments = quant_conv(encoder(x)); x [B,3,H,W] NCHW in [-1,1]; out fp32 NCHW [B, 2*z, H/8, W/8] (mean | logvar).
 * Replaces encode_first_stage (modules/sd_samplers_common.py:87-112 -> third-party AutoencoderKL.encode). */
int sdmi_vae_encode(sdmi_engine* e, const void* x, int io_dtype, void* out_f32, int B, int H, int W, void* stream);

/* Introspection for tests / bench. */
int64_t sdmi_engine_arena_bytes(sdmi_engine* e);
int sdmi_engine_set_option(sdmi_engine* e, const char* name, int value);   /* "force_generic", "glds", "trace" (activation taps),
                                                                               "tiling" (p.tiling: padded 3x3 convs wrap around, modules/sd_hijack.py:311-318),
                                                                               "vae_range_extend" (VAE decoder residual stream at 1/64 scale: the
                                                                               engine's form of the fp16 -> fp32 VAE fallback, modules/processing.py:636-665),
                                                                               "cfg_pairs" / "uniform_t" (promises of the caller about the next forwards' x and t:
                                                                               rows [Bn/2, Bn) repeat rows [0, Bn/2); one timestep for all rows),
                                                                               "auto_promises" (both derived per forward from x and t by a synchronising compare:
                                                                               for callers that cannot know), "fuse_rows", "residual_fp32"  */

/* Activation taps for the parity error budget (tests/test_gpu_c1_parity.py): with option "trace" = 1 the engine records, by
 * the reference's module name ("input_blocks.4.1", "middle_block.1.transformer_blocks.0", "decoder.up.2.block.1", ...), the
 * NHWC fp16 output of every block of the LAST UNet forward / VAE decode; the tensors stay valid until the next forward
 * (the activation arena never reuses memory within one).  tap_read copies tap `index` ([B][H][W][C] fp16) to a device buffer. */
int sdmi_engine_tap_count(sdmi_engine* e);
int sdmi_engine_tap_info(sdmi_engine* e, int index, char* name_out, int capacity, int64_t* dims_bhwc);
int sdmi_engine_tap_read(sdmi_engine* e, int index, void* out_f16_nhwc, void* stream);

/* Tuning knobs for benchmarks: "gemm_cfg" (-1 heuristic, 0..7 force a tile configuration when it fits the shape),
 * "attn_kvt" (0 heuristic, 64 force 64-key tiles), "attn_tau" (slack of the flash kernels' lazy exponent re-basing in log2 units:
 * default 8; 0 = re-base whenever a running maximum moves, -1 = also rescale O in every tile — both give the round-2 bits),
 * "attn_fold_min_m" (shortest d = 40 self-attention that takes the folded-shift form: default 1024, 0 = never). */
int sdmi_debug_set(const char* name, int value);
/* String-valued knob: "gemm_override" = "M,N,K,taps,kind:cfg:split;..." forces a tile configuration / split-K factor for exact GEMM
 * shapes (kind 0 plain epilogue, 1 GEGLU, 2 transposed output; "" clears) — the in-engine shape autotuner tools/gpu/shape_tune.py. */
int sdmi_debug_set_str(const char* name, const char* value);

/* Optional per-launch HIP-event profiler: between begin and end every kernel launch of the library is bracketed by events
 * on its own stream; end() synchronises once and writes {"kernels":[{"name","launches","ms","flops","bytes"},...]} with the
 * ALGORITHMIC flops / bytes of each launch into json_out. */
int sdmi_profile_begin(void);
int sdmi_profile_end(char* json_out, int capacity);

/* Micro-benchmarks used by bench.py's roofline block (HIP-event timed inside the library; returns ms per launch). */
int sdmi_bench_conv_gemm(const sdmi_conv_desc* d, int iters, float* ms_out, void* stream);

/* ---- CLIP text encoder ---------------------------------------------------------------------------
 * The transformer behind FrozenCLIPEmbedderWithCustomWords.encode_with_transformers (modules/sd_hijack_clip.py:351-360:
 * `self.wrapped.transformer(input_ids=tokens, output_hidden_states=-opts.CLIP_stop_at_last_layers)`, i.e. transformers'
 * CLIPTextModel; in-repo plain-torch twin modules/models/sd3/other_impls.py:61-150): token + position embeddings,
 * `layers` pre-LN blocks (causal self-attention, MLP with quick_gelu or gelu), final LayerNorm.  Prompt parsing, chunking,
 * emphasis and textual-inversion vector injection stay on the host (they feed `inputs_embeds`).  Two slots: SDXL carries
 * two text encoders. */
typedef struct sdmi_clip_config {
    int vocab_size;        /* 49408 */
    int max_positions;     /* 77 */
    int hidden;            /* 768 (CLIP-L), 1280 (OpenCLIP bigG) */
    int layers;            /* 12 / 32 */
    int heads;             /* hidden / 64 */
    int intermediate;      /* 4 * hidden */
    int act;               /* 0 = quick_gelu, 1 = gelu (erf) */
    float eps;             /* LayerNorm eps, 1e-5 */
} sdmi_clip_config;

int sdmi_clip_configure(sdmi_engine* e, int slot, const sdmi_clip_config* cfg);
/* keys as in transformers' CLIPTextModel state dict below "text_model.": "embeddings.token_embedding.weight",
 * "embeddings.position_embedding.weight", "encoder.layers.<i>.{layer_norm1,layer_norm2}.{weight,bias}",
 * "encoder.layers.<i>.self_attn.{q_proj,k_proj,v_proj,out_proj}.{weight,bias}", "encoder.layers.<i>.mlp.{fc1,fc2}.{weight,bias}",
 * "final_layer_norm.{weight,bias}", and optionally "text_projection.weight" [proj_dim, hidden] (nn.Linear layout). */
int sdmi_clip_load_tensor(sdmi_engine* e, int slot, const char* key, const void* data, int dtype, int ndim,
                          const int64_t* shape, int on_device);
int sdmi_clip_finalize(sdmi_engine* e, int slot);
/* tokens int32 [B, L] (device); inputs_embeds fp32 [B, L, hidden] or NULL (token embeddings already looked up / patched by
 * the caller: textual inversion, modules/sd_hijack.py EmbeddingsWithFixes).  Runs the first `layers - skip + 1` blocks
 * (skip = opts.CLIP_stop_at_last_layers >= 1: hidden_states[-skip]) and, if apply_final_ln, final_layer_norm — which is
 * last_hidden_state for skip = 1 and the clip-skip branch of sd_hijack_clip.py:354-356 otherwise; SDXL's CLIP-L takes
 * hidden_states[-2] without the norm (sd_hijack_clip.py:369-377).  out fp32 [B, L, hidden]; pooled fp32 [B, proj_dim or
 * hidden] or NULL = final_layer_norm(last block)'s row at the EOS (largest id) position, times text_projection when loaded
 * (transformers pooler_output / text_embeds; open_clip `pool(ln_final(x)) @ text_projection`, the SDXL "pooled" vector read
 * at modules/sd_hijack_open_clip.py:60-66) — always taken after the LAST block, whatever `skip` is. */
int sdmi_clip_forward(sdmi_engine* e, int slot, const void* tokens_i32, const void* inputs_embeds_f32_or_null, int B, int L,
                      int skip, int apply_final_ln, void* out_f32, void* pooled_f32_or_null, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDMI_H */
