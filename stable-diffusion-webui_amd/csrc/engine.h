// engine.h — host-side engine of libsdmi: weight registry, packed layouts, activation arena and the UNet / VAE
// launch graphs.  One engine per GPU, used by one caller at a time (the reference serialises all GPU jobs behind one
// FIFO lock: /root/reference/modules/call_queue.py:8, modules/fifo_lock.py:6-37).
#pragma once
#include "common.h"
#include "../../include/sdmi.h"

#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace sdmi {

struct RawTensor {
    void* ptr = nullptr;          // device copy (as given dtype)
    int dtype = 0;
    std::vector<int64_t> shape;
    size_t bytes = 0;
};

// packed conv / linear weights: w [n_pad][taps][cin_pad] fp16, b [n_pad] fp32 (may be null)
struct ConvW {
    half_t* w = nullptr;
    float* b = nullptr;
    int cin = 0, cin_pad = 0, cout = 0, n_pad = 0, taps = 1;
    bool geglu = false;
    // LayerNorm folded into this layer (engine option "ln_fold"; GemmP::ln_stats): w_ln = fp16(w * gamma), s_ln[n] = sum_k w_ln[n][k],
    // c_ln = beta . w^T + b.  Built lazily by ensure_ln_fold, rebuilt when fold_epoch falls behind the engine's weights_epoch (any
    // sdmi_unet_update_weight / _vector).  mutable: the packed model is const during a forward, the cache is not part of it.
    mutable half_t* w_ln = nullptr;
    mutable float* s_ln = nullptr;
    mutable float* c_ln = nullptr;
    mutable long fold_epoch = -1;
};
struct NormW {
    float* g = nullptr;
    float* b = nullptr;
    int c = 0;
};

struct ResW {
    NormW n1, n2;
    ConvW c1, c2, skip;
    bool has_skip = false;
    int cin = 0, cout = 0;
    int emb_off = -1;             // column offset into the fused emb projection output (UNet only)
};
struct TBlockW {
    NormW ln1, ln2, ln3;
    ConvW qk1;                    // [2C][C]  (to_q ; to_k)
    ConvW v1;                     // [C][C]   used as the "activation" operand of the V^T GEMM
    ConvW o1, q2, k2, v2, o2, ff1, ff2;
    int ctx_slot = -1;            // index into the per-layer context K / V^T cache
    // feed-forward chain (engine option "fuse_rows", rowchain.hip): packed operand stream of ff.net.0.proj / ff.net.2, built lazily,
    // rebuilt when it falls behind the engine's weights_epoch
    mutable char* ff_packs = nullptr;
    mutable long ff_epoch = -1;
};
struct STW {
    NormW norm;
    ConvW proj_in, proj_out;
    std::vector<TBlockW> blocks;
    int ch = 0, heads = 0, dhead = 0;
};
struct UNetLayer {
    enum Kind { CONV_IN, RES, ST, DOWN, UP } kind;
    ResW res;
    STW st;
    ConvW conv;                   // CONV_IN / DOWN / UP
    int c0 = 0, c1 = 0;           // RES in output blocks: split of the concatenated input (h, skip)
};
struct UNetW {
    sdmi_unet_config cfg{};
    ConvW te0, te2, le0, le2;     // time_embed / label_emb linears (small-M path: w is plain [N][K] fp16)
    ConvW emb_all;                // all ResBlock emb_layers.1 stacked: [sum Cout][ted]
    int emb_cols = 0;
    std::vector<std::vector<UNetLayer>> input, output;
    std::vector<UNetLayer> middle;
    NormW out_norm;
    ConvW out_conv;
    int n_ctx_slots = 0;
    bool ready = false;
};

struct VAEAttnW {
    NormW norm;
    ConvW qk, v, proj;
    int c = 0;
};
struct VAELevel {
    std::vector<ResW> blocks;
    ConvW resample;               // upsample.conv (decoder) / downsample.conv (encoder)
    bool has_resample = false;
};
struct ClipLayerW {
    NormW ln1, ln2;
    ConvW qk, v, o, fc1, fc2;     // q;k stacked, V separate (produced transposed), out_proj, MLP
};
struct ClipW {
    sdmi_clip_config cfg{};
    void* tok_emb = nullptr;      // [vocab][hidden] in the checkpoint dtype
    int tok_dtype = 0;
    float* pos_emb = nullptr;     // [max_positions][hidden] fp32
    std::vector<ClipLayerW> layers;
    NormW final_ln;
    half_t* text_proj = nullptr;  // optional [proj_dim][hidden] fp16 (rows = output features): pooled @ text_projection
    int proj_dim = 0;
    bool configured = false, ready = false;
};

struct VAEW {
    sdmi_vae_config cfg{};
    // decoder
    float* pqc_w = nullptr;       // post_quant_conv as fp32 [z][z]
    float* pqc_b = nullptr;
    ConvW d_conv_in, d_conv_out;
    ResW d_mid1, d_mid2;
    VAEAttnW d_attn;
    std::vector<VAELevel> d_up;   // index = level (0 = finest)
    NormW d_norm_out;
    // encoder
    ConvW e_conv_in, e_conv_out;  // e_conv_out has quant_conv folded in
    ResW e_mid1, e_mid2;
    VAEAttnW e_attn;
    std::vector<VAELevel> e_down;
    NormW e_norm_out;
    bool has_encoder = false;
    bool ready = false;
};

// Hypernetworks (modules/hypernetworks/hypernetwork.py): per feature width a pair of small MLPs (one for the K path, one for the V path)
// applied to the attention CONTEXT before to_k / to_v: context_k = x + multiplier * MLP_k(x) (forward, :104-105; apply_hypernetworks
// :358-379).  An MLP is a sequence of Linear / activation / LayerNorm ops (dropout is the identity at inference).
struct HnOp {
    int kind = 0;                 // 0 linear, 1 activation, 2 layer norm
    ConvW lin;
    int act = 0;
    NormW ln;
};
struct HnModule {
    std::vector<HnOp> ops;
};
struct HnNet {
    float multiplier = 1.f;
    std::map<int, std::pair<HnModule, HnModule>> by_dim;    // feature width -> (K module, V module)
};

class Arena {
public:
    char* base = nullptr;
    size_t cap = 0, off = 0, high = 0;
    bool dry = false;
    void reset() { off = 0; }
    size_t mark() const { return off; }
    void rewind(size_t m) { off = m; }            // everything taken since mark() is dead: later launches of the same stream may overwrite it
    void* take(size_t bytes) {
        const size_t a = (off + 255) & ~size_t(255);
        off = a + bytes;
        if (off > high) high = off;
        if (dry) return (void*)(uintptr_t)(0x1000 + a);   // never dereferenced in dry mode
        return base + a;
    }
};

}  // namespace sdmi

struct sdmi_engine {
    int device = 0;
    std::map<std::string, sdmi::RawTensor> raw_unet, raw_vae;
    // where each "<layer>.weight" of the UNet checkpoint landed in the packed weights (filled by the build), so that a single
    // layer can be re-packed in place when a LoRA changes it (sdmi_unet_update_weight)
    struct PackSite { half_t* dst; int O, cin, kh, Opad, cin_pad, geglu; };
    std::map<std::string, std::vector<PackSite>> unet_sites;
    // the same for the 1-D parameters ("<layer>.bias", "<norm>.weight", "<norm>.bias"): LyCORIS norm modules and bias deltas
    // (extensions-builtin/Lora/network_norm.py, network.py:196-216 ex_bias) rewrite them through sdmi_unet_update_vector
    struct VecSite { float* dst; int n, n_pad, geglu; };
    std::map<std::string, std::vector<VecSite>> unet_vec_sites;
    bool recording_unet_sites = false;
    std::vector<void*> owned;                 // persistent device allocations (weights)
    std::vector<void*> owned_vae;             // the VAE's packed weights: freed when another VAE is loaded (sd_vae.load_vae)
    std::vector<void*>* alloc_sink = nullptr; // where dev_alloc records (null = owned)
    std::vector<sdmi::HnNet> hypernets;       // loaded hypernetworks, in application order (shared.loaded_hypernetworks)
    std::vector<void*> owned_hn;
    char* hn_ctx_scratch = nullptr;           // intermediates of the hypernetwork pass over the text context (set_context)
    size_t hn_ctx_scratch_bytes = 0;
    sdmi::UNetW unet;
    sdmi::VAEW vae;
    sdmi::ClipW clip[2];
    std::map<std::string, sdmi::RawTensor> raw_clip[2];
    sdmi::Arena arena;
    // option "streams" (> 1): a UNet call's rows are cut into that many slices, each on its own HIP stream / arena (engine.cpp unet_forward)
    int n_streams = 1;
    int arena_reuse = 0;                      // 1: the temporaries of a ResBlock / transformer block are released when the block returns (option "arena_reuse")
    std::vector<hipStream_t> aux_streams;
    std::vector<sdmi::Arena> aux_arenas;
    std::vector<hipEvent_t> ev_join;
    hipEvent_t ev_fork = nullptr;
    // options
    bool force_generic = false;
    bool use_glds = true;
    // range-extended VAE decode (the engine's form of the reference's fp16 -> fp32 VAE fallback, modules/processing.py:636-665): the
    // decoder's residual stream is stored at this scale (1/64 when on), every GroupNorm that reads it uses eps * scale^2
    float vae_stream_scale = 1.f;
    // LayerNorm folded into the consuming GEMMs of the transformer blocks (norm1 -> to_q|to_k, to_v; norm2 -> attn2.to_q; norm3 ->
    // ff.net.0): only the per-row (mean, rstd) are computed, the normalised tensors never reach HBM.  Off by default until measured.
    int ln_fold = 0;                          // 1: row statistics from ln_rowstats_kernel; 2: also per-tile partial sums from the producing GEMMs' epilogues
    // The feed-forward chain of the transformer blocks at the 320-wide level as a single launch (rowchain.hip), option "fuse_rows" bit 1:
    // norm3 -> GEGLU -> ff.net.2 -> + x.  Default 2 (round 5, same-box A/Bs of the forward, profiles/r05_fwd_ab_*): 203-219 us against
    // 263 us of LayerNorm + GEGLU GEMM + GEMM.  (Bit 0 was the cross-attention chain: measured slower, removed in round 6; the bit is ignored.)
    int fuse_rows = [] { const char* e = getenv("SDMI_FUSE_ROWS"); return e ? atoi(e) : 2; }();
    // Accuracy mode (option "residual_fp32", off by default — the engine's counterpart of the reference's --no-half / upcast options,
    // modules/devices.py:284-295, modules/sd_hijack_optimizations.py:232-233): the UNet's carried stream — conv_in / ResBlock /
    // transformer-block / proj_out / down- and upsample outputs and the skip_connection 1x1 — is kept as (hi, lo) fp16 pairs
    // (GemmP::out_lo), i.e. with ~22 bits, so the residual sums are no longer rounded to fp16 once per block.  DESIGN.md section 7.
    bool residual_fp32 = false;
    long weights_epoch = 0;                   // bumped by every in-place weight / vector update: folded copies older than this are stale
    bool cfg_pairs = false;                   // rows [Bn/2, Bn) repeat the latent and timestep of rows [0, Bn/2): the layers in front of the first cross-attention run once (option "cfg_pairs")
    bool uniform_t = false;                   // every row of the call sits at the same timestep: the embedding path runs for one row (option "uniform_t")
    bool auto_promises = false;               // option "auto_promises": cfg_pairs / uniform_t are DERIVED per call from x and t (a synchronising device -> host
                                              // compare) instead of promised by the caller — for callers that cannot know: the stock CFG denoiser behind Mi355xUnet.forward
    bool tiling = false;                      // p.tiling: every padded 3x3 conv wraps around (modules/sd_hijack.py:311-318)
    // activation taps (parity error budget): with `trace` on, every block output of the last forward is recorded by name
    // — the arena never reuses memory within a forward, so the tensors stay readable until the next forward
    bool trace = false;
    struct Tap { std::string name; const half_t* ptr; int B, H, W, C; };
    std::vector<Tap> taps;
    // ControlNet residuals for the NEXT forward only (sdmi_unet_set_control; consumed and cleared by it): one NCHW tensor per input
    // block output + one for the middle block, in the caller's io dtype (ldm cldm.py ControlledUnetModel.forward)
    std::vector<const void*> control;
    std::vector<int64_t> control_numel;
    bool only_mid_control = false;
    // context cache (persistent between forwards)
    half_t* ctx_f16 = nullptr;                // [Bn][Lpad][ctx_dim]
    std::vector<half_t*> ctx_k;               // per slot [Bn*Lpad][C]
    std::vector<half_t*> ctx_vt;              // per slot [Bn][C][Lpad]
    std::vector<void*> ctx_owned;
    int ctx_B = 0, ctx_L = 0, ctx_Lpad = 0;
    bool ctx_valid = false;
    int* ctx_gate = nullptr;                  // device flag of the conditional re-projection (sdmi_unet_set_context_cached)

    ~sdmi_engine();
};
