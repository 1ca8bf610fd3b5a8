// engine.cpp — weight packing, activation arena and the UNet / VAE launch sequences of libsdmi (host side, C++).
//
// The UNet walk restates the constructor of ldm's UNetModel (third-party, Stability-AI/stablediffusion@cf1d67a6;
// architecture from /root/reference/configs/v1-inference.yaml:29-44 and configs/sd_xl_inpaint.yaml:19-37; module layout
// pinned in-tree at /root/reference/extensions-builtin/Lora/networks.py:43-98), and the forward order of
// /root/reference/modules/sd_hijack_unet.py:83-102 (SpatialTransformer) — but as a flat sequence of HIP launches on NHWC
// fp16 buffers: no module tree, no torch.cat (dual-source GroupNorm / GEMM), no F.interpolate (upsample fused in the conv
// gather), no permutes (NHWC tokens == NHWC pixels), V written pre-transposed for the attention kernel.
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace sdmi {

// ------------------------------------------------------------------------------------------------------------
// error string, zero page
// ------------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
void set_error(const std::string& m) { g_err = m; }
const char* get_error() { return g_err.c_str(); }

const half_t* zero_page() {
    static thread_local int cached_dev = -1;
    static thread_local half_t* cached = nullptr;
    static std::map<int, half_t*> pages;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (dev == cached_dev) return cached;
    auto it = pages.find(dev);
    if (it == pages.end()) {
        void* p = nullptr;
        if (hipMalloc(&p, kZeroPageHalfs * sizeof(half_t)) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, kZeroPageHalfs * sizeof(half_t)) != hipSuccess) return nullptr;
        it = pages.emplace(dev, (half_t*)p).first;
    }
    cached_dev = dev;
    cached = it->second;
    return cached;
}

static inline int rup(int x, int m) { return (x + m - 1) / m * m; }

struct Run {
    sdmi_engine* e;
    hipStream_t s;
    bool dry;
    Arena* ar;                     // the activation arena this pass bump-allocates from (one per concurrent batch slice: option "streams")
    int b0 = 0, Btot = 0;          // UNet batch slices: first row of this slice / rows of the whole call (indexes the context cache)
    Run(sdmi_engine* e_, hipStream_t s_, bool dry_, Arena* ar_ = nullptr, int b0_ = 0, int Btot_ = 0)
        : e(e_), s(s_), dry(dry_), ar(ar_ ? ar_ : &e_->arena), b0(b0_), Btot(Btot_) {}
    half_t* H(size_t n) { return (half_t*)ar->take(n * sizeof(half_t)); }
    // Option "residual_fp32": a tensor of the carried stream is a (hi, lo) pair of fp16 tensors, allocated back to back — S(n) takes
    // room for both, lo(p, n) is the second half (null when the option is off: every consumer then sees a plain fp16 tensor).
    // (the pairs need the LDS-direct MFMA kernels: under force_generic / use_glds = 0 — cross-check settings — the forward falls back to
    // the plain fp16 stream instead of failing the job)
    bool acc() const { return e->residual_fp32 && !e->force_generic && e->use_glds; }
    half_t* S(size_t n) { return H(acc() ? 2 * n : n); }
    half_t* lo(const half_t* p, size_t n) const { return (acc() && p) ? const_cast<half_t*>(p) + n : nullptr; }
    // option "arena_reuse": a block's temporaries are released when the block returns — the next block's launches (same stream, so
    // ordered behind every reader) write over them while their lines are still in the 256 MB Infinity Cache, instead of every
    // activation of a forward (~10 GB) being written back to HBM once.  Not while block outputs are tapped or LayerNorm partials live.
    bool reuse() const { return e->arena_reuse && !e->trace && !e->ln_fold && !acc(); }
    float* F(size_t n) { return (float*)ar->take(n * sizeof(float)); }
    void tap(const std::string& name, const half_t* p, int B, int H, int W, int C) {
        if (!dry && e->trace) e->taps.push_back({name, p, B, H, W, C});
    }
    // GroupNorm statistics handed from a producing GEMM's epilogue to the norm that reads its output (launch_gemm: stats_out)
    const half_t* st_tensor = nullptr;
    float* st_ws = nullptr;
    int st_nchunk = 0;
    // LayerNorm row partial sums handed from a producing GEMM's epilogue to the folded consumers of its output (option "ln_fold"):
    // set lnp_want before the producer's run_conv; lnp_tensor / lnp_ws / lnp_np describe what the last run_conv produced
    bool lnp_want = false;
    const half_t* lnp_tensor = nullptr;
    float* lnp_ws = nullptr;
    int lnp_np = 0;
};

#define TRY(x)                  \
    do {                        \
        if ((x) != 0) return 1; \
    } while (0)

// ------------------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------------------
static int dev_alloc(sdmi_engine* e, void** p, size_t bytes) {
    SDMI_CHECK_HIP(hipMalloc(p, std::max<size_t>(bytes, 256)));
    (e->alloc_sink ? *e->alloc_sink : e->owned).push_back(*p);
    return 0;
}

static const RawTensor* find_raw(const std::map<std::string, RawTensor>& m, const std::string& k) {
    auto it = m.find(k);
    return it == m.end() ? nullptr : &it->second;
}

// pack `names` (each a conv / linear "<name>.weight" [+ ".bias"]) stacked along the output dim into one ConvW
static int pack_stack(sdmi_engine* e, const std::map<std::string, RawTensor>& m, const std::vector<std::string>& names,
                      bool pad64, bool geglu, ConvW* out, bool dup_in = false) {
    int cin = -1, taps = -1, total = 0;
    bool any_bias = false;
    for (auto& n : names) {
        const RawTensor* w = find_raw(m, n + ".weight");
        SDMI_REQUIRE(w != nullptr, "missing weight " + n + ".weight");
        SDMI_REQUIRE(w->shape.size() == 2 || w->shape.size() == 4, "weight must be 2-D or 4-D: " + n);
        const int I = (int)w->shape[1];
        const int t = w->shape.size() == 4 ? (int)(w->shape[2] * w->shape[3]) : 1;
        SDMI_REQUIRE(cin < 0 || (cin == I && taps == t), "stacked weights disagree in shape: " + n);
        cin = I; taps = t;
        total += (int)w->shape[0];
        if (find_raw(m, n + ".bias")) any_bias = true;
    }
    SDMI_REQUIRE(taps == 1 || taps == 9, "only 1x1 and 3x3 kernels");
    out->cin = cin;
    out->cin_pad = pad64 ? rup(cin, 64) : cin;
    out->cout = total;
    out->n_pad = pad64 ? rup(total, 64) : total;
    out->taps = taps;
    out->geglu = geglu;
    SDMI_REQUIRE(names.size() == 1 || (total == out->n_pad), "stacked outputs must be multiples of 64");
    const size_t wn = (size_t)out->n_pad * taps * out->cin_pad;
    TRY(dev_alloc(e, (void**)&out->w, wn * sizeof(half_t)));
    SDMI_CHECK_HIP(hipMemset(out->w, 0, wn * sizeof(half_t)));
    if (any_bias) {
        TRY(dev_alloc(e, (void**)&out->b, (size_t)out->n_pad * sizeof(float)));
        SDMI_CHECK_HIP(hipMemset(out->b, 0, (size_t)out->n_pad * sizeof(float)));
    }
    int row = 0;
    for (size_t i = 0; i < names.size(); ++i) {
        const RawTensor* w = find_raw(m, names[i] + ".weight");
        const int O = (int)w->shape[0];
        const int Opad = (names.size() == 1) ? out->n_pad : O;
        const int kh = taps == 9 ? 3 : 1;
        const int pflags = (geglu ? 1 : 0) | ((dup_in && 2 * cin <= out->cin_pad) ? 2 : 0);      // launch_pack_conv_weight's flag bits
        TRY(launch_pack_conv_weight(w->ptr, w->dtype, out->w + (size_t)row * taps * out->cin_pad, O, cin, kh, kh, Opad,
                                    out->cin_pad, pflags, 0));
        if (e->recording_unet_sites)
            e->unet_sites[names[i] + ".weight"].push_back({out->w + (size_t)row * taps * out->cin_pad, O, cin, kh, Opad,
                                                           out->cin_pad, pflags});
        const RawTensor* b = find_raw(m, names[i] + ".bias");
        if (b) TRY(launch_pack_bias(b->ptr, b->dtype, out->b + row, O, Opad, geglu ? 1 : 0, 0));
        if (b && e->recording_unet_sites) e->unet_vec_sites[names[i] + ".bias"].push_back({out->b + row, O, Opad, geglu ? 1 : 0});
        row += Opad;
    }
    return 0;
}
static int pack_one(sdmi_engine* e, const std::map<std::string, RawTensor>& m, const std::string& name, bool pad64,
                    bool geglu, ConvW* out, bool dup_in = false) {
    return pack_stack(e, m, {name}, pad64, geglu, out, dup_in);
}
static int pack_norm(sdmi_engine* e, const std::map<std::string, RawTensor>& m, const std::string& name, NormW* out) {
    const RawTensor* g = find_raw(m, name + ".weight");
    const RawTensor* b = find_raw(m, name + ".bias");
    SDMI_REQUIRE(g && b, "missing norm " + name);
    const int c = (int)g->shape[0];
    out->c = c;
    TRY(dev_alloc(e, (void**)&out->g, c * sizeof(float)));
    TRY(dev_alloc(e, (void**)&out->b, c * sizeof(float)));
    TRY(launch_convert_to_f32(g->ptr, g->dtype, out->g, c, 0));
    TRY(launch_convert_to_f32(b->ptr, b->dtype, out->b, c, 0));
    if (e->recording_unet_sites) {
        e->unet_vec_sites[name + ".weight"].push_back({out->g, c, c, 0});
        e->unet_vec_sites[name + ".bias"].push_back({out->b, c, c, 0});
    }
    return 0;
}

static int load_raw(std::map<std::string, RawTensor>& m, const char* key, const void* data, int dtype, int ndim,
                    const int64_t* shape, int on_device) {
    SDMI_REQUIRE(key && data && ndim >= 1 && ndim <= 4, "bad tensor arguments");
    SDMI_REQUIRE(dtype == SDMI_F16 || dtype == SDMI_F32, "dtype must be SDMI_F16 or SDMI_F32");
    RawTensor t;
    t.dtype = dtype;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.bytes = n * (dtype == SDMI_F16 ? 2 : 4);
    SDMI_CHECK_HIP(hipMalloc(&t.ptr, std::max<size_t>(t.bytes, 16)));
    SDMI_CHECK_HIP(hipMemcpy(t.ptr, data, t.bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    auto it = m.find(key);
    if (it != m.end()) { (void)hipFree(it->second.ptr); m.erase(it); }
    m.emplace(key, t);
    return 0;
}
static void free_raw(std::map<std::string, RawTensor>& m) {
    for (auto& kv : m) (void)hipFree(kv.second.ptr);
    m.clear();
}

// ------------------------------------------------------------------------------------------------------------
// UNet construction
// ------------------------------------------------------------------------------------------------------------
static int build_res(sdmi_engine* e, const std::string& name, int cin, int cout, bool vae, ResW* r) {
    auto& m = vae ? e->raw_vae : e->raw_unet;
    r->cin = cin; r->cout = cout;
    if (vae) {
        TRY(pack_norm(e, m, name + ".norm1", &r->n1));
        TRY(pack_one(e, m, name + ".conv1", true, false, &r->c1));
        TRY(pack_norm(e, m, name + ".norm2", &r->n2));
        TRY(pack_one(e, m, name + ".conv2", true, false, &r->c2));
        r->has_skip = cin != cout;
        if (r->has_skip) TRY(pack_one(e, m, name + ".nin_shortcut", true, false, &r->skip));
    } else {
        TRY(pack_norm(e, m, name + ".in_layers.0", &r->n1));
        TRY(pack_one(e, m, name + ".in_layers.2", true, false, &r->c1));
        TRY(pack_norm(e, m, name + ".out_layers.0", &r->n2));
        TRY(pack_one(e, m, name + ".out_layers.3", true, false, &r->c2));
        r->has_skip = cin != cout;
        if (r->has_skip) TRY(pack_one(e, m, name + ".skip_connection", true, false, &r->skip));
    }
    return 0;
}

static int build_st(sdmi_engine* e, const std::string& name, int ch, int heads, int dhead, int depth, STW* st) {
    auto& m = e->raw_unet;
    st->ch = ch; st->heads = heads; st->dhead = dhead;
    SDMI_REQUIRE(heads * dhead == ch, "inner_dim must equal channels");
    TRY(pack_norm(e, m, name + ".norm", &st->norm));
    TRY(pack_one(e, m, name + ".proj_in", true, false, &st->proj_in));
    TRY(pack_one(e, m, name + ".proj_out", true, false, &st->proj_out));
    for (int d = 0; d < depth; ++d) {
        const std::string tb = name + ".transformer_blocks." + std::to_string(d);
        TBlockW b;
        TRY(pack_norm(e, m, tb + ".norm1", &b.ln1));
        TRY(pack_norm(e, m, tb + ".norm2", &b.ln2));
        TRY(pack_norm(e, m, tb + ".norm3", &b.ln3));
        TRY(pack_stack(e, m, {tb + ".attn1.to_q", tb + ".attn1.to_k"}, true, false, &b.qk1));
        TRY(pack_one(e, m, tb + ".attn1.to_v", true, false, &b.v1));
        TRY(pack_one(e, m, tb + ".attn1.to_out.0", true, false, &b.o1));
        TRY(pack_one(e, m, tb + ".attn2.to_q", true, false, &b.q2));
        TRY(pack_one(e, m, tb + ".attn2.to_k", true, false, &b.k2));
        TRY(pack_one(e, m, tb + ".attn2.to_v", true, false, &b.v2));
        TRY(pack_one(e, m, tb + ".attn2.to_out.0", true, false, &b.o2));
        TRY(pack_one(e, m, tb + ".ff.net.0.proj", true, true, &b.ff1));
        TRY(pack_one(e, m, tb + ".ff.net.2", true, false, &b.ff2));
        // room for the fused feed-forward chain's packed operand stream (rowchain.hip; filled lazily, refilled after weight updates): taken
        // here, not inside a forward — a hipMalloc between two launches of a forward synchronises the device
        if (rowchain_supports(ch) && b.ff1.geglu && b.ff2.cin_pad * 2 == b.ff1.n_pad && b.ff1.cin_pad == ch && b.ff2.n_pad == ch) {
            const size_t pb = rowchain_ff_pack_bytes(ch, b.ff2.cin_pad);
            if (pb) TRY(dev_alloc(e, (void**)&b.ff_packs, pb));
        }
        b.ctx_slot = e->unet.n_ctx_slots++;
        st->blocks.push_back(b);
    }
    return 0;
}

static void heads_for(const sdmi_unet_config& c, int ch, int* nh, int* dh) {
    if (c.num_head_channels == -1) { *nh = c.num_heads; *dh = ch / c.num_heads; }
    else { *nh = ch / c.num_head_channels; *dh = c.num_head_channels; }
}

static int unet_build(sdmi_engine* e) {
    UNetW& u = e->unet;
    const sdmi_unet_config& c = u.cfg;
    auto& m = e->raw_unet;
    const int mc = c.model_channels;
    u.n_ctx_slots = 0;
    u.input.clear(); u.output.clear(); u.middle.clear();
    TRY(pack_one(e, m, "time_embed.0", false, false, &u.te0));
    TRY(pack_one(e, m, "time_embed.2", false, false, &u.te2));
    if (c.adm_in_channels > 0) {
        TRY(pack_one(e, m, "label_emb.0.0", false, false, &u.le0));
        TRY(pack_one(e, m, "label_emb.0.2", false, false, &u.le2));
    }
    std::vector<std::string> emb_names;
    int emb_cols = 0;
    auto add_res = [&](const std::string& name, int cin, int cout, int c0, int c1, std::vector<UNetLayer>* dst) -> int {
        UNetLayer L;
        L.kind = UNetLayer::RES;
        L.c0 = c0; L.c1 = c1;
        TRY(build_res(e, name, cin, cout, false, &L.res));
        L.res.emb_off = emb_cols;
        emb_cols += cout;
        emb_names.push_back(name + ".emb_layers.1");
        dst->push_back(L);
        return 0;
    };
    auto add_st = [&](const std::string& name, int ch, int level, std::vector<UNetLayer>* dst) -> int {
        UNetLayer L;
        L.kind = UNetLayer::ST;
        int nh, dh;
        heads_for(c, ch, &nh, &dh);
        TRY(build_st(e, name, ch, nh, dh, c.transformer_depth[level], &L.st));
        dst->push_back(L);
        return 0;
    };
    // input blocks
    {
        std::vector<UNetLayer> blk;
        UNetLayer L;
        L.kind = UNetLayer::CONV_IN;
        // (input channels repeated into the zero padding: the accuracy mode feeds the latent's lo half there — launch_nchw_to_nhwc)
        TRY(pack_one(e, m, "input_blocks.0.0", true, false, &L.conv, true));
        blk.push_back(L);
        u.input.push_back(blk);
    }
    std::vector<int> chans{mc};
    int ch = mc, idx = 1;
    const int nlev = c.num_levels;
    for (int level = 0; level < nlev; ++level) {
        for (int i = 0; i < c.num_res_blocks; ++i) {
            std::vector<UNetLayer> blk;
            const int cout = c.channel_mult[level] * mc;
            TRY(add_res("input_blocks." + std::to_string(idx) + ".0", ch, cout, ch, 0, &blk));
            ch = cout;
            if (c.attn_level[level]) TRY(add_st("input_blocks." + std::to_string(idx) + ".1", ch, level, &blk));
            u.input.push_back(blk);
            chans.push_back(ch);
            ++idx;
        }
        if (level != nlev - 1) {
            std::vector<UNetLayer> blk;
            UNetLayer L;
            L.kind = UNetLayer::DOWN;
            TRY(pack_one(e, m, "input_blocks." + std::to_string(idx) + ".0.op", true, false, &L.conv));
            blk.push_back(L);
            u.input.push_back(blk);
            chans.push_back(ch);
            ++idx;
        }
    }
    TRY(add_res("middle_block.0", ch, ch, ch, 0, &u.middle));
    TRY(add_st("middle_block.1", ch, nlev - 1, &u.middle));
    TRY(add_res("middle_block.2", ch, ch, ch, 0, &u.middle));
    idx = 0;
    for (int level = nlev - 1; level >= 0; --level) {
        for (int i = 0; i <= c.num_res_blocks; ++i) {
            std::vector<UNetLayer> blk;
            const int ich = chans.back();
            chans.pop_back();
            const int cout = mc * c.channel_mult[level];
            const std::string base = "output_blocks." + std::to_string(idx);
            TRY(add_res(base + ".0", ch + ich, cout, ch, ich, &blk));
            ch = cout;
            int li = 1;
            if (c.attn_level[level]) { TRY(add_st(base + "." + std::to_string(li), ch, level, &blk)); ++li; }
            if (level && i == c.num_res_blocks) {
                UNetLayer L;
                L.kind = UNetLayer::UP;
                TRY(pack_one(e, m, base + "." + std::to_string(li) + ".conv", true, false, &L.conv));
                blk.push_back(L);
            }
            u.output.push_back(blk);
            ++idx;
        }
    }
    TRY(pack_norm(e, m, "out.0", &u.out_norm));
    TRY(pack_one(e, m, "out.2", true, false, &u.out_conv));
    // fused emb projection: stack emb_layers.1 of every ResBlock -> [sum Cout][ted]
    TRY(pack_stack(e, m, emb_names, false, false, &u.emb_all));
    u.emb_cols = emb_cols;
    SDMI_CHECK_HIP(hipDeviceSynchronize());
    u.ready = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------------------
struct ConvArgs {
    const half_t* a0 = nullptr;
    const half_t* a1 = nullptr;
    int c0 = 0, c1 = 0;
    int B = 1, Hi = 1, Wi = 1, Ho = 1, Wo = 1;
    int stride = 1, pad = 0, up = 0;
    const float* rowbias = nullptr;
    int ldrb = 0;
    const half_t* resid = nullptr;
    const half_t* resid_lo = nullptr;   // (hi, lo) stream tensors: GemmP::resid_lo / out_lo
    half_t* out_lo = nullptr;
    int ldr = 0;
    void* out = nullptr;
    int ldo = 0;
    int flags = 0;
    int n_real = 0;
    float alpha = 1.f;
    float bias_scale = 1.f;
    int batch = 1;
    long a_bs = 0, w_bs = 0, o_bs = 0, r_bs = 0;
    const int* gate = nullptr;
    int stats_C = 0;              // > 0: the output feeds a GroupNorm(32) over stats_C channels: emit its partial sums from the epilogue
    bool no_split = false;        // never take a split-K workspace from the arena (callers outside a sized forward pass)
    float* stats_ws_pre = nullptr;    // arena_reuse: room for the GroupNorm partial sums taken by the caller (it must outlive the caller's rewind)
};

static int run_conv(Run& r, const ConvW& W, const ConvArgs& a) {
    // split-K slabs (only deep / small-M layers qualify); allocated in the dry pass too so the arena layout is identical
    float* splitk_ws = nullptr;
    if (!(a.flags & EP_NCHW) && !W.geglu && !a.no_split) {
        const size_t wsb = gemm_splitk_ws_bytes(a.B * a.Ho * a.Wo, W.n_pad, W.taps * (a.c0 + a.c1), a.batch);
        if (wsb) splitk_ws = r.F(wsb / sizeof(float));
    }
    // room for the GroupNorm partial sums of the output ([B][<= 64 chunks][32 groups][2]); allocated in the dry pass too
    float* stats_ws = nullptr;
    if (a.stats_C > 0 && W.n_pad == a.stats_C && a.stats_C % 32 == 0) stats_ws = a.stats_ws_pre ? a.stats_ws_pre : r.F((size_t)a.B * 64 * 32 * 2);
    r.st_tensor = nullptr; r.st_nchunk = 0;
    // room for the LayerNorm row partials of the output ([M][N / 64 tiles at most][2]); allocated in the dry pass too
    float* lnp_ws = nullptr;
    const bool want_lnp = r.lnp_want && W.taps == 1 && !W.geglu && !(a.flags & (EP_NCHW | EP_OUT_F32 | EP_TRANSPOSE));
    r.lnp_want = false;
    if (want_lnp) lnp_ws = r.F((size_t)a.B * a.Ho * a.Wo * (W.n_pad / 64) * 2);
    r.lnp_tensor = nullptr; r.lnp_np = 0;
    if (r.dry) return 0;
    GemmP p{};
    p.splitk_ws = splitk_ws;
    p.lnp_out = lnp_ws;
    p.a0 = a.a0; p.a1 = a.a1; p.w = W.w; p.bias = W.b; p.rowbias = a.rowbias; p.resid = a.resid; p.out = a.out;
    p.c0 = a.c0; p.c1 = a.c1; p.cin = a.c0 + a.c1; p.lda0 = a.c0; p.lda1 = a.c1;
    SDMI_REQUIRE(p.cin == W.cin_pad, "conv input channels do not match the packed weight");
    p.Hi = a.Hi; p.Wi = a.Wi; p.Ho = a.Ho; p.Wo = a.Wo;
    p.taps = W.taps; p.stride = a.stride; p.pad = a.pad; p.up = a.up;
    p.M = a.B * a.Ho * a.Wo; p.N = W.n_pad; p.K = W.taps * p.cin;
    p.ldo = a.ldo; p.ldr = a.ldr; p.ldw = p.K; p.ldrb = a.ldrb;
    p.rows_per_batch = a.Ho * a.Wo;
    p.n_real = a.n_real ? a.n_real : W.n_pad;
    p.flags = a.flags | (W.geglu ? EP_GEGLU : 0);
    if (a.out_lo || a.resid_lo) {                          // (hi, lo) stream tensors ride in the LayerNorm-fold consumer's fields (GemmP, EP_HILO)
        p.flags |= EP_HILO;
        p.ln_stats = reinterpret_cast<const float*>(a.out_lo);
        p.ln_s = reinterpret_cast<const float*>(a.resid_lo);
    }
    if (r.e->tiling && W.taps == 9 && a.pad == 1) p.flags |= EP_WRAP;    // Conv2d(padding=1, padding_mode='circular')
    p.alpha = a.alpha;
    p.bias_scale = a.bias_scale;
    p.a_bs = a.a_bs; p.w_bs = a.w_bs; p.o_bs = a.o_bs; p.r_bs = a.r_bs;
    p.gate = a.gate;
    int nchunk = 0;
    if (stats_ws) { p.stats_out = stats_ws; p.stats_cpg = a.stats_C / 32; }
    int lnp_np = 0;
    TRY(launch_gemm(p, a.batch, r.e->force_generic, r.e->use_glds, r.s, &nchunk, &lnp_np));
    if (nchunk > 0) { r.st_tensor = (const half_t*)a.out; r.st_ws = stats_ws; r.st_nchunk = nchunk; }
    if (lnp_np > 0) { r.lnp_tensor = (const half_t*)a.out; r.lnp_ws = lnp_ws; r.lnp_np = lnp_np; }
    return 0;
}

// plain [rows, K] x W^T GEMM on token matrices
static int run_linear(Run& r, const ConvW& W, const half_t* a, int rows, const half_t* resid, half_t* out, int ldo, float ss = 1.f,
                      bool no_split = false, const half_t* resid_lo = nullptr, half_t* out_lo = nullptr) {
    ConvArgs c;
    c.resid_lo = resid_lo; c.out_lo = out_lo;
    c.alpha = ss; c.bias_scale = ss;
    c.no_split = no_split;
    c.a0 = a; c.c0 = W.cin_pad;
    c.B = 1; c.Hi = rows; c.Wi = 1; c.Ho = rows; c.Wo = 1;
    c.resid = resid; c.ldr = ldo; c.out = out; c.ldo = ldo;
    return run_conv(r, W, c);
}

static int run_gn(Run& r, const NormW& n, const half_t* x0, const half_t* x1, int c0, int c1, int B, int HW, float eps,
                  bool silu, half_t* out, const half_t* x0_lo = nullptr, const half_t* x1_lo = nullptr) {
    float* ws = r.F(groupnorm_ws_bytes(B, HW, 32) / sizeof(float));
    if (r.dry) return 0;
    SDMI_REQUIRE(n.c == c0 + c1, "GroupNorm channel mismatch");
    if (x1 == nullptr && r.st_nchunk > 0 && r.st_tensor == x0)      // the producing GEMM already summed this tensor (hi + lo of a pair)
        return launch_groupnorm(x0, nullptr, c0, 0, n.g, n.b, out, B, HW, 32, eps, silu, r.st_ws, r.s, r.st_nchunk, x0_lo, nullptr);
    return launch_groupnorm(x0, x1, c0, c1, n.g, n.b, out, B, HW, 32, eps, silu, ws, r.s, 0, x0_lo, x1_lo);
}
static int run_ln(Run& r, const NormW& n, const half_t* x, int64_t rows, half_t* out, const half_t* x_lo = nullptr) {
    if (r.dry) return 0;
    return launch_layernorm(x, n.g, n.b, out, rows, n.c, 1e-5f, r.s, x_lo);
}

// ---- LayerNorm folded into the consuming GEMM (option "ln_fold") -----------------------------------------------------------
// Row statistics of a LayerNorm input: the producer's per-tile partial sums when the GEMM that wrote x left them (np > 0), else
// (mean, rstd) from the row-statistics kernel (np == 0).  The arena slot of the kernel's output is taken in both cases, so the
// arena layout does not depend on which producer tile was chosen.
struct LnStats { const float* ptr = nullptr; int np = 0; };
static int run_ln_stats(Run& r, const NormW& n, const half_t* x, int64_t rows, LnStats* out) {
    float* own = r.F((size_t)rows * 2);
    if (r.dry) { out->ptr = own; out->np = 0; return 0; }
    if (r.lnp_tensor == x && r.lnp_np > 0) { out->ptr = r.lnp_ws; out->np = r.lnp_np; return 0; }
    out->ptr = own; out->np = 0;
    return launch_ln_rowstats(x, own, rows, n.c, 1e-5f, r.s);
}
// folded copy of W for the LayerNorm `n` in front of it; persistent, rebuilt after weight updates
static int ensure_ln_fold(sdmi_engine* e, const ConvW& W, const NormW& n, hipStream_t s) {
    if (W.w_ln && W.fold_epoch == e->weights_epoch) return 0;
    SDMI_REQUIRE(W.taps == 1 && n.c <= W.cin_pad, "LayerNorm fold: linear layers only");
    if (!W.w_ln) {
        void* p = nullptr;
        SDMI_CHECK_HIP(hipMalloc(&p, (size_t)W.n_pad * W.cin_pad * sizeof(half_t))); e->owned.push_back(p); W.w_ln = (half_t*)p;
        SDMI_CHECK_HIP(hipMalloc(&p, (size_t)W.n_pad * sizeof(float))); e->owned.push_back(p); W.s_ln = (float*)p;
        SDMI_CHECK_HIP(hipMalloc(&p, (size_t)W.n_pad * sizeof(float))); e->owned.push_back(p); W.c_ln = (float*)p;
    }
    TRY(launch_ln_fold_weights(W.w, n.g, n.b, W.b, W.w_ln, W.s_ln, W.c_ln, W.n_pad, W.cin_pad, n.c, s));
    W.fold_epoch = e->weights_epoch;
    return 0;
}
// out = LN_n(x) W^T + b through the folded weights: x is the UN-normalised input, `stats` its row statistics
static int run_linear_ln(Run& r, const ConvW& W, const NormW& n, const half_t* x, const LnStats& stats, int rows, half_t* out, int ldo) {
    if (!r.dry) TRY(ensure_ln_fold(r.e, W, n, r.s));
    // same split-K workspace bookkeeping as run_conv (dry pass included): the arena layout must not depend on the option's timing
    float* splitk_ws = nullptr;
    if (!W.geglu) {
        const size_t wsb = gemm_splitk_ws_bytes(rows, W.n_pad, W.cin_pad, 1);
        if (wsb) splitk_ws = r.F(wsb / sizeof(float));
    }
    r.st_tensor = nullptr; r.st_nchunk = 0;
    if (r.dry) return 0;
    GemmP p{};
    p.splitk_ws = splitk_ws;
    p.a0 = x; p.w = W.w_ln; p.bias = W.c_ln; p.out = out;
    p.c0 = W.cin_pad; p.cin = W.cin_pad; p.lda0 = W.cin_pad;
    p.Hi = rows; p.Wi = 1; p.Ho = rows; p.Wo = 1;
    p.taps = 1; p.stride = 1; p.pad = 0; p.up = 0;
    p.M = rows; p.N = W.n_pad; p.K = W.cin_pad;
    p.ldo = ldo; p.ldw = p.K;
    p.rows_per_batch = rows; p.n_real = W.n_pad;
    p.flags = EP_LNFOLD | (W.geglu ? EP_GEGLU : 0);
    p.alpha = 1.f; p.bias_scale = 1.f;
    p.ln_stats = stats.ptr; p.ln_np = stats.np; p.ln_inv_c = 1.0f / (float)n.c; p.ln_eps = 1e-5f; p.ln_s = W.s_ln;
    return launch_gemm(p, 1, false, true, r.s);
}
// V^T [B][C][tokens_pad] = (LN_n(x) Wv^T + bv)^T through the folded weights (the token-major EP_TRANSPOSE form of run_vt)
static int run_vt_ln(Run& r, const ConvW& Wv, const NormW& n, const half_t* x, const LnStats& stats, int B, int tokens, int tokens_pad,
                     half_t* vt) {
    if (r.dry) return 0;
    TRY(ensure_ln_fold(r.e, Wv, n, r.s));
    GemmP p{};
    const int C = Wv.n_pad, K = Wv.cin_pad;
    p.a0 = x; p.c0 = K; p.cin = K; p.lda0 = K;
    p.w = Wv.w_ln; p.ldw = K;
    p.bias = Wv.c_ln;                                    // beta . Wv^T (+ bv): per output channel
    p.out = vt;
    p.Hi = tokens; p.Wi = 1; p.Ho = tokens; p.Wo = 1;
    p.taps = 1; p.stride = 1; p.pad = 0; p.up = 0;
    p.M = B * tokens; p.N = C; p.K = K; p.n_valid = C;
    p.ldo = tokens_pad; p.rows_per_batch = tokens; p.n_real = C;
    p.flags = EP_TRANSPOSE | EP_LNFOLD;
    p.alpha = 1.f; p.bias_scale = 1.f;
    p.ln_stats = stats.ptr; p.ln_np = stats.np; p.ln_inv_c = 1.0f / (float)n.c; p.ln_eps = 1e-5f; p.ln_s = Wv.s_ln;
    return launch_gemm(p, 1, false, true, r.s);
}

// ResBlock (UNet: GroupNorm32 eps 1e-5 + emb add; VAE: eps 1e-6, no emb).  Returns the output buffer.
// `ss` (VAE range-extended decode only): the residual stream — x0 / x1 in, *out out — is stored multiplied by ss; GroupNorm is
// scale-invariant once eps is multiplied by ss^2, the block-internal tensors stay at true scale, and the two layers that write the
// stream scale their accumulator (alpha) and / or bias (bias_scale).
// `hilo` (the UNet's calls under option "residual_fp32"): x0 / x1 and the output are (hi, lo) pairs of the carried stream, the lo half
// `lo_rows` rows behind the hi half (0: the B * H * W rows of this call; unet_run's shared CFG prefix computes B = Bn / 2 rows of
// buffers sized for Bn) — and so is the first conv's output, which only GroupNorm reads (every tensor that is not
// a matrix-core operand keeps ~22 bits).
static int run_res(Run& r, const ResW& w, const half_t* x0, const half_t* x1, int c0, int c1, int B, int H, int Wd,
                   float eps, const float* embs, int emb_ld, half_t** out, float ss = 1.f, half_t* out_buf = nullptr, bool hilo = false,
                   size_t lo_rows = 0) {
    const int HW = H * Wd;
    const size_t M = (size_t)B * HW;
    const size_t LR = lo_rows ? lo_rows : M;
    // arena_reuse: what outlives the block — its output and the GroupNorm partial sums the last conv leaves for the next norm — is taken
    // first, everything after the mark is released on return
    const bool reuse = r.reuse();
    half_t* o_pre = reuse ? r.H(M * w.cout) : nullptr;
    float* st_pre = (reuse && ss == 1.f && w.cout % 32 == 0) ? r.F((size_t)B * 64 * 32 * 2) : nullptr;
    const size_t mk = r.ar->mark();
    // option "residual_fp32" (UNet only, ss == 1): x0 / x1 and the output are (hi, lo) pairs of the carried stream
    const bool acc = hilo && r.acc() && ss == 1.f;                           // (hilo: the UNet's calls; the VAE's tensors are plain fp16)
    const half_t* x0_lo = acc ? r.lo(x0, LR * c0) : nullptr;
    const half_t* x1_lo = (acc && x1) ? r.lo(x1, LR * c1) : nullptr;
    half_t* t1 = r.H(M * w.cin);
    TRY(run_gn(r, w.n1, x0, x1, c0, c1, B, HW, eps * ss * ss, true, t1, x0_lo, x1_lo));
    half_t* h1 = acc ? r.S(M * w.cout) : r.H(M * w.cout);
    {
        ConvArgs c;
        c.a0 = t1; c.c0 = w.cin; c.B = B; c.Hi = H; c.Wi = Wd; c.Ho = H; c.Wo = Wd; c.pad = 1;
        if (embs) { c.rowbias = embs + w.emb_off; c.ldrb = emb_ld; }
        c.out = h1; c.ldo = w.cout;
        c.out_lo = acc ? r.lo(h1, M * w.cout) : nullptr;
        c.stats_C = w.cout;                                   // h1 is read by norm2 only
        TRY(run_conv(r, w.c1, c));
    }
    half_t* t2 = r.H(M * w.cout);
    TRY(run_gn(r, w.n2, h1, nullptr, w.cout, 0, B, HW, eps, true, t2, acc ? r.lo(h1, M * w.cout) : nullptr));
    const half_t* resid = x0;
    const half_t* resid_lo = x0_lo;
    if (w.has_skip) {
        half_t* sk = acc ? r.S(M * w.cout) : r.H(M * w.cout);
        ConvArgs c;
        c.a0 = x0; c.a1 = x1; c.c0 = c0; c.c1 = c1; c.B = B; c.Hi = H; c.Wi = Wd; c.Ho = H; c.Wo = Wd;
        c.out = sk; c.ldo = w.cout;
        c.resid_lo = nullptr;
        c.out_lo = acc ? r.lo(sk, M * w.cout) : nullptr;      // the skip_connection output is part of the stream
        c.bias_scale = ss;                                    // the input already carries ss
        TRY(run_conv(r, w.skip, c));
        resid = sk;
        resid_lo = c.out_lo;
    } else {
        SDMI_REQUIRE(x1 == nullptr, "identity skip with a concatenated input");
    }
    half_t* o = out_buf ? out_buf : (reuse ? o_pre : (acc ? r.S(M * w.cout) : r.H(M * w.cout)));       // (out_buf: the caller's, larger buffer — unet_run's shared CFG prefix)
    {
        ConvArgs c;
        c.a0 = t2; c.c0 = w.cout; c.B = B; c.Hi = H; c.Wi = Wd; c.Ho = H; c.Wo = Wd; c.pad = 1;
        c.resid = resid; c.ldr = w.cout; c.out = o; c.ldo = w.cout;
        c.resid_lo = resid_lo; c.out_lo = acc ? r.lo(o, (out_buf ? LR : M) * w.cout) : nullptr;
        c.alpha = ss; c.bias_scale = ss;
        c.stats_C = ss == 1.f ? w.cout : 0;                   // the block output usually feeds the next GroupNorm (ignored if not)
        c.stats_ws_pre = st_pre;
        TRY(run_conv(r, w.c2, c));
    }
    if (reuse) r.ar->rewind(mk);
    *out = o;
    return 0;
}

static int run_attn(Run& r, const half_t* q, const half_t* k, const half_t* vt, half_t* out, int B, int H, int N, int M,
                    int D, int ldq, int ldk, int vt_ld, int ldo) {
    if (r.dry) return 0;
    AttnP p{};
    p.q = q; p.k = k; p.vt = vt; p.out = out;
    p.B = B; p.H = H; p.N = N; p.M = M; p.D = D;
    p.ldq = ldq; p.ldk = ldk; p.vt_ld = vt_ld; p.ldo = ldo;
    p.scale_log2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
    return launch_attention(p, r.e->force_generic, r.s);
}

// V^T[b] = Wv x X[b]^T :  the weight matrix plays the "activation" role, the tokens of image b the "weight" role.
// Columns [tokens, tokens_pad) of V^T are written as zeros (n_valid), so the attention kernel's padded keys are finite.
static int run_vt(Run& r, const ConvW& Wv, const half_t* x, int ldx, int B, int tokens, int tokens_pad, half_t* vt,
                  bool bias_row, const int* gate = nullptr) {
    if (r.dry) return 0;
    GemmP p{};
    const int C = Wv.n_pad, K = Wv.cin_pad;
    if (g_vt_mode == 1 && tokens == tokens_pad && tokens % 4 == 0 && ldx == K && gate == nullptr) {
        // token-major tiles (M = B*tokens rows: full 256 / 128-row tiles whatever C is) with the MFMA operand roles swapped, so the
        // epilogue stores V^T[b][c][token] in 8-byte pieces along the token dimension (EP_TRANSPOSE).  Same products, same K order
        // as the ordinary projection.  (The weights-as-rows form below pads M = C = 320 up to 512 tile rows.)
        p.a0 = x; p.c0 = K; p.cin = K; p.lda0 = ldx;
        p.w = Wv.w; p.ldw = K;
        p.bias = bias_row ? Wv.b : nullptr;                  // per output channel n
        p.out = vt;
        p.Hi = tokens; p.Wi = 1; p.Ho = tokens; p.Wo = 1;      // B "images" of tokens x 1 pixels: the gather's image stride
        p.taps = 1; p.stride = 1; p.pad = 0; p.up = 0;
        p.M = B * tokens; p.N = C; p.K = K; p.n_valid = C;
        p.ldo = tokens_pad; p.rows_per_batch = tokens; p.n_real = C;
        p.flags = EP_TRANSPOSE;
        p.alpha = 1.f;
        return launch_gemm(p, 1, r.e->force_generic, r.e->use_glds, r.s);
    }
    p.a0 = Wv.w; p.c0 = K; p.cin = K; p.lda0 = K;
    p.w = x; p.ldw = ldx;
    p.bias = bias_row ? Wv.b : nullptr;
    p.out = vt;
    p.Hi = C; p.Wi = 1; p.Ho = C; p.Wo = 1;
    p.taps = 1; p.stride = 1; p.pad = 0; p.up = 0;
    p.M = C; p.N = tokens_pad; p.K = K; p.n_valid = tokens;
    p.ldo = tokens_pad; p.rows_per_batch = C; p.n_real = tokens_pad;
    p.flags = bias_row ? EP_BIAS_ROW : 0;
    p.alpha = 1.f;
    p.a_bs = 0; p.w_bs = (long)tokens * ldx;
    p.o_bs = (long)C * tokens_pad;
    p.gate = gate;
    return launch_gemm(p, B, r.e->force_generic, r.e->use_glds, r.s);
}

// ------------------------------------------------------------------------------------------------------------
// hypernetworks: context_k / context_v = x + multiplier * MLP(x), chained over the loaded networks (hypernetwork.py:358-379)
// ------------------------------------------------------------------------------------------------------------
static bool hn_has_dim(const sdmi_engine* e, int dim) {
    for (const auto& n : e->hypernets)
        if (n.by_dim.count(dim)) return true;
    return false;
}

// `alloc(n)` hands out n halfs (the forward's arena, or the persistent context scratch).  Returns x itself when no loaded network
// has modules for this width.  Intermediates stay fp16 (the reference runs the small MLPs in fp32 and casts back, :366-367).
template <typename Alloc>
static int run_hn(Run& r, int dim, int which, const half_t* x, size_t rows, Alloc alloc, bool no_split, const half_t** out) {
    const half_t* cur = x;
    for (const HnNet& net : r.e->hypernets) {
        auto it = net.by_dim.find(dim);
        if (it == net.by_dim.end()) continue;
        const HnModule& mod = which ? it->second.second : it->second.first;
        const half_t* h = cur;
        int width = dim;
        half_t* y = nullptr;
        for (size_t i = 0; i < mod.ops.size(); ++i) {
            const HnOp& op = mod.ops[i];
            const bool last = i + 1 == mod.ops.size();
            if (op.kind == 0) {
                SDMI_REQUIRE(op.lin.cin_pad == width || r.dry, "hypernetwork layer widths do not chain");
                half_t* t = alloc(rows * (size_t)op.lin.n_pad);
                if (last) {                                   // x + multiplier * (h W^T + b): folded into the GEMM epilogue
                    SDMI_REQUIRE(op.lin.n_pad == dim, "hypernetwork output width must equal its input width");
                    TRY(run_linear(r, op.lin, h, (int)rows, cur, t, op.lin.n_pad, net.multiplier, no_split));
                    y = t;
                } else {
                    TRY(run_linear(r, op.lin, h, (int)rows, nullptr, t, op.lin.n_pad, 1.f, no_split));
                }
                h = t;
                width = op.lin.n_pad;
            } else if (op.kind == 1) {
                if (!r.dry) TRY(launch_act_f16(const_cast<half_t*>(h), (int64_t)rows * width, op.act, r.s));   // h is never x itself here
            } else {
                half_t* t = alloc(rows * (size_t)width);
                if (!r.dry) TRY(launch_layernorm(h, op.ln.g, op.ln.b, t, (int64_t)rows, width, 1e-5f, r.s));
                h = t;
            }
        }
        if (y == nullptr) {                                    // the module ends in an activation / LayerNorm: separate x + m * h
            SDMI_REQUIRE(width == dim, "hypernetwork output width must equal its input width");
            y = alloc(rows * (size_t)dim);
            if (!r.dry) TRY(launch_axpy_f16(y, cur, h, net.multiplier, (int64_t)rows * dim, r.s));
        }
        cur = y;
    }
    *out = cur;
    return 0;
}

// Bh > 0 (unet_run's shared CFG prefix, engine option "cfg_pairs"): rows [Bh, B) of x repeat rows [0, Bh) and differ from them only in
// their text context, so everything in front of the first cross-attention — GroupNorm, proj_in, and norm1 / q, k, v / self-attention /
// out-projection of the first block — is computed for Bh rows and copied to the other half; x itself holds all B rows (proj_out's residual).
static int run_st(Run& r, const STW& st, const half_t* x, int B, int H, int Wd, int L, half_t** out,
                  const std::string& name = std::string(), int Bh = 0) {
    sdmi_engine* e = r.e;
    const int C = st.ch, HW = H * Wd;
    const size_t M = (size_t)B * HW;
    int B1 = Bh > 0 ? Bh : B;                            // rows of the part in front of the first cross-attention
    size_t M1 = (size_t)B1 * HW;
    const int Npad = rup(HW, 64);
    // arena_reuse: the output, the token stream after proj_in and two ping-pong buffers for the transformer blocks' outputs are taken
    // first; everything after the mark lives for one block (or for proj_in) only
    const bool reuse = r.reuse();
    half_t* o_pre = reuse ? r.H(M * C) : nullptr;
    half_t* cur_pre = reuse ? r.H(M * C) : nullptr;
    half_t* pp[2] = {reuse ? r.H(M * C) : nullptr, (reuse && st.blocks.size() > 1) ? r.H(M * C) : nullptr};
    const size_t mk = r.ar->mark();
    // option "residual_fp32": x, the token stream and the output are (hi, lo) pairs (Run::S / Run::lo); the fused chains and the
    // LayerNorm fold are not combined with it
    const bool acc = r.acc();
    const size_t MC = M * C;
    half_t* n0 = r.H(M * C);
    TRY(run_gn(r, st.norm, x, nullptr, C, 0, B1, HW, 1e-6f, false, n0, r.lo(x, MC)));
    half_t* cur = reuse ? cur_pre : r.S(M * C);
    // with "ln_fold" the GEMMs that write a LayerNorm's input also leave its row sums (Run::lnp_want; a no-op otherwise)
    const bool fold_any = e->ln_fold && !acc && !hn_has_dim(e, C) && !e->force_generic && e->use_glds && g_vt_mode == 1 && HW == Npad && HW % 4 == 0;
    r.lnp_want = fold_any && e->ln_fold >= 2;
    TRY(run_linear(r, st.proj_in, n0, (int)M1, nullptr, cur, C, 1.f, false, nullptr, r.lo(cur, MC)));
    if (reuse) r.ar->rewind(mk);
    int bi = 0, blk = 0;
    for (const TBlockW& b : st.blocks) {
        const std::string bname = name + ".transformer_blocks." + std::to_string(bi++);
        // --- self attention
        // option "ln_fold": the three LayerNorms are finished inside the GEMMs that read them (folded weights + per-row statistics):
        // not with hypernetworks (they transform the normalised tokens) nor while block outputs are being tapped or cross-checked
        // on the generic kernels, and only when V^T takes the token-major transposed form (run_vt)
        const bool fold = fold_any;
        half_t* a1 = nullptr;
        if (fold) {
            LnStats st1;
            TRY(run_ln_stats(r, b.ln1, cur, M, &st1));
            half_t* qk = r.H(M * 2 * C);
            TRY(run_linear_ln(r, b.qk1, b.ln1, cur, st1, (int)M, qk, 2 * C));
            half_t* vt = r.H((size_t)B * C * Npad);
            TRY(run_vt_ln(r, b.v1, b.ln1, cur, st1, B, HW, Npad, vt));
            a1 = r.H(M * C);
            TRY(run_attn(r, qk, qk + C, vt, a1, B, st.heads, HW, HW, st.dhead, 2 * C, 2 * C, Npad, C));
        }
        half_t* n1 = fold ? nullptr : r.H(M * C);
        if (!fold) TRY(run_ln(r, b.ln1, cur, M1, n1, r.lo(cur, MC)));
        if (fold) {
        } else if (!hn_has_dim(e, C)) {
            half_t* qk = r.H(M * 2 * C);
            TRY(run_linear(r, b.qk1, n1, (int)M1, nullptr, qk, 2 * C));
            half_t* vt = r.H((size_t)B * C * Npad);
            TRY(run_vt(r, b.v1, n1, C, B1, HW, Npad, vt, false));
            a1 = r.H(M * C);
            TRY(run_attn(r, qk, qk + C, vt, a1, B1, st.heads, HW, HW, st.dhead, 2 * C, 2 * C, Npad, C));
        } else {
            // hypernetworks loaded for this width: K and V are projected from their own transformed copies of the context (= n1),
            // so the stacked q|k GEMM splits into its two halves (views into the same packed weight)
            auto alloc = [&](size_t n) { return r.H(n); };
            const half_t *xk = nullptr, *xv = nullptr;
            TRY(run_hn(r, C, 0, n1, M, alloc, false, &xk));
            TRY(run_hn(r, C, 1, n1, M, alloc, false, &xv));
            ConvW wq = b.qk1, wk = b.qk1;
            wq.n_pad = wk.n_pad = C; wq.cout = wk.cout = C;
            wk.w = b.qk1.w + (size_t)C * b.qk1.cin_pad;
            if (b.qk1.b) wk.b = b.qk1.b + C;
            half_t* q = r.H(M * C);
            half_t* k = r.H(M * C);
            TRY(run_linear(r, wq, n1, (int)M, nullptr, q, C));
            TRY(run_linear(r, wk, xk, (int)M, nullptr, k, C));
            half_t* vt = r.H((size_t)B * C * Npad);
            TRY(run_vt(r, b.v1, xv, C, B, HW, Npad, vt, false));
            a1 = r.H(M * C);
            TRY(run_attn(r, q, k, vt, a1, B, st.heads, HW, HW, st.dhead, C, C, Npad, C));
        }
        half_t* x1 = r.S(M * C);
        r.lnp_want = fold && e->ln_fold >= 2;
        TRY(run_linear(r, b.o1, a1, (int)M1, cur, x1, C, 1.f, false, r.lo(cur, MC), r.lo(x1, MC)));
        if (B1 != B) {                                       // end of the shared prefix: the other half of the batch continues from a copy
            if (!r.dry) SDMI_CHECK_HIP(hipMemcpyAsync(x1 + M1 * C, x1, M1 * C * sizeof(half_t), hipMemcpyDeviceToDevice, r.s));
            if (!r.dry && acc) SDMI_CHECK_HIP(hipMemcpyAsync(x1 + MC + M1 * C, x1 + MC, M1 * C * sizeof(half_t), hipMemcpyDeviceToDevice, r.s));
            B1 = B; M1 = M;
        }
        r.tap(bname + ".attn1+x", x1, B, H, Wd, C);
        // --- cross attention (K / V^T of the context are cached per layer by set_context)
        // option "fuse_rows" bit 1 (below): the feed-forward third of the block as one launch (rowchain.hip)
        // (not with option "streams" > 1: the feed-forward pack is built lazily on the stream of the slice that meets it first, and the
        // other slices would read it with no event dependency on that stream — the same rule as ln_fold's folded weights)
        // (not with (hi, lo) token streams: a pair form of the chain was built and measured in round 6 — the lo rows passing through the
        // wave's one LDS staging region after the hi rows, in and out: 324 us against 269 us for LayerNorm + GEGLU GEMM + GEMM on the pair,
        // profiles/r06_fwd_ab_accuracy_with_hilo_chain.txt — and removed again)
        const bool chain_ok = !fold && !acc && !e->force_generic && e->use_glds && rowchain_supports(C) && HW % 128 == 0 && e->n_streams <= 1;
        half_t* x2 = nullptr;
        {
        half_t* q2 = nullptr;
        if (fold) {
            LnStats st2;
            TRY(run_ln_stats(r, b.ln2, x1, M, &st2));
            q2 = r.H(M * C);
            TRY(run_linear_ln(r, b.q2, b.ln2, x1, st2, (int)M, q2, C));
        } else {
            half_t* n2 = r.H(M * C);
            TRY(run_ln(r, b.ln2, x1, M, n2, r.lo(x1, MC)));
            q2 = r.H(M * C);
            TRY(run_linear(r, b.q2, n2, (int)M, nullptr, q2, C));
        }
        half_t* a2 = r.H(M * C);
        if (!r.dry) {
            SDMI_REQUIRE(e->ctx_valid && e->ctx_B == (r.Btot ? r.Btot : B), "context not set for this batch size");
            // K rows of image b start at b*Lpad: express through ldk and a per-batch offset = Lpad*C
            AttnP p{};
            p.q = q2; p.out = a2;
            p.k = e->ctx_k[b.ctx_slot] + (size_t)r.b0 * e->ctx_L * C;          // (a batch slice starts at its own rows of the cache)
            p.vt = e->ctx_vt[b.ctx_slot] + (size_t)r.b0 * C * e->ctx_Lpad;
            p.B = B; p.H = st.heads; p.N = HW; p.M = e->ctx_L; p.D = st.dhead;
            p.ldq = C; p.ldk = C; p.vt_ld = e->ctx_Lpad; p.ldo = C;
            p.scale_log2 = (1.0f / sqrtf((float)st.dhead)) * 1.4426950408889634f;
            // the attention kernel addresses K as k + b*M*ldk; the cache is laid out with Lpad rows per image, so the
            // cache stores K compactly with exactly L rows per image (see set_context)
            TRY(launch_attention(p, e->force_generic, r.s));
        }
        x2 = r.S(M * C);
        r.lnp_want = fold && e->ln_fold >= 2;
        TRY(run_linear(r, b.o2, a2, (int)M, x1, x2, C, 1.f, false, r.lo(x1, MC), r.lo(x2, MC)));
        }
        r.tap(bname + ".attn2+x", x2, B, H, Wd, C);
        // --- feed forward (GEGLU fused in the first GEMM's epilogue)
        half_t* x3 = reuse ? pp[blk & 1] : r.S(M * C);                          // (block k reads pp[(k - 1) & 1] — or proj_in's buffer — and writes pp[k & 1])
        // option "fuse_rows" bit 1: norm3 -> ff.net.0.proj (GEGLU) -> ff.net.2 -> + x2 as one launch: the 4C-wide hidden tensor stays on the CU
        if (chain_ok && (e->fuse_rows & 2) && b.ff1.geglu && b.ff1.n_pad % 64 == 0 && b.ff2.cin_pad * 2 == b.ff1.n_pad && b.ff1.cin_pad == C &&
            b.ff2.n_pad == C) {
            if (!r.dry) {
                const int hidden = b.ff2.cin_pad;
                if (!b.ff_packs) {                            // (normally taken at build time: build_st; a model built before the option was set)
                    void* pk = nullptr;
                    SDMI_CHECK_HIP(hipMalloc(&pk, rowchain_ff_pack_bytes(C, hidden)));
                    e->owned.push_back(pk);
                    b.ff_packs = (char*)pk;
                }
                if (b.ff_epoch != e->weights_epoch) {
                    TRY(launch_rowchain_ff_pack(b.ff1.w, b.ff1.b, b.ff2.w, b.ff_packs, C, hidden, true, r.s));
                    b.ff_epoch = e->weights_epoch;
                }
                TRY(launch_rowchain_ff(x2, x3, b.ln3.g, b.ln3.b, b.ff_packs, b.ff2.b, (long)M, C, hidden, 1e-5f, r.s));
            }
        } else {
        half_t* g = nullptr;
        if (fold) {
            LnStats st3;
            TRY(run_ln_stats(r, b.ln3, x2, M, &st3));
            g = r.H(M * 4 * C);
            TRY(run_linear_ln(r, b.ff1, b.ln3, x2, st3, (int)M, g, 4 * C));
        } else {
            half_t* n3 = r.H(M * C);
            TRY(run_ln(r, b.ln3, x2, M, n3, r.lo(x2, MC)));
            g = r.H(M * 4 * C);
            TRY(run_linear(r, b.ff1, n3, (int)M, nullptr, g, 4 * C));
        }
        r.lnp_want = fold && e->ln_fold >= 2;                                   // read by the next block's norm1 (SDXL: depth > 1); unused after the last
        TRY(run_linear(r, b.ff2, g, (int)M, x2, x3, C, 1.f, false, r.lo(x2, MC), r.lo(x3, MC)));
        }
        r.tap(bname, x3, B, H, Wd, C);
        cur = x3;
        ++blk;
        if (reuse) r.ar->rewind(mk);
    }
    half_t* o = reuse ? o_pre : r.S(M * C);
    TRY(run_linear(r, st.proj_out, cur, (int)M, x, o, C, 1.f, false, r.lo(x, MC), r.lo(o, MC)));
    *out = o;
    (void)L;
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// context cache
// ------------------------------------------------------------------------------------------------------------
static void ctx_free(sdmi_engine* e) {
    for (void* p : e->ctx_owned) (void)hipFree(p);
    e->ctx_owned.clear(); e->ctx_k.clear(); e->ctx_vt.clear();
    e->ctx_f16 = nullptr; e->ctx_valid = false;
}

static void collect_st(const UNetW& u, std::vector<const STW*>* out) {
    for (auto& blk : u.input) for (auto& L : blk) if (L.kind == UNetLayer::ST) out->push_back(&L.st);
    for (auto& L : u.middle) if (L.kind == UNetLayer::ST) out->push_back(&L.st);
    for (auto& blk : u.output) for (auto& L : blk) if (L.kind == UNetLayer::ST) out->push_back(&L.st);
}

// `conditional`: the caller does not know whether `ctx` differs from the context of the previous call (the webui re-catenates
// cond | uncond on every step, modules/sd_samplers_cfg_denoiser.py:246).  The decision is taken ON THE DEVICE: a compare kernel
// raises a flag when any fp16-converted element differs from the cached copy, and the copy + every K / V^T projection launch
// below is gated on that flag (GemmP::gate) — an unchanged context costs ~35 empty launches and no host synchronisation.
static int unet_set_context(sdmi_engine* e, const void* ctx, int dtype, int Bn, int L, hipStream_t s, bool conditional = false) {
    UNetW& u = e->unet;
    SDMI_REQUIRE(u.ready, "unet not finalized");
    const int cd = u.cfg.context_dim;
    const int Lpad = rup(L, 64);
    const int* gate = nullptr;
    if (conditional && e->ctx_valid && e->ctx_B == Bn && e->ctx_L == L && !e->ctx_k.empty() && !hn_has_dim(e, cd)) {
        if (!e->ctx_gate) {
            SDMI_CHECK_HIP(hipMalloc((void**)&e->ctx_gate, 256));
            e->owned.push_back(e->ctx_gate);
        }
        SDMI_CHECK_HIP(hipMemsetAsync(e->ctx_gate, 0, sizeof(int), s));
        TRY(launch_ctx_compare(ctx, dtype, e->ctx_f16, Bn, L, Lpad, cd, e->ctx_gate, s));
        gate = e->ctx_gate;
    }
    if (e->ctx_B != Bn || e->ctx_L != L || e->ctx_k.empty()) {
        SDMI_CHECK_HIP(hipStreamSynchronize(s));
        ctx_free(e);
        void* p = nullptr;
        SDMI_CHECK_HIP(hipMalloc(&p, (size_t)Bn * Lpad * cd * sizeof(half_t)));
        e->ctx_owned.push_back(p);
        e->ctx_f16 = (half_t*)p;
        SDMI_CHECK_HIP(hipMemsetAsync(p, 0, (size_t)Bn * Lpad * cd * sizeof(half_t), s));
        std::vector<const STW*> sts;
        collect_st(u, &sts);
        e->ctx_k.assign(u.n_ctx_slots, nullptr);
        e->ctx_vt.assign(u.n_ctx_slots, nullptr);
        for (const STW* st : sts)
            for (const TBlockW& b : st->blocks) {
                SDMI_CHECK_HIP(hipMalloc(&p, (size_t)Bn * L * st->ch * sizeof(half_t)));
                e->ctx_owned.push_back(p);
                e->ctx_k[b.ctx_slot] = (half_t*)p;
                SDMI_CHECK_HIP(hipMalloc(&p, (size_t)Bn * st->ch * Lpad * sizeof(half_t)));
                e->ctx_owned.push_back(p);
                e->ctx_vt[b.ctx_slot] = (half_t*)p;
            }
        e->ctx_B = Bn; e->ctx_L = L; e->ctx_Lpad = Lpad;
    }
    // context -> fp16 rows [Bn][Lpad][cd] (padding rows stay zero)
    if (gate) {
        TRY(launch_ctx_update_gated(ctx, dtype, e->ctx_f16, Bn, L, Lpad, cd, gate, s));
    } else {
        for (int b = 0; b < Bn; ++b) {
            const char* src = (const char*)ctx + (size_t)b * L * cd * (dtype == SDMI_F16 ? 2 : 4);
            TRY(launch_convert_to_f16(src, dtype, e->ctx_f16 + (size_t)b * Lpad * cd, (int64_t)L * cd, s));
        }
    }
    Run r{e, s, false};
    // hypernetworks of the context width: context_k / context_v replace the context in the K / V projections of every layer
    const half_t *ctx_k_src = e->ctx_f16, *ctx_v_src = e->ctx_f16;
    if (hn_has_dim(e, cd)) {
        SDMI_REQUIRE(gate == nullptr, "internal: gated context update with hypernetworks");
        const size_t rows = (size_t)Bn * Lpad;
        for (int pass = 0; pass < 2; ++pass) {               // pass 0 sizes the persistent scratch, pass 1 runs
            size_t off = 0;
            const bool dry = pass == 0;
            auto alloc = [&](size_t n) {
                const size_t a = (off + 255) & ~size_t(255);
                off = a + n * sizeof(half_t);
                return (half_t*)(dry ? (char*)0x1000 + a : e->hn_ctx_scratch + a);
            };
            Run rr{e, s, dry};
            TRY(run_hn(rr, cd, 0, e->ctx_f16, rows, alloc, true, &ctx_k_src));
            TRY(run_hn(rr, cd, 1, e->ctx_f16, rows, alloc, true, &ctx_v_src));
            if (dry && off > e->hn_ctx_scratch_bytes) {
                SDMI_CHECK_HIP(hipStreamSynchronize(s));
                if (e->hn_ctx_scratch) (void)hipFree(e->hn_ctx_scratch);
                SDMI_CHECK_HIP(hipMalloc((void**)&e->hn_ctx_scratch, off + 256));
                e->hn_ctx_scratch_bytes = off;
            }
        }
    }
    std::vector<const STW*> sts;
    collect_st(u, &sts);
    for (const STW* st : sts)
        for (const TBlockW& b : st->blocks) {
            // K[b] = ctx[b] Wk^T : rows L per image, batched over images (compact [Bn*L][C] output)
            ConvArgs c;
            c.a0 = ctx_k_src; c.c0 = cd; c.B = 1; c.Hi = L; c.Wi = 1; c.Ho = L; c.Wo = 1;
            c.out = e->ctx_k[b.ctx_slot]; c.ldo = st->ch;
            c.batch = Bn; c.a_bs = (long)Lpad * cd; c.o_bs = (long)L * st->ch;
            c.gate = gate;
            // the context projections run OUTSIDE a forward's dry-sized arena pass: with a wide context (SDXL: K = 2048) run_conv
            // would hand the GEMM a split-K workspace carved from an arena that may not exist yet (round-2 SDXL test: page fault)
            c.no_split = true;
            TRY(run_conv(r, b.k2, c));
            // V^T[b] = Wv ctx[b]^T : [C][Lpad]
            TRY(run_vt(r, b.v2, ctx_v_src, cd, Bn, Lpad, Lpad, e->ctx_vt[b.ctx_slot], false, gate));
        }
    e->ctx_valid = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// UNet forward
// ------------------------------------------------------------------------------------------------------------
struct Act {
    half_t* p;
    int C, H, W;
};

static int unet_run(Run& r, const void* x, const void* t, const void* y, void* out, int io_dtype, int Bn, int h, int w,
                    int L) {
    sdmi_engine* e = r.e;
    UNetW& u = e->unet;
    const sdmi_unet_config& c = u.cfg;
    const int mc = c.model_channels, ted = mc * 4;
    r.ar->reset();
    // ---- embeddings (fp32 activations, fp16 weights): time_embed(sinusoid(t)) [+ label_emb(y)] --------------
    float* sinus = r.F((size_t)Bn * mc);
    float* e1 = r.F((size_t)Bn * ted);
    float* emb = r.F((size_t)Bn * ted);
    float* embs = r.F((size_t)Bn * u.emb_cols);
    float* emb_act = r.F((size_t)Bn * ted);              // silu(emb), computed once instead of once per output column
    float* yf = c.adm_in_channels > 0 ? r.F((size_t)Bn * c.adm_in_channels) : nullptr;
    float* l1 = c.adm_in_channels > 0 ? r.F((size_t)Bn * ted) : nullptr;
    // Option "uniform_t" (the samplers set it: every row of a CFG batch sits at the same timestep, sd_samplers.py builds the vector with
    // torch.full): the embedding MLP and the 20160-wide ResBlock projection run for ONE row and every image reads that row (row stride 0 of
    // the GEMM epilogues' per-image bias).  Same arithmetic per row: identical bits.  Not with a vector conditioning (label_emb(y) differs
    // per row).  The arena slots keep their Bn-row sizes, so the layout does not depend on the option.
    const int Be = (r.e->uniform_t && c.adm_in_channels == 0) ? 1 : Bn;
    const int emb_ld = Be == 1 ? 0 : u.emb_cols;
    if (!r.dry) {
        TRY(launch_timestep_embedding(t, io_dtype, sinus, Be, mc, r.s));
        TRY(launch_small_linear(sinus, u.te0.w, u.te0.b, nullptr, e1, Be, ted, mc, mc, ted, false, true, r.s));
        TRY(launch_small_linear(e1, u.te2.w, u.te2.b, nullptr, emb, Be, ted, ted, ted, ted, false, false, r.s));
        if (c.adm_in_channels > 0) {
            SDMI_REQUIRE(y != nullptr, "this UNet needs the vector conditioning y");
            TRY(launch_convert_to_f32(y, io_dtype, yf, (int64_t)Bn * c.adm_in_channels, r.s));
            TRY(launch_small_linear(yf, u.le0.w, u.le0.b, nullptr, l1, Bn, ted, c.adm_in_channels, c.adm_in_channels,
                                    ted, false, true, r.s));
            TRY(launch_small_linear(l1, u.le2.w, u.le2.b, emb, emb, Bn, ted, ted, ted, ted, false, false, r.s));
        }
        // every ResBlock's emb_layers (SiLU -> Linear) in one launch
        TRY(launch_silu_f32(emb, emb_act, (int64_t)Be * ted, r.s));
        TRY(launch_small_linear(emb_act, u.emb_all.w, u.emb_all.b, nullptr, embs, Be, u.emb_cols, ted, ted, u.emb_cols,
                                false, false, r.s));
    }
    // ---- input: NCHW -> NHWC fp16, channels zero-padded to the packed conv_in width ----------------------------
    const ConvW& cin_w = u.input[0][0].conv;
    half_t* xin = r.H((size_t)Bn * h * w * cin_w.cin_pad);
    // (accuracy mode: an fp32 latent enters conv_in as a (hi, lo) pair in the zero-padded input channels — its fp16 rounding alone is
    // 2.9e-4 of the first activation)
    const bool xin_lo = r.acc() && io_dtype != SDMI_F16 && 2 * c.in_channels <= cin_w.cin_pad;
    if (!r.dry) TRY(launch_nchw_to_nhwc(x, io_dtype, xin, Bn, c.in_channels, h * w, cin_w.cin_pad, 1.0f, nullptr, nullptr, r.s, xin_lo));

    std::vector<Act> hs;
    Act cur{nullptr, 0, h, w};
    // Option "cfg_pairs" (set per call by the CFG denoiser for its plain [cond | uncond] batch): rows [Bn / 2, Bn) carry the same latent
    // and timestep as rows [0, Bn / 2) — only the text context differs — so every layer in front of the first cross-attention gives the
    // same result for both halves: conv_in, the first ResBlock, and GroupNorm / proj_in / norm1 / self-attention of the first transformer
    // block (SD1.x / 2.x: the largest self-attention launch of the forward) run for Bn / 2 rows and are copied (three 21 MB device copies
    // at the C1 batch against ~0.48 ms of kernels).  Same function; the fp32 summation order of those layers follows the halved M.
    // Not with a vector conditioning (label_emb(y) differs per row), taps, LayerNorm fold, hypernetworks, arena reuse or batch slices.
    const bool pairs = e->cfg_pairs && c.adm_in_channels == 0 && Bn % 2 == 0 && Bn >= 2 && !e->trace && !e->ln_fold && !r.reuse() &&
                       e->hypernets.empty() && (r.Btot == 0 || r.Btot == Bn);
    const int Bh = Bn / 2;
    bool shared = pairs;                                     // cur.p: rows [0, Bh) computed, buffer sized for Bn rows
    auto dup = [&](const Act& a) -> int {                    // copy the computed half of a full-size activation to the other half
        const size_t n = (size_t)Bh * a.H * a.W * a.C;
        if (!r.dry) SDMI_CHECK_HIP(hipMemcpyAsync((half_t*)a.p + n, a.p, n * sizeof(half_t), hipMemcpyDeviceToDevice, r.s));
        if (!r.dry && r.acc())                               // (hi, lo) pair: the lo half sits 2 n elements behind the hi half
            SDMI_CHECK_HIP(hipMemcpyAsync((half_t*)a.p + 3 * n, a.p + 2 * n, n * sizeof(half_t), hipMemcpyDeviceToDevice, r.s));
        return 0;
    };
    auto run_block = [&](const std::vector<UNetLayer>& blk, const Act* skip, const std::string& bname) -> int {
        bool first = true;
        int li = 0;
        for (const UNetLayer& Lr : blk) {
            const std::string lname = bname + "." + std::to_string(li++);
            if (shared && (Lr.kind == UNetLayer::DOWN || Lr.kind == UNetLayer::UP || (Lr.kind == UNetLayer::RES && skip && first))) {
                TRY(dup(cur));                               // (no transformer before the first resolution change: give up the sharing here)
                shared = false;
            }
            switch (Lr.kind) {
                case UNetLayer::CONV_IN: {
                    half_t* o = r.S((size_t)Bn * cur.H * cur.W * Lr.conv.n_pad);
                    if (shared) {
                        ConvArgs a;
                        a.a0 = xin; a.c0 = Lr.conv.cin_pad; a.B = Bh; a.Hi = cur.H; a.Wi = cur.W; a.Ho = cur.H; a.Wo = cur.W;
                        a.pad = 1; a.out = o; a.ldo = Lr.conv.n_pad;
                        a.out_lo = r.lo(o, (size_t)Bn * cur.H * cur.W * Lr.conv.n_pad);
                        TRY(run_conv(r, Lr.conv, a));
                        cur = Act{o, Lr.conv.cout, cur.H, cur.W};
                        break;
                    }
                    ConvArgs a;
                    a.a0 = xin; a.c0 = Lr.conv.cin_pad; a.B = Bn; a.Hi = cur.H; a.Wi = cur.W; a.Ho = cur.H; a.Wo = cur.W;
                    a.pad = 1; a.out = o; a.ldo = Lr.conv.n_pad;
                    a.out_lo = r.lo(o, (size_t)Bn * cur.H * cur.W * Lr.conv.n_pad);
                    TRY(run_conv(r, Lr.conv, a));
                    cur = Act{o, Lr.conv.cout, cur.H, cur.W};
                    break;
                }
                case UNetLayer::RES: {
                    half_t* o = nullptr;
                    if (skip && first) {
                        SDMI_REQUIRE(Lr.c0 == cur.C && Lr.c1 == skip->C, "skip concat channel mismatch");
                        SDMI_REQUIRE(cur.H == skip->H && cur.W == skip->W, "skip connection spatial mismatch");
                        TRY(run_res(r, Lr.res, cur.p, skip->p, cur.C, skip->C, Bn, cur.H, cur.W, 1e-5f, embs, emb_ld, &o, 1.f, nullptr, true));
                    } else if (shared) {
                        half_t* full = r.S((size_t)Bn * cur.H * cur.W * Lr.res.cout);
                        TRY(run_res(r, Lr.res, cur.p, nullptr, cur.C, 0, Bh, cur.H, cur.W, 1e-5f, embs, emb_ld, &o, 1.f, full, true,
                                    (size_t)Bn * cur.H * cur.W));
                    } else {
                        TRY(run_res(r, Lr.res, cur.p, nullptr, cur.C, 0, Bn, cur.H, cur.W, 1e-5f, embs, emb_ld, &o, 1.f, nullptr, true));
                    }
                    cur = Act{o, Lr.res.cout, cur.H, cur.W};
                    break;
                }
                case UNetLayer::ST: {
                    half_t* o = nullptr;
                    if (shared) TRY(dup(cur));               // proj_out adds the block's input to all Bn rows
                    TRY(run_st(r, Lr.st, cur.p, Bn, cur.H, cur.W, L, &o, lname, shared ? Bh : 0));
                    shared = false;
                    cur.p = o;
                    break;
                }
                case UNetLayer::DOWN: {
                    const int Ho = (cur.H + 2 - 3) / 2 + 1, Wo = (cur.W + 2 - 3) / 2 + 1;
                    half_t* o = r.S((size_t)Bn * Ho * Wo * cur.C);
                    ConvArgs a;
                    a.a0 = cur.p; a.c0 = cur.C; a.B = Bn; a.Hi = cur.H; a.Wi = cur.W; a.Ho = Ho; a.Wo = Wo;
                    a.stride = 2; a.pad = 1; a.out = o; a.ldo = cur.C;
                    a.out_lo = r.lo(o, (size_t)Bn * Ho * Wo * cur.C);
                    TRY(run_conv(r, Lr.conv, a));
                    cur = Act{o, cur.C, Ho, Wo};
                    break;
                }
                case UNetLayer::UP: {
                    const int Ho = cur.H * 2, Wo = cur.W * 2;
                    half_t* o = r.S((size_t)Bn * Ho * Wo * cur.C);
                    ConvArgs a;
                    a.a0 = cur.p; a.c0 = cur.C; a.B = Bn; a.Hi = cur.H; a.Wi = cur.W; a.Ho = Ho; a.Wo = Wo;
                    a.up = 1; a.pad = 1; a.out = o; a.ldo = cur.C;
                    a.out_lo = r.lo(o, (size_t)Bn * Ho * Wo * cur.C);
                    TRY(run_conv(r, Lr.conv, a));
                    cur = Act{o, cur.C, Ho, Wo};
                    break;
                }
            }
            first = false;
            r.tap(lname, cur.p, Bn, cur.H, cur.W, cur.C);
        }
        return 0;
    };
    if (!r.dry) e->taps.clear();
    // ControlNet residuals (sdmi_unet_set_control; ldm cldm.py ControlledUnetModel.forward): control[i] is added to input block i's
    // output where the output blocks read it as their skip connection — the stream itself goes on without it — and the last entry to the
    // middle block's output.  One transposing add per tensor (they arrive NCHW) into a buffer of its own: the arena layout of a forward
    // with residuals differs from one without, so the dry pass sees them too.  Rows of a batch slice (option "streams") start at b0.
    const bool have_ctl = !e->control.empty();
    const size_t n_ctl = e->control.size();
    if (have_ctl) SDMI_REQUIRE(n_ctl == u.input.size() + 1, "control: one tensor per input block output and one for the middle block");
    auto add_control = [&](Act a, size_t idx, Act* out) -> int {
        const size_t n = (size_t)Bn * a.H * a.W * a.C;
        half_t* o = r.S(n);
        if (!r.dry) {
            const size_t full = (size_t)(r.Btot ? r.Btot : Bn) * a.C * a.H * a.W;
            SDMI_REQUIRE((size_t)e->control_numel[idx] == full, "control tensor " + std::to_string(idx) + " does not have the shape of the activation it is added to");
            const size_t elt = io_dtype == SDMI_F16 ? 2 : 4;
            const char* src = (const char*)e->control[idx] + (size_t)r.b0 * a.C * a.H * a.W * elt;
            TRY(launch_add_nchw_residual(a.p, r.lo(a.p, n), src, io_dtype, o, r.lo(o, n), Bn, a.C, a.H * a.W, r.s));
        }
        *out = Act{o, a.C, a.H, a.W};
        return 0;
    };
    int bidx = 0;
    for (auto& blk : u.input) {
        TRY(run_block(blk, nullptr, "input_blocks." + std::to_string(bidx++)));
        if (shared) TRY(dup(cur));                           // the skip connection is read for all Bn rows (the next block goes on with Bh)
        Act sk = cur;
        if (have_ctl && !e->only_mid_control) TRY(add_control(cur, hs.size(), &sk));
        hs.push_back(sk);
    }
    shared = false;                                          // (a UNet without a transformer on the way down: every hs entry is complete)
    TRY(run_block(u.middle, nullptr, "middle_block"));
    if (have_ctl) TRY(add_control(cur, n_ctl - 1, &cur));
    bidx = 0;
    for (auto& blk : u.output) {
        Act sk = hs.back();
        hs.pop_back();
        TRY(run_block(blk, &sk, "output_blocks." + std::to_string(bidx++)));
    }
    // ---- out: GroupNorm32 + SiLU + conv 3x3 -> fp32 NCHW ---------------------------------------------------
    const size_t M = (size_t)Bn * cur.H * cur.W;
    half_t* tn = r.H(M * cur.C);
    TRY(run_gn(r, u.out_norm, cur.p, nullptr, cur.C, 0, Bn, cur.H * cur.W, 1e-5f, true, tn, r.lo(cur.p, M * cur.C)));
    float* eps = r.F((size_t)Bn * c.out_channels * cur.H * cur.W);
    {
        ConvArgs a;
        a.a0 = tn; a.c0 = cur.C; a.B = Bn; a.Hi = cur.H; a.Wi = cur.W; a.Ho = cur.H; a.Wo = cur.W; a.pad = 1;
        a.out = eps; a.flags = EP_NCHW; a.n_real = c.out_channels;
        TRY(run_conv(r, u.out_conv, a));
    }
    if (!r.dry) TRY(launch_copy_out(eps, out, io_dtype, (int64_t)Bn * c.out_channels * cur.H * cur.W, r.s));
    return 0;
}

static int ensure_arena(Arena& ar, size_t need, hipStream_t s) {
    if (need <= ar.cap) return 0;
    SDMI_CHECK_HIP(hipStreamSynchronize(s));
    if (ar.base) SDMI_CHECK_HIP(hipFree(ar.base));
    ar.base = nullptr; ar.cap = 0;
    const size_t cap = need + (need >> 4) + (1 << 20);
    SDMI_CHECK_HIP(hipMalloc((void**)&ar.base, cap));
    ar.cap = cap;
    return 0;
}
static int ensure_arena(sdmi_engine* e, size_t need, hipStream_t s) { return ensure_arena(e->arena, need, s); }

// one UNet pass over rows [b0, b0 + Bn) of a call of Btot rows, on stream s out of arena ar: dry pass to size the arena (pure host
// arithmetic), then the launches
static int unet_forward_slice(sdmi_engine* e, Arena& ar, const void* x, const void* t, const void* y, void* out, int io_dtype,
                              int Bn, int b0, int Btot, int h, int w, int L, hipStream_t s) {
    Run dry(e, s, true, &ar, b0, Btot);
    ar.dry = true; ar.high = 0;
    TRY(unet_run(dry, x, t, y, out, io_dtype, Bn, h, w, L));
    ar.dry = false;
    TRY(ensure_arena(ar, ar.high, s));
    Run run(e, s, false, &ar, b0, Btot);
    return unet_run(run, x, t, y, out, io_dtype, Bn, h, w, L);
}

int unet_forward(sdmi_engine* e, const void* x, const void* t, const void* ctx, const void* y, void* out, int io_dtype,
                 int Bn, int h, int w, int L, hipStream_t s) {
    SDMI_REQUIRE(e->unet.ready, "unet not finalized");
    SDMI_CHECK_HIP(hipSetDevice(e->device));
    // Options "cfg_pairs" / "uniform_t" are promises of the CALLER about x and t (rows [Bn/2, Bn) repeat rows [0, Bn/2); one timestep for
    // all rows) that the forward does not re-derive: a caller that sets them through sdmi_engine_set_option and then passes other data
    // gets the second half silently overwritten.  SDMI_CHECK_PROMISES=1 (tests, tools/gpu A/B scripts) verifies them on the host — a
    // synchronising copy, so it is a debugging switch, not a default.
    const char* cpv = getenv("SDMI_CHECK_PROMISES");
    const bool check_promises = cpv && atoi(cpv) != 0;
    if (check_promises && (e->cfg_pairs || e->uniform_t)) {
        const size_t elt = io_dtype == SDMI_F16 ? 2 : 4;
        SDMI_CHECK_HIP(hipStreamSynchronize(s));
        if (e->uniform_t && Bn > 1) {
            std::vector<char> th((size_t)Bn * elt);
            SDMI_CHECK_HIP(hipMemcpy(th.data(), t, th.size(), hipMemcpyDeviceToHost));
            for (int b = 1; b < Bn; ++b)
                SDMI_REQUIRE(memcmp(th.data(), th.data() + (size_t)b * elt, elt) == 0, "option uniform_t is set but the timesteps of the call differ");
        }
        if (e->cfg_pairs && Bn % 2 == 0 && Bn >= 2) {
            const size_t half_bytes = (size_t)(Bn / 2) * e->unet.cfg.in_channels * h * w * elt;
            std::vector<char> xh(2 * half_bytes), th((size_t)Bn * elt);
            SDMI_CHECK_HIP(hipMemcpy(xh.data(), x, xh.size(), hipMemcpyDeviceToHost));
            SDMI_CHECK_HIP(hipMemcpy(th.data(), t, th.size(), hipMemcpyDeviceToHost));
            SDMI_REQUIRE(memcmp(xh.data(), xh.data() + half_bytes, half_bytes) == 0,
                         "option cfg_pairs is set but rows [Bn/2, Bn) of x do not repeat rows [0, Bn/2)");
            SDMI_REQUIRE(memcmp(th.data(), th.data() + th.size() / 2, th.size() / 2) == 0,
                         "option cfg_pairs is set but the timesteps of the two halves differ");
        }
    }
    // Option "auto_promises": the same two facts DERIVED from the call's data, for a caller that cannot know them (the webui's stock CFG
    // denoiser behind Mi355xUnet.forward hands over an anonymous batch): cfg_pairs / uniform_t hold for this call exactly if the compare
    // says so, and are restored afterwards.  One synchronising copy of x and t per forward (1 MB at the C1 batch).
    struct Restore {
        sdmi_engine* e; bool pairs, uni, on;
        ~Restore() { if (on) { e->cfg_pairs = pairs; e->uniform_t = uni; } }
    } restore{e, e->cfg_pairs, e->uniform_t, e->auto_promises};
    if (e->auto_promises) {
        const size_t elt = io_dtype == SDMI_F16 ? 2 : 4;
        const size_t half_bytes = (size_t)(Bn / 2) * e->unet.cfg.in_channels * h * w * elt;
        std::vector<char> th((size_t)Bn * elt), xh(Bn % 2 == 0 && Bn >= 2 ? 2 * half_bytes : 0);
        SDMI_CHECK_HIP(hipStreamSynchronize(s));
        SDMI_CHECK_HIP(hipMemcpy(th.data(), t, th.size(), hipMemcpyDeviceToHost));
        bool uni = true;
        for (int b = 1; b < Bn && uni; ++b) uni = memcmp(th.data(), th.data() + (size_t)b * elt, elt) == 0;
        bool pairs = uni && !xh.empty();
        if (pairs) {
            SDMI_CHECK_HIP(hipMemcpy(xh.data(), x, xh.size(), hipMemcpyDeviceToHost));
            pairs = memcmp(xh.data(), xh.data() + half_bytes, half_bytes) == 0;
        }
        e->uniform_t = uni;
        e->cfg_pairs = pairs;
    }
    struct ClearControl {                                    // the residuals belong to this call alone
        sdmi_engine* e;
        ~ClearControl() { e->control.clear(); e->control_numel.clear(); e->only_mid_control = false; }
    } clear_control{e};
    if (ctx) TRY(unet_set_context(e, ctx, io_dtype, Bn, L, s));
    // Option "streams" = n > 1: the rows of a call are independent (own timestep, own context rows), so the batch is cut into n
    // equal slices that run the same launch sequence on n HIP streams out of n arenas.  The GPU then always has a second, independent
    // kernel queue to draw workgroups from: the HBM-bound norm / elementwise launches of one slice run under the MFMA-bound GEMMs of
    // the other, and a launch that leaves CUs idle (tail wave, small-M levels) no longer idles them.  Same arithmetic per row; the tile
    // configuration follows the slice's M, so the bits are those of a call with Bn / n rows.  Not while block outputs are tapped.
    const int ns = (e->n_streams > 1 && !e->trace && Bn % e->n_streams == 0) ? e->n_streams : 1;
    // (ln_fold rebuilds its folded weight copies lazily on the stream of the slice that meets them first; the other slices would read
    // them with no event dependency on that stream — the two experimental options are mutually exclusive)
    SDMI_REQUIRE(!(ns > 1 && e->ln_fold), "engine options ln_fold and streams > 1 cannot be combined");
    if (ns == 1) return unet_forward_slice(e, e->arena, x, t, y, out, io_dtype, Bn, 0, Bn, h, w, L, s);
    const sdmi_unet_config& c = e->unet.cfg;
    while ((int)e->aux_streams.size() < ns - 1) {
        hipStream_t st;
        SDMI_CHECK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        e->aux_streams.push_back(st);
        hipEvent_t ev;
        SDMI_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        e->ev_join.push_back(ev);
        e->aux_arenas.emplace_back();
    }
    if (!e->ev_fork) SDMI_CHECK_HIP(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    const int Bc = Bn / ns;
    const size_t elt = io_dtype == SDMI_F16 ? 2 : 4;
    const size_t xs = (size_t)Bc * c.in_channels * h * w * elt, os = (size_t)Bc * c.out_channels * h * w * elt;
    const size_t ts = (size_t)Bc * elt, ys = (size_t)Bc * (c.adm_in_channels > 0 ? c.adm_in_channels : 0) * elt;
    SDMI_CHECK_HIP(hipEventRecord(e->ev_fork, s));           // the inputs (and the context cache) are ready on the caller's stream
    for (int i = 1; i < ns; ++i) SDMI_CHECK_HIP(hipStreamWaitEvent(e->aux_streams[i - 1], e->ev_fork, 0));
    for (int i = 0; i < ns; ++i) {
        hipStream_t st = i == 0 ? s : e->aux_streams[i - 1];
        Arena& ar = i == 0 ? e->arena : e->aux_arenas[i - 1];
        TRY(unet_forward_slice(e, ar, (const char*)x + i * xs, (const char*)t + i * ts, y ? (const char*)y + i * ys : nullptr,
                               (char*)out + i * os, io_dtype, Bc, i * Bc, Bn, h, w, L, st));
    }
    for (int i = 1; i < ns; ++i) {
        SDMI_CHECK_HIP(hipEventRecord(e->ev_join[i - 1], e->aux_streams[i - 1]));
        SDMI_CHECK_HIP(hipStreamWaitEvent(s, e->ev_join[i - 1], 0));
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// VAE
// ------------------------------------------------------------------------------------------------------------
static int build_vae_attn(sdmi_engine* e, const std::string& name, int c, VAEAttnW* a) {
    auto& m = e->raw_vae;
    a->c = c;
    TRY(pack_norm(e, m, name + ".norm", &a->norm));
    TRY(pack_stack(e, m, {name + ".q", name + ".k"}, true, false, &a->qk));
    TRY(pack_one(e, m, name + ".v", true, false, &a->v));
    TRY(pack_one(e, m, name + ".proj_out", true, false, &a->proj));
    return 0;
}

static int vae_build(sdmi_engine* e) {
    VAEW& v = e->vae;
    const sdmi_vae_config& c = v.cfg;
    auto& m = e->raw_vae;
    const int nres = c.num_levels, ch = c.ch, z = c.z_channels;
    // post_quant_conv as fp32 matrices
    {
        const RawTensor* w = find_raw(m, "post_quant_conv.weight");
        const RawTensor* b = find_raw(m, "post_quant_conv.bias");
        SDMI_REQUIRE(w && b, "missing post_quant_conv");
        TRY(dev_alloc(e, (void**)&v.pqc_w, (size_t)z * z * sizeof(float)));
        TRY(dev_alloc(e, (void**)&v.pqc_b, (size_t)z * sizeof(float)));
        TRY(launch_convert_to_f32(w->ptr, w->dtype, v.pqc_w, (int64_t)z * z, 0));
        TRY(launch_convert_to_f32(b->ptr, b->dtype, v.pqc_b, z, 0));
    }
    int bi = ch * c.ch_mult[nres - 1];
    TRY(pack_one(e, m, "decoder.conv_in", true, false, &v.d_conv_in));
    TRY(build_res(e, "decoder.mid.block_1", bi, bi, true, &v.d_mid1));
    TRY(build_vae_attn(e, "decoder.mid.attn_1", bi, &v.d_attn));
    TRY(build_res(e, "decoder.mid.block_2", bi, bi, true, &v.d_mid2));
    v.d_up.assign(nres, VAELevel{});
    for (int i = nres - 1; i >= 0; --i) {
        const int bo = ch * c.ch_mult[i];
        for (int j = 0; j <= c.num_res_blocks; ++j) {
            ResW r;
            TRY(build_res(e, "decoder.up." + std::to_string(i) + ".block." + std::to_string(j), bi, bo, true, &r));
            v.d_up[i].blocks.push_back(r);
            bi = bo;
        }
        if (i != 0) {
            v.d_up[i].has_resample = true;
            TRY(pack_one(e, m, "decoder.up." + std::to_string(i) + ".upsample.conv", true, false, &v.d_up[i].resample));
        }
    }
    TRY(pack_norm(e, m, "decoder.norm_out", &v.d_norm_out));
    TRY(pack_one(e, m, "decoder.conv_out", true, false, &v.d_conv_out));

    v.has_encoder = find_raw(m, "encoder.conv_in.weight") != nullptr && find_raw(m, "quant_conv.weight") != nullptr;
    if (v.has_encoder) {
        TRY(pack_one(e, m, "encoder.conv_in", true, false, &v.e_conv_in));
        v.e_down.assign(nres, VAELevel{});
        int bin = ch;
        for (int i = 0; i < nres; ++i) {
            const int bo = ch * c.ch_mult[i];
            for (int j = 0; j < c.num_res_blocks; ++j) {
                ResW r;
                TRY(build_res(e, "encoder.down." + std::to_string(i) + ".block." + std::to_string(j), bin, bo, true, &r));
                v.e_down[i].blocks.push_back(r);
                bin = bo;
            }
            if (i != nres - 1) {
                v.e_down[i].has_resample = true;
                TRY(pack_one(e, m, "encoder.down." + std::to_string(i) + ".downsample.conv", true, false, &v.e_down[i].resample));
            }
        }
        TRY(build_res(e, "encoder.mid.block_1", bin, bin, true, &v.e_mid1));
        TRY(build_vae_attn(e, "encoder.mid.attn_1", bin, &v.e_attn));
        TRY(build_res(e, "encoder.mid.block_2", bin, bin, true, &v.e_mid2));
        TRY(pack_norm(e, m, "encoder.norm_out", &v.e_norm_out));
        // fold quant_conv (1x1, 2z -> 2z) into encoder.conv_out on the host:  W' = Wq Wc,  b' = Wq bc + bq
        const RawTensor* wc = find_raw(m, "encoder.conv_out.weight");
        const RawTensor* bc = find_raw(m, "encoder.conv_out.bias");
        const RawTensor* wq = find_raw(m, "quant_conv.weight");
        const RawTensor* bq = find_raw(m, "quant_conv.bias");
        SDMI_REQUIRE(wc && bc && wq && bq, "missing encoder.conv_out / quant_conv");
        const int O = (int)wc->shape[0], I = (int)wc->shape[1];
        const size_t per = (size_t)I * 9;
        std::vector<float> hwc((size_t)O * per), hbc(O), hwq((size_t)O * O), hbq(O);
        float *dwc, *dbc, *dwq, *dbq;
        SDMI_CHECK_HIP(hipMalloc((void**)&dwc, hwc.size() * 4));
        SDMI_CHECK_HIP(hipMalloc((void**)&dbc, O * 4));
        SDMI_CHECK_HIP(hipMalloc((void**)&dwq, hwq.size() * 4));
        SDMI_CHECK_HIP(hipMalloc((void**)&dbq, O * 4));
        TRY(launch_convert_to_f32(wc->ptr, wc->dtype, dwc, hwc.size(), 0));
        TRY(launch_convert_to_f32(bc->ptr, bc->dtype, dbc, O, 0));
        TRY(launch_convert_to_f32(wq->ptr, wq->dtype, dwq, hwq.size(), 0));
        TRY(launch_convert_to_f32(bq->ptr, bq->dtype, dbq, O, 0));
        SDMI_CHECK_HIP(hipMemcpy(hwc.data(), dwc, hwc.size() * 4, hipMemcpyDeviceToHost));
        SDMI_CHECK_HIP(hipMemcpy(hbc.data(), dbc, O * 4, hipMemcpyDeviceToHost));
        SDMI_CHECK_HIP(hipMemcpy(hwq.data(), dwq, hwq.size() * 4, hipMemcpyDeviceToHost));
        SDMI_CHECK_HIP(hipMemcpy(hbq.data(), dbq, O * 4, hipMemcpyDeviceToHost));
        std::vector<float> fw((size_t)O * per, 0.f), fb(O, 0.f);
        for (int o = 0; o < O; ++o) {
            double bacc = hbq[o];
            for (int k = 0; k < O; ++k) {
                const float q = hwq[(size_t)o * O + k];
                bacc += (double)q * hbc[k];
                for (size_t j = 0; j < per; ++j) fw[(size_t)o * per + j] += q * hwc[(size_t)k * per + j];
            }
            fb[o] = (float)bacc;
        }
        (void)hipFree(dwc); (void)hipFree(dbc); (void)hipFree(dwq); (void)hipFree(dbq);
        std::map<std::string, RawTensor> tmp;
        const int64_t wshape[4] = {O, I, 3, 3}, bshape[1] = {O};
        TRY(load_raw(tmp, "f.weight", fw.data(), SDMI_F32, 4, wshape, 0));
        TRY(load_raw(tmp, "f.bias", fb.data(), SDMI_F32, 1, bshape, 0));
        const int rc = pack_one(e, tmp, "f", true, false, &v.e_conv_out);
        SDMI_CHECK_HIP(hipDeviceSynchronize());
        free_raw(tmp);
        TRY(rc);
    }
    SDMI_CHECK_HIP(hipDeviceSynchronize());
    v.ready = true;
    return 0;
}

// single-head spatial attention of the VAE mid block (N = H*W tokens, d = C = 512): scores materialised through the
// GEMM kernel (fp32), row softmax, then P V — modules/sd_hijack_optimizations.py:554-610 computes the same product chunked.
int g_vae_attn_rows = [] { const char* e = getenv("SDMI_VAE_ATTN_ROWS"); return e ? atoi(e) : 16384; }();   // query rows per block of the VAE's attention (debug knob "vae_attn_rows")
static int run_vae_attn(Run& r, const VAEAttnW& a, const half_t* x, int B, int H, int Wd, half_t** out, float ss = 1.f) {
    const int C = a.c, HW = H * Wd, Npad = rup(HW, 64);
    const size_t M = (size_t)B * HW;
    half_t* n0 = r.H(M * C);
    TRY(run_gn(r, a.norm, x, nullptr, C, 0, B, HW, 1e-6f * ss * ss, false, n0));
    half_t* qk = r.H(M * 2 * C);
    TRY(run_linear(r, a.qk, n0, (int)M, nullptr, qk, 2 * C));
    half_t* vt = r.H((size_t)B * C * Npad);
    TRY(run_vt(r, a.v, n0, C, B, HW, Npad, vt, true));
    // Scores are materialised (one head, d = 512: softmax_rows between two batched GEMMs).  Beyond `g_vae_attn_rows` query rows the
    // product runs in blocks of that many rows over the whole key set — softmax rows are independent, so the result is the unblocked
    // product's (modules/sub_quadratic_attention.py / sd_hijack_optimizations.py:554-610 chunk the reference's product the same way):
    // a 2048x2048 decode (N = 65536) would otherwise need 17 GB of fp32 scores per image, a 4096x4096 one 275 GB (round 6).
    const int RB = (g_vae_attn_rows > 0 && HW > g_vae_attn_rows) ? g_vae_attn_rows : HW;
    const int nblk = cdiv(HW, RB);
    float* S = r.F((size_t)(nblk == 1 ? B : 1) * RB * Npad);
    half_t* P = r.H((size_t)(nblk == 1 ? B : 1) * RB * Npad);
    half_t* o = r.H(M * C);
    if (!r.dry) {
        for (int b = 0; b < (nblk == 1 ? 1 : B); ++b)
            for (int k = 0; k < nblk; ++k) {
                const int row0 = k * RB, rows = std::min(RB, HW - row0), nb = nblk == 1 ? B : 1;
                const size_t img = (size_t)b * HW;
                GemmP p{};
                p.a0 = qk + (img + row0) * 2 * C; p.c0 = C; p.cin = C; p.lda0 = 2 * C;
                p.w = qk + img * 2 * C + C; p.ldw = 2 * C;
                p.out = S;
                p.Hi = rows; p.Wi = 1; p.Ho = rows; p.Wo = 1; p.taps = 1; p.stride = 1;
                p.M = rows; p.N = Npad; p.n_valid = HW; p.K = C; p.ldo = Npad; p.rows_per_batch = rows; p.n_real = Npad;
                p.flags = EP_OUT_F32;
                p.alpha = 1.0f / sqrtf((float)C);
                p.a_bs = (long)HW * 2 * C; p.w_bs = (long)HW * 2 * C; p.o_bs = (long)RB * Npad;
                TRY(launch_gemm(p, nb, r.e->force_generic, r.e->use_glds, r.s));
                TRY(launch_softmax_rows(S, P, (int64_t)nb * rows, HW, Npad, Npad, r.s));
                GemmP g{};
                g.a0 = P; g.c0 = Npad; g.cin = Npad; g.lda0 = Npad;
                g.w = vt + (size_t)b * C * Npad; g.ldw = Npad;
                g.out = o + (img + row0) * C;
                g.Hi = rows; g.Wi = 1; g.Ho = rows; g.Wo = 1; g.taps = 1; g.stride = 1;
                g.M = rows; g.N = C; g.K = Npad; g.ldo = C; g.rows_per_batch = rows; g.n_real = C;
                g.alpha = 1.f;
                g.a_bs = (long)RB * Npad; g.w_bs = (long)C * Npad; g.o_bs = (long)HW * C;
                TRY(launch_gemm(g, nb, r.e->force_generic, r.e->use_glds, r.s));
            }
    }
    half_t* y = r.H(M * C);
    TRY(run_linear(r, a.proj, o, (int)M, x, y, C, ss));
    *out = y;
    return 0;
}

static int vae_decode_run(Run& r, const void* z, int io_dtype, float* out, int B, int h, int w) {
    sdmi_engine* e = r.e;
    VAEW& v = e->vae;
    const sdmi_vae_config& c = v.cfg;
    e->arena.reset();
    const float ss = e->vae_stream_scale;                    // 1, or 1/64 in the range-extended mode (see engine.h)
    half_t* zin = r.H((size_t)B * h * w * v.d_conv_in.cin_pad);
    if (!r.dry)
        TRY(launch_nchw_to_nhwc(z, io_dtype, zin, B, c.z_channels, h * w, v.d_conv_in.cin_pad, 1.0f / c.scale_factor,
                                v.pqc_w, v.pqc_b, r.s));
    int H = h, W = w;
    int C = v.d_conv_in.cout;
    half_t* cur = r.H((size_t)B * H * W * C);
    {
        ConvArgs a;
        a.a0 = zin; a.c0 = v.d_conv_in.cin_pad; a.B = B; a.Hi = H; a.Wi = W; a.Ho = H; a.Wo = W; a.pad = 1;
        a.out = cur; a.ldo = C;
        a.alpha = ss; a.bias_scale = ss;
        TRY(run_conv(r, v.d_conv_in, a));
    }
    if (!r.dry) e->taps.clear();
    r.tap("decoder.conv_in", cur, B, H, W, C);
    half_t* o = nullptr;
    TRY(run_res(r, v.d_mid1, cur, nullptr, C, 0, B, H, W, 1e-6f, nullptr, 0, &o, ss)); cur = o;
    r.tap("decoder.mid.block_1", cur, B, H, W, C);
    TRY(run_vae_attn(r, v.d_attn, cur, B, H, W, &o, ss)); cur = o;
    r.tap("decoder.mid.attn_1", cur, B, H, W, C);
    TRY(run_res(r, v.d_mid2, cur, nullptr, C, 0, B, H, W, 1e-6f, nullptr, 0, &o, ss)); cur = o;
    r.tap("decoder.mid.block_2", cur, B, H, W, C);
    for (int i = c.num_levels - 1; i >= 0; --i) {
        int bj = 0;
        for (const ResW& rb : v.d_up[i].blocks) {
            TRY(run_res(r, rb, cur, nullptr, C, 0, B, H, W, 1e-6f, nullptr, 0, &o, ss));
            cur = o; C = rb.cout;
            r.tap("decoder.up." + std::to_string(i) + ".block." + std::to_string(bj++), cur, B, H, W, C);
        }
        if (v.d_up[i].has_resample) {
            half_t* u = r.H((size_t)B * (2 * H) * (2 * W) * C);
            ConvArgs a;
            a.a0 = cur; a.c0 = C; a.B = B; a.Hi = H; a.Wi = W; a.Ho = 2 * H; a.Wo = 2 * W; a.up = 1; a.pad = 1;
            a.out = u; a.ldo = C;
            a.bias_scale = ss;                                // stream in, stream out
            TRY(run_conv(r, v.d_up[i].resample, a));
            cur = u; H *= 2; W *= 2;
            r.tap("decoder.up." + std::to_string(i) + ".upsample", cur, B, H, W, C);
        }
    }
    half_t* tn = r.H((size_t)B * H * W * C);
    TRY(run_gn(r, v.d_norm_out, cur, nullptr, C, 0, B, H * W, 1e-6f * ss * ss, true, tn));
    {
        ConvArgs a;
        a.a0 = tn; a.c0 = C; a.B = B; a.Hi = H; a.Wi = W; a.Ho = H; a.Wo = W; a.pad = 1;
        a.out = out; a.flags = EP_NCHW; a.n_real = c.out_ch;
        TRY(run_conv(r, v.d_conv_out, a));
    }
    return 0;
}

static int vae_encode_run(Run& r, const void* x, int io_dtype, float* out, int B, int Hin, int Win) {
    sdmi_engine* e = r.e;
    VAEW& v = e->vae;
    const sdmi_vae_config& c = v.cfg;
    e->arena.reset();
    int H = Hin, W = Win;
    half_t* xin = r.H((size_t)B * H * W * v.e_conv_in.cin_pad);
    if (!r.dry) TRY(launch_nchw_to_nhwc(x, io_dtype, xin, B, c.in_channels, H * W, v.e_conv_in.cin_pad, 1.0f, nullptr, nullptr, r.s));
    int C = v.e_conv_in.cout;
    half_t* cur = r.H((size_t)B * H * W * C);
    {
        ConvArgs a;
        a.a0 = xin; a.c0 = v.e_conv_in.cin_pad; a.B = B; a.Hi = H; a.Wi = W; a.Ho = H; a.Wo = W; a.pad = 1;
        a.out = cur; a.ldo = C;
        TRY(run_conv(r, v.e_conv_in, a));
    }
    half_t* o = nullptr;
    for (int i = 0; i < c.num_levels; ++i) {
        for (const ResW& rb : v.e_down[i].blocks) {
            TRY(run_res(r, rb, cur, nullptr, C, 0, B, H, W, 1e-6f, nullptr, 0, &o));
            cur = o; C = rb.cout;
        }
        if (v.e_down[i].has_resample) {
            // pad (0,1,0,1) then 3x3 stride 2 pad 0  (sd3_impls.py:227-236): out = floor((H + 1 - 3)/2) + 1
            const int Ho = (H + 1 - 3) / 2 + 1, Wo = (W + 1 - 3) / 2 + 1;
            half_t* d = r.H((size_t)B * Ho * Wo * C);
            ConvArgs a;
            a.a0 = cur; a.c0 = C; a.B = B; a.Hi = H; a.Wi = W; a.Ho = Ho; a.Wo = Wo; a.stride = 2; a.pad = 0;
            a.out = d; a.ldo = C;
            TRY(run_conv(r, v.e_down[i].resample, a));
            cur = d; H = Ho; W = Wo;
        }
    }
    TRY(run_res(r, v.e_mid1, cur, nullptr, C, 0, B, H, W, 1e-6f, nullptr, 0, &o)); cur = o;
    TRY(run_vae_attn(r, v.e_attn, cur, B, H, W, &o)); cur = o;
    TRY(run_res(r, v.e_mid2, cur, nullptr, C, 0, B, H, W, 1e-6f, nullptr, 0, &o)); cur = o;
    half_t* tn = r.H((size_t)B * H * W * C);
    TRY(run_gn(r, v.e_norm_out, cur, nullptr, C, 0, B, H * W, 1e-6f, true, tn));
    {
        ConvArgs a;
        a.a0 = tn; a.c0 = C; a.B = B; a.Hi = H; a.Wi = W; a.Ho = H; a.Wo = W; a.pad = 1;
        a.out = out; a.flags = EP_NCHW; a.n_real = 2 * c.z_channels;
        TRY(run_conv(r, v.e_conv_out, a));
    }
    return 0;
}

int vae_decode(sdmi_engine* e, const void* z, int io_dtype, float* out, int B, int h, int w, hipStream_t s) {
    SDMI_REQUIRE(e->vae.ready, "vae not finalized");
    SDMI_CHECK_HIP(hipSetDevice(e->device));
    Run dry{e, s, true};
    e->arena.dry = true; e->arena.high = 0;
    TRY(vae_decode_run(dry, z, io_dtype, out, B, h, w));
    e->arena.dry = false;
    TRY(ensure_arena(e, e->arena.high, s));
    Run run{e, s, false};
    return vae_decode_run(run, z, io_dtype, out, B, h, w);
}

int vae_encode(sdmi_engine* e, const void* x, int io_dtype, float* out, int B, int H, int W, hipStream_t s) {
    SDMI_REQUIRE(e->vae.ready && e->vae.has_encoder, "vae encoder weights not loaded");
    SDMI_CHECK_HIP(hipSetDevice(e->device));
    Run dry{e, s, true};
    e->arena.dry = true; e->arena.high = 0;
    TRY(vae_encode_run(dry, x, io_dtype, out, B, H, W));
    e->arena.dry = false;
    TRY(ensure_arena(e, e->arena.high, s));
    Run run{e, s, false};
    return vae_encode_run(run, x, io_dtype, out, B, H, W);
}

// ------------------------------------------------------------------------------------------------------------
// CLIP text encoder
// ------------------------------------------------------------------------------------------------------------
static int clip_build(sdmi_engine* e, int slot) {
    ClipW& c = e->clip[slot];
    auto& m = e->raw_clip[slot];
    const sdmi_clip_config& cfg = c.cfg;
    const RawTensor* te = find_raw(m, "embeddings.token_embedding.weight");
    const RawTensor* pe = find_raw(m, "embeddings.position_embedding.weight");
    SDMI_REQUIRE(te && pe, "missing CLIP embeddings");
    SDMI_REQUIRE(te->shape.size() == 2 && te->shape[0] == cfg.vocab_size && te->shape[1] == cfg.hidden, "token_embedding shape");
    SDMI_REQUIRE(pe->shape.size() == 2 && pe->shape[0] == cfg.max_positions && pe->shape[1] == cfg.hidden, "position_embedding shape");
    TRY(dev_alloc(e, &c.tok_emb, te->bytes));
    SDMI_CHECK_HIP(hipMemcpy(c.tok_emb, te->ptr, te->bytes, hipMemcpyDeviceToDevice));
    c.tok_dtype = te->dtype;
    TRY(dev_alloc(e, (void**)&c.pos_emb, (size_t)cfg.max_positions * cfg.hidden * sizeof(float)));
    TRY(launch_convert_to_f32(pe->ptr, pe->dtype, c.pos_emb, (int64_t)cfg.max_positions * cfg.hidden, 0));
    c.layers.clear();
    for (int i = 0; i < cfg.layers; ++i) {
        const std::string b = "encoder.layers." + std::to_string(i);
        ClipLayerW L;
        TRY(pack_norm(e, m, b + ".layer_norm1", &L.ln1));
        TRY(pack_norm(e, m, b + ".layer_norm2", &L.ln2));
        TRY(pack_stack(e, m, {b + ".self_attn.q_proj", b + ".self_attn.k_proj"}, true, false, &L.qk));
        TRY(pack_one(e, m, b + ".self_attn.v_proj", true, false, &L.v));
        TRY(pack_one(e, m, b + ".self_attn.out_proj", true, false, &L.o));
        TRY(pack_one(e, m, b + ".mlp.fc1", true, false, &L.fc1));
        TRY(pack_one(e, m, b + ".mlp.fc2", true, false, &L.fc2));
        c.layers.push_back(L);
    }
    TRY(pack_norm(e, m, "final_layer_norm", &c.final_ln));
    // optional pooled-output projection, given in nn.Linear layout [proj_dim][hidden] (transformers' text_projection.weight;
    // open_clip stores the transpose, the host converts): kept as plain fp16 rows for the small-M linear kernel
    c.text_proj = nullptr; c.proj_dim = 0;
    if (const RawTensor* tp = find_raw(m, "text_projection.weight")) {
        SDMI_REQUIRE(tp->shape.size() == 2 && tp->shape[1] == cfg.hidden && tp->shape[1] % 8 == 0, "text_projection must be [proj][hidden]");
        ConvW tmp;
        TRY(pack_one(e, m, "text_projection", false, false, &tmp));
        c.text_proj = tmp.w; c.proj_dim = (int)tp->shape[0];
    }
    SDMI_CHECK_HIP(hipDeviceSynchronize());
    c.ready = true;
    return 0;
}

static int clip_run(Run& r, const ClipW& c, const int* tokens, const float* inputs_embeds, int B, int L, int skip,
                    int apply_final_ln, float* out, float* pooled) {
    const sdmi_clip_config& cfg = c.cfg;
    const int C = cfg.hidden, H = cfg.heads, D = C / H;
    const size_t M = (size_t)B * L;
    const int Lpad = rup(L, 64);
    r.e->arena.reset();
    half_t* cur = r.H(M * C);
    if (!r.dry) TRY(launch_clip_embed(tokens, c.tok_emb, c.tok_dtype, c.pos_emb, inputs_embeds, cur, B, L, C, cfg.vocab_size, r.s));
    // `out` taps the residual stream after block layers-skip+1; the pooled row always comes from the LAST block + final norm
    // (transformers' pooler_output; open_clip's pool(ln_final(x)) @ text_projection), so run on when it is requested
    const int ntap = cfg.layers - skip + 1;
    const int nrun = pooled ? cfg.layers : ntap;
    half_t* tap = nullptr;
    for (int i = 0; i < nrun; ++i) {
        if (i == ntap) tap = cur;
        const ClipLayerW& w = c.layers[i];
        half_t* n1 = r.H(M * C);
        if (!r.dry) TRY(launch_layernorm(cur, w.ln1.g, w.ln1.b, n1, (int64_t)M, C, cfg.eps, r.s));
        half_t* qk = r.H(M * 2 * C);
        TRY(run_linear(r, w.qk, n1, (int)M, nullptr, qk, 2 * C));
        half_t* vt = r.H((size_t)B * C * Lpad);
        TRY(run_vt(r, w.v, n1, C, B, L, Lpad, vt, true));
        half_t* a = r.H(M * C);
        if (!r.dry) {
            AttnP p{};
            p.q = qk; p.k = qk + C; p.vt = vt; p.out = a;
            p.B = B; p.H = H; p.N = L; p.M = L; p.D = D;
            p.ldq = 2 * C; p.ldk = 2 * C; p.vt_ld = Lpad; p.ldo = C;
            p.scale_log2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
            p.causal = 1;
            TRY(launch_attention(p, r.e->force_generic, r.s));
        }
        half_t* x1 = r.H(M * C);
        TRY(run_linear(r, w.o, a, (int)M, cur, x1, C));
        half_t* n2 = r.H(M * C);
        if (!r.dry) TRY(launch_layernorm(x1, w.ln2.g, w.ln2.b, n2, (int64_t)M, C, cfg.eps, r.s));
        half_t* hmid = r.H(M * cfg.intermediate);
        {
            ConvArgs ca;
            ca.a0 = n2; ca.c0 = w.fc1.cin_pad;
            ca.B = 1; ca.Hi = (int)M; ca.Wi = 1; ca.Ho = (int)M; ca.Wo = 1;
            ca.out = hmid; ca.ldo = cfg.intermediate;
            ca.flags = cfg.act == 0 ? EP_QUICK_GELU : EP_GELU;
            TRY(run_conv(r, w.fc1, ca));
        }
        half_t* x2 = r.H(M * C);
        TRY(run_linear(r, w.fc2, hmid, (int)M, x1, x2, C));
        cur = x2;
    }
    if (tap == nullptr) tap = cur;                           // ntap == nrun
    half_t* fin = tap;
    if (apply_final_ln) {
        fin = r.H(M * C);
        if (!r.dry) TRY(launch_layernorm(tap, c.final_ln.g, c.final_ln.b, fin, (int64_t)M, C, cfg.eps, r.s));
    }
    half_t* last_ln = nullptr;
    float* pool_raw = nullptr;
    if (pooled) {
        last_ln = (apply_final_ln && tap == cur) ? fin : r.H(M * C);
        if (c.text_proj) pool_raw = r.F((size_t)B * C);
    }
    if (r.dry) return 0;
    TRY(launch_convert_to_f32(fin, SDMI_F16, out, (int64_t)M * C, r.s));
    if (pooled) {
        if (last_ln != fin) TRY(launch_layernorm(cur, c.final_ln.g, c.final_ln.b, last_ln, (int64_t)M, C, cfg.eps, r.s));
        if (c.text_proj) {
            TRY(launch_clip_pool(tokens, last_ln, pool_raw, B, L, C, r.s));
            TRY(launch_small_linear(pool_raw, c.text_proj, nullptr, nullptr, pooled, B, c.proj_dim, C, C, c.proj_dim, false, false, r.s));
        } else {
            TRY(launch_clip_pool(tokens, last_ln, pooled, B, L, C, r.s));
        }
    }
    return 0;
}

int engine_clip_configure(sdmi_engine* e, int slot, const sdmi_clip_config* cfg) {
    SDMI_REQUIRE(slot == 0 || slot == 1, "clip slot must be 0 or 1");
    SDMI_REQUIRE(cfg->hidden % 64 == 0 && cfg->intermediate % 64 == 0 && cfg->heads > 0 && cfg->hidden % cfg->heads == 0,
                 "clip: hidden and intermediate must be multiples of 64");
    SDMI_REQUIRE(cfg->layers >= 1 && cfg->max_positions >= 1 && cfg->vocab_size >= 1, "clip: bad config");
    e->clip[slot].cfg = *cfg;
    e->clip[slot].configured = true;
    e->clip[slot].ready = false;
    return 0;
}
int engine_clip_load_tensor(sdmi_engine* e, int slot, const char* key, const void* data, int dtype, int ndim,
                            const int64_t* shape, int on_device) {
    SDMI_REQUIRE(slot == 0 || slot == 1, "clip slot must be 0 or 1");
    SDMI_CHECK_HIP(hipSetDevice(e->device));
    return load_raw(e->raw_clip[slot], key, data, dtype, ndim, shape, on_device);
}
int engine_clip_finalize(sdmi_engine* e, int slot) {
    SDMI_REQUIRE((slot == 0 || slot == 1) && e->clip[slot].configured, "clip slot not configured");
    SDMI_CHECK_HIP(hipSetDevice(e->device));
    const int rc = clip_build(e, slot);
    free_raw(e->raw_clip[slot]);
    return rc;
}
int engine_clip_forward(sdmi_engine* e, int slot, const int* tokens, const float* inputs_embeds, int B, int L, int skip,
                        int apply_final_ln, float* out, float* pooled, hipStream_t s) {
    SDMI_REQUIRE((slot == 0 || slot == 1) && e->clip[slot].ready, "clip not finalized");
    const ClipW& c = e->clip[slot];
    SDMI_REQUIRE(B >= 1 && L >= 1 && L <= c.cfg.max_positions, "clip: 1 <= L <= max_positions");
    SDMI_REQUIRE(skip >= 1 && skip <= c.cfg.layers, "clip: 1 <= skip <= layers");
    SDMI_REQUIRE(tokens != nullptr && out != nullptr, "null argument");
    SDMI_CHECK_HIP(hipSetDevice(e->device));
    Run dry{e, s, true};
    e->arena.dry = true; e->arena.high = 0;
    TRY(clip_run(dry, c, tokens, inputs_embeds, B, L, skip, apply_final_ln, out, pooled));
    e->arena.dry = false;
    TRY(ensure_arena(e, e->arena.high, s));
    Run run{e, s, false};
    return clip_run(run, c, tokens, inputs_embeds, B, L, skip, apply_final_ln, out, pooled);
}

int engine_load_unet_tensor(sdmi_engine* e, const char* key, const void* data, int dtype, int ndim, const int64_t* shape,
                            int on_device) {
    SDMI_CHECK_HIP(hipSetDevice(e->device));
    return load_raw(e->raw_unet, key, data, dtype, ndim, shape, on_device);
}
int engine_load_vae_tensor(sdmi_engine* e, const char* key, const void* data, int dtype, int ndim, const int64_t* shape,
                           int on_device) {
    SDMI_CHECK_HIP(hipSetDevice(e->device));
    return load_raw(e->raw_vae, key, data, dtype, ndim, shape, on_device);
}
int engine_unet_finalize(sdmi_engine* e) {
    SDMI_CHECK_HIP(hipSetDevice(e->device));
    e->unet_sites.clear();
    e->unet_vec_sites.clear();
    e->recording_unet_sites = true;
    const int rc = unet_build(e);
    e->recording_unet_sites = false;
    free_raw(e->raw_unet);
    return rc;
}
// Re-pack one conv / linear weight of the finalized UNet in place (same shape as at load time).  The caller re-runs
// sdmi_unet_set_context afterwards: cached cross-attention K / V^T depend on attn2.to_k / to_v.
int engine_unet_update_weight(sdmi_engine* e, const char* key, const void* data, int dtype, int ndim, const int64_t* shape,
                              int on_device) {
    SDMI_CHECK_HIP(hipSetDevice(e->device));
    SDMI_REQUIRE(e->unet.ready, "unet not finalized");
    SDMI_REQUIRE(key && data && (ndim == 2 || ndim == 4), "bad tensor arguments");
    SDMI_REQUIRE(dtype == SDMI_F16 || dtype == SDMI_F32, "dtype must be SDMI_F16 or SDMI_F32");
    auto it = e->unet_sites.find(key);
    SDMI_REQUIRE(it != e->unet_sites.end(), std::string("no packed conv / linear weight named ") + key);
    const int O = (int)shape[0], I = (int)shape[1], kh = ndim == 4 ? (int)shape[2] : 1;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    const void* src = data;
    void* tmp = nullptr;
    if (!on_device) {
        const size_t bytes = n * (dtype == SDMI_F16 ? 2 : 4);
        SDMI_CHECK_HIP(hipMalloc(&tmp, bytes));
        SDMI_CHECK_HIP(hipMemcpy(tmp, data, bytes, hipMemcpyHostToDevice));
        src = tmp;
    }
    int rc = 0;
    for (const auto& st : it->second) {
        if (st.O != O || st.cin != I || st.kh != kh || (ndim == 4 && shape[3] != kh)) {
            set_error(std::string("shape of ") + key + " differs from the loaded weight");
            rc = 1;
            break;
        }
        rc = launch_pack_conv_weight(src, dtype, st.dst, O, I, kh, kh, st.Opad, st.cin_pad, st.geglu, 0);
        if (rc) break;
    }
    SDMI_CHECK_HIP(hipDeviceSynchronize());
    if (tmp) (void)hipFree(tmp);
    e->ctx_valid = false;                                     // cached K / V^T depend on attn2.to_k / to_v
    ++e->weights_epoch;                                       // LayerNorm-folded copies of this weight are stale
    return rc;
}
// ---- hypernetwork hand-over ------------------------------------------------------------------------------------------------
int engine_hypernet_clear(sdmi_engine* e) {
    SDMI_CHECK_HIP(hipSetDevice(e->device));
    SDMI_CHECK_HIP(hipDeviceSynchronize());
    for (void* p : e->owned_hn) (void)hipFree(p);
    e->owned_hn.clear();
    e->hypernets.clear();
    e->ctx_valid = false;                                     // cached context K / V^T depend on the loaded hypernetworks
    return 0;
}
int engine_hypernet_begin(sdmi_engine* e, float multiplier) {
    HnNet n;
    n.multiplier = multiplier;
    e->hypernets.push_back(n);
    e->ctx_valid = false;
    return 0;
}
static HnModule* hn_module(sdmi_engine* e, int dim, int which) {
    if (e->hypernets.empty()) return nullptr;
    auto& pr = e->hypernets.back().by_dim[dim];
    return which ? &pr.second : &pr.first;
}
int engine_hypernet_linear(sdmi_engine* e, int dim, int which, const void* w, const void* b, int dtype, int out_f, int in_f, int on_device) {
    SDMI_CHECK_HIP(hipSetDevice(e->device));
    HnModule* m = hn_module(e, dim, which);
    SDMI_REQUIRE(m != nullptr, "sdmi_unet_hypernet_begin first");
    SDMI_REQUIRE(w && out_f > 0 && in_f > 0 && in_f % 8 == 0, "hypernetwork linear: bad shape");
    std::map<std::string, RawTensor> tmp;
    const int64_t ws[2] = {out_f, in_f}, bs[1] = {out_f};
    TRY(load_raw(tmp, "l.weight", w, dtype, 2, ws, on_device));
    if (b) TRY(load_raw(tmp, "l.bias", b, dtype, 1, bs, on_device));
    HnOp op;
    op.kind = 0;
    e->alloc_sink = &e->owned_hn;
    const int rc = pack_one(e, tmp, "l", true, false, &op.lin);
    e->alloc_sink = nullptr;
    SDMI_CHECK_HIP(hipDeviceSynchronize());
    free_raw(tmp);
    TRY(rc);
    m->ops.push_back(op);
    return 0;
}
int engine_hypernet_act(sdmi_engine* e, int dim, int which, int act) {
    HnModule* m = hn_module(e, dim, which);
    SDMI_REQUIRE(m != nullptr, "sdmi_unet_hypernet_begin first");
    SDMI_REQUIRE(act >= 1 && act <= 15, "unknown hypernetwork activation");
    HnOp op;
    op.kind = 1; op.act = act;
    m->ops.push_back(op);
    return 0;
}
int engine_hypernet_layernorm(sdmi_engine* e, int dim, int which, const void* g, const void* b, int dtype, int n, int on_device) {
    SDMI_CHECK_HIP(hipSetDevice(e->device));
    HnModule* m = hn_module(e, dim, which);
    SDMI_REQUIRE(m != nullptr, "sdmi_unet_hypernet_begin first");
    SDMI_REQUIRE(g && b && n % 64 == 0 && n <= 3072, "hypernetwork LayerNorm: width must be a multiple of 64 and <= 3072");
    std::map<std::string, RawTensor> tmp;
    const int64_t sh[1] = {n};
    TRY(load_raw(tmp, "n.weight", g, dtype, 1, sh, on_device));
    TRY(load_raw(tmp, "n.bias", b, dtype, 1, sh, on_device));
    HnOp op;
    op.kind = 2;
    e->alloc_sink = &e->owned_hn;
    const int rc = pack_norm(e, tmp, "n", &op.ln);
    e->alloc_sink = nullptr;
    SDMI_CHECK_HIP(hipDeviceSynchronize());
    free_raw(tmp);
    TRY(rc);
    m->ops.push_back(op);
    return 0;
}

// Replace one 1-D parameter of the finalized UNet (a conv / linear bias, a GroupNorm / LayerNorm gain or shift) in place.
int engine_unet_update_vector(sdmi_engine* e, const char* key, const void* data, int dtype, int64_t n, int on_device) {
    SDMI_CHECK_HIP(hipSetDevice(e->device));
    SDMI_REQUIRE(e->unet.ready, "unet not finalized");
    SDMI_REQUIRE(key && data && n > 0, "bad arguments");
    SDMI_REQUIRE(dtype == SDMI_F16 || dtype == SDMI_F32, "dtype must be SDMI_F16 or SDMI_F32");
    auto it = e->unet_vec_sites.find(key);
    SDMI_REQUIRE(it != e->unet_vec_sites.end(), std::string("no bias / norm parameter named ") + key);
    const void* src = data;
    void* tmp = nullptr;
    if (!on_device) {
        const size_t bytes = (size_t)n * (dtype == SDMI_F16 ? 2 : 4);
        SDMI_CHECK_HIP(hipMalloc(&tmp, bytes));
        SDMI_CHECK_HIP(hipMemcpy(tmp, data, bytes, hipMemcpyHostToDevice));
        src = tmp;
    }
    int rc = 0;
    for (const auto& st : it->second) {
        if (st.n != (int)n) { set_error(std::string("length of ") + key + " differs from the loaded parameter"); rc = 1; break; }
        rc = launch_pack_bias(src, dtype, st.dst, st.n, st.n_pad, st.geglu, 0);
        if (rc) break;
    }
    SDMI_CHECK_HIP(hipDeviceSynchronize());
    if (tmp) (void)hipFree(tmp);
    ++e->weights_epoch;                                       // a LayerNorm gain / shift or a folded bias may have changed
    return rc;
}
int engine_vae_finalize(sdmi_engine* e) {
    SDMI_CHECK_HIP(hipSetDevice(e->device));
    // a second finalize replaces the first stage (external VAE file / checkpoint reload): drop the previous packed weights
    SDMI_CHECK_HIP(hipDeviceSynchronize());
    for (void* p : e->owned_vae) (void)hipFree(p);
    e->owned_vae.clear();
    e->vae.ready = false;
    e->alloc_sink = &e->owned_vae;
    const int rc = vae_build(e);
    e->alloc_sink = nullptr;
    free_raw(e->raw_vae);
    return rc;
}
int engine_set_context(sdmi_engine* e, const void* ctx, int dtype, int Bn, int L, hipStream_t s, bool conditional) {
    SDMI_CHECK_HIP(hipSetDevice(e->device));
    return unet_set_context(e, ctx, dtype, Bn, L, s, conditional);
}

}  // namespace sdmi

sdmi_engine::~sdmi_engine() {
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    sdmi::free_raw(raw_unet);
    sdmi::free_raw(raw_vae);
    sdmi::free_raw(raw_clip[0]);
    sdmi::free_raw(raw_clip[1]);
    sdmi::ctx_free(this);
    for (void* p : owned) (void)hipFree(p);
    for (void* p : owned_vae) (void)hipFree(p);
    for (void* p : owned_hn) (void)hipFree(p);
    if (hn_ctx_scratch) (void)hipFree(hn_ctx_scratch);
    if (arena.base) (void)hipFree(arena.base);
    for (auto& a : aux_arenas) if (a.base) (void)hipFree(a.base);
    for (hipStream_t st : aux_streams) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : ev_join) (void)hipEventDestroy(ev);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
}
