// rowchain.hip — the feed-forward chain of a BasicTransformerBlock as ONE launch (gfx950; fp16 operands, fp32 accumulate).
//
// Everything a transformer block does after its self-attention is local to a token row
// (/root/reference/modules/sd_hijack_unet.py:83-102 around ldm's BasicTransformerBlock._forward); this file fuses the last third,
//     x3 = x2 + W2 ( (W1v LN3(x2) + b1v) * gelu(W1g LN3(x2) + b1g) ) + b2    (ff: GEGLU -> Linear)
// As separate launches (LayerNorm, GEGLU GEMM, GEMM) the 42 MB token stream of the 320-wide level is written and re-read three times and
// the 4C-wide hidden tensor (168 MB) is written and read once.  Here a workgroup owns 128 whole rows and the chain never leaves the CU.
//
// The chain has the shape of flash attention WITHOUT the online softmax, with the row's width C as the "head size":
//     stage 1   S^T[u][row] = B1_j[u][:] . n[row][:]          u = the 32 rows of a B1 unit, contraction over C
//     middle    P = f(S)    per row, inside one lane           (GEGLU)
//     stage 2   O^T[c][row] += B2_j[c][u] P[u][row]           c = 0 .. C-1, contraction over the unit's 32 columns
// and uses attention.hip's formulation: v_mfma_f32_32x32x16_f16 with the MATRIX rows coming from LDS as the A operand and the token
// rows as the B operand held in registers — a lane owns one token row (q = lane & 31) and half of its columns, so LayerNorm and GEGLU
// need no cross-lane traffic beyond one v_permlane32_swap, and P feeds stage 2 straight from the accumulator registers (B1 rows are
// read with bits 2/3 of the row index swapped, so the 8 values a lane packs are 8 consecutive units: attention.hip header).
//   j = 40 chunks of 32 hidden units: B1 = the chunk's 32 value rows and its 32 gate rows of ff.net.0.proj, B2 = the chunk's 32 columns
//   of ff.net.2.  Same flops as the two GEMMs.
//
// Register budget: the row fragments (C / 16 x h8 = 80 VGPRs), the O^T accumulators (C / 32 x 16 = 160) and the score blocks make
// this a one-wave-per-SIMD kernel (4 waves x 32 rows, up to 512 registers).  The operands are streamed by the same waves with LDS-direct
// loads: the packed operand stream in HBM IS the LDS image (padded rows, accumulator-init tables in the unit tails), so a stage is a
// linear copy of whole 1 KB pieces, one phase ahead of its use, one barrier per phase.  GEGLU of chunk j - 1 sits between the stage-1
// MFMAs of chunk j.
//
// Round 6 froze the file at this form.  Removed: the cross-attention chain (norm2 -> to_q -> attention over the 77 text
// keys -> to_out as one launch with to_q / to_out folded into per-image key / value matrices: 118-166 us in the forward against 115 us for
// the four launches it replaced), the 8-wave feed-forward form (two waves per SIMD: +5 %) and the component-removal timing variants —
// measurements and analysis stay in profiles/r05_rowchain_parts.txt, r05_fwd_ab_fuse_rows.txt, r05_fwd_ab_ff8.txt and DESIGN.md 9.2; the
// code is in the history (round 5).  What limits the chain — one in-order wave per SIMD, a fresh 1 KB LDS fragment per MFMA — is the
// design (one token row per lane), not the schedule, so neither register blocking nor a 640-wide instantiation was built.
#include "common.h"
#include "prof.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace sdmi {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS / packed-stream geometry for row width C
template <int C>
struct RcGeo {
    static constexpr int NDC = C / 16, NDB = C / 32;
    static constexpr int B1STR = C * 2 + 16;                  // bytes per B1 row: an odd number of 16-byte slots (conflict-free ds_read_b128)
    static constexpr int B1TAB = 32 * B1STR;                  // the unit's 32 accumulator-init values (fp32: bias, or 0 / -inf key mask)
    static constexpr int B1UNIT = (B1TAB + 128 + 1023) / 1024 * 1024;
    static constexpr int B2STR = 80;                          // 32 halfs + 16 bytes: 5 slots
    static constexpr int B2UNIT = C * B2STR;
    static constexpr int FFPACK = (2 * B1UNIT + B2UNIT + 4095) / 4096 * 4096;            // feed-forward pack (whole pieces for all 4 waves)
    static_assert(C % 64 == 0, "row width must be a multiple of 64 (whole 1 KB pieces)");
};

struct RowChainP {
    const half_t* x;          // [M][C] token stream: LayerNorm input and residual
    half_t* out;              // [M][C]
    const float* gamma;       // LayerNorm affine [C]
    const float* beta;
    const char* packs;        // packed operand stream (rowchain_ff_pack_kernel)
    const float* bias_out;    // [C] (the launcher substitutes zeros for a null pointer)
    int M;                    // rows, a multiple of 128
    int nchunk;               // hidden / 32
    float eps;
    const half_t* zero;       // the zero page (padding lanes of the row staging)
};

__device__ __forceinline__ float rc_gelu_erf(float g) {      // gemm.hip's gelu_erf (exact-erf GELU, A&S 7.1.26), instruction for instruction
    const float x = fabsf(g) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(x * x * -1.4426950408889634f);
    const float erf_abs = fmaf(-(poly * t), e, 1.0f);
    const float erf_v = copysignf(erf_abs, g);
    const float hg = 0.5f * g;
    return fmaf(hg, erf_v, hg);
}

// both half-waves' values of a per-row scalar: returns (value of lane q, value of lane q + 32) in every lane of the pair
__device__ __forceinline__ void rc_pair(float v, float& lo, float& hi) {
    lo = v; hi = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
}

// The wave's 32 token rows -> its LDS staging region (one B1UNIT: rows padded to B1STR, the conflict-free stride of the row-per-lane
// fragment reads), as 21 coalesced 1 KB LDS-direct pieces.  (Row-per-lane global accesses — each lane its own 640-byte row — made the
// prologue + epilogue 37 us of the launch: 32 partial lines per load instruction, 8-byte partial-line stores.)
template <int C>
__device__ __forceinline__ void rc_stage_rows(const half_t* xwave, char* region, int lane, const half_t* zero) {
    typedef RcGeo<C> G;
    constexpr int SLOTS = G::B1STR / 16, NPIECE = G::B1UNIT / 1024;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int n = i * 64 + lane;
        const int r = (n * 1599) >> 16;                     // n / 41 for n < 1344
        const int c = n - r * SLOTS;
        const half_t* src = (r < 32 && c < C / 8) ? xwave + r * C + c * 8 : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(region + i * 1024), 16, 0, 0);
    }
    static_assert(SLOTS == 41, "the division constant above is for 41 slots per row");
}

// The lane's half of its staged row -> LayerNorm -> fp16 fragments (B operand of stage 1): xf[dc] = n[row][dc*16 + half*8 .. +8).
// Two-pass statistics from the fp16 values, as norm.hip's layernorm_kernel.
template <int C>
__device__ __forceinline__ void rc_load_ln(const char* region, int lq, const float* gamma, const float* beta, float eps, int half,
                                           h8 (&xf)[C / 16]) {
    constexpr int NDC = C / 16;
    const char* xrow = region + lq * RcGeo<C>::B1STR + half * 16;
#pragma unroll
    for (int dc = 0; dc < NDC; ++dc) xf[dc] = *reinterpret_cast<const h8*>(xrow + dc * 32);
    float s = 0.f;
#pragma unroll
    for (int dc = 0; dc < NDC; ++dc)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)xf[dc][e];
    float lo, hi;
    rc_pair(s, lo, hi);
    const float inv_c = 1.0f / (float)C;
    const float mean = (lo + hi) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int dc = 0; dc < NDC; ++dc)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = (float)xf[dc][e] - mean; q = fmaf(d, d, q); }
    rc_pair(q, lo, hi);
    const float rstd = rsqrtf(fmaf(lo + hi, inv_c, eps));
#pragma unroll
    for (int dc = 0; dc < NDC; ++dc) {
        const int c0 = dc * 16 + half * 8;
        const f4 g0 = *reinterpret_cast<const f4*>(gamma + c0), g1 = *reinterpret_cast<const f4*>(gamma + c0 + 4);
        const f4 b0 = *reinterpret_cast<const f4*>(beta + c0), b1 = *reinterpret_cast<const f4*>(beta + c0 + 4);
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float g = e < 4 ? g0[e] : g1[e - 4], bb = e < 4 ? b0[e] : b1[e - 4];
            o[e] = (half_t)(((float)xf[dc][e] - mean) * rstd * g + bb);
        }
        xf[dc] = o;
    }
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <int... Is, typename F>
__device__ __forceinline__ void rc_static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void rc_static_for(F&& f) { rc_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

__device__ __forceinline__ h8 rc_lds(const char* p) { return *reinterpret_cast<const h8*>(p); }

// accumulator init of a stage-1 block from its unit's table: sc[r] <-> unit row 16 * (r >> 3) + 8 * half + (r & 7)
template <int C>
__device__ __forceinline__ f16v rc_init(const char* unit, int half) {
    const f4* tab = reinterpret_cast<const f4*>(unit + RcGeo<C>::B1TAB + half * 32);
    const f4 t0 = tab[0], t1 = tab[1], t2 = tab[4], t3 = tab[5];
    return f16v{t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3], t2[0], t2[1], t2[2], t2[3], t3[0], t3[1], t3[2], t3[3]};
}

// out[row][c] = O + bias + x:  o[db][r] is O[row][db*32 + (r & 3) + 8 * (r >> 2) + 4 * half].  Through the wave's LDS staging region:
// the residual rows arrive as coalesced pieces (rc_stage_rows), every lane adds its accumulators to its row in place, and the finished
// rows leave as coalesced 16-byte stores.  Call with all waves past their last operand read (the regions overlay the operand buffers).
template <int C>
__device__ __forceinline__ void rc_store(const RowChainP& p, long wave_row0, char* region, int lane, const f16v (&o)[C / 32]) {
    typedef RcGeo<C> G;
    constexpr int SLOTS = G::B1STR / 16, NPIECE = G::B1UNIT / 1024;
    const int half = lane >> 5, lq = lane & 31;
    rc_stage_rows<C>(p.x + wave_row0 * C, region, lane, p.zero);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    char* xrow = region + lq * G::B1STR;
#pragma unroll
    for (int db = 0; db < C / 32; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = db * 32 + g * 8 + half * 4;
            const h4 res = *reinterpret_cast<const h4*>(xrow + c0 * 2);
            const f4 bb = *reinterpret_cast<const f4*>(p.bias_out + c0);
            h4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (half_t)((o[db][g * 4 + e] + bb[e]) + (float)res[e]);
            *reinterpret_cast<h4*>(xrow + c0 * 2) = v;
        }
    __builtin_amdgcn_wave_barrier();
    half_t* owave = p.out + wave_row0 * C;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int n = i * 64 + lane;
        const int r = (n * 1599) >> 16;
        const int c = n - r * SLOTS;
        const h8 v = *reinterpret_cast<const h8*>(region + n * 16);
        if (r < 32 && c < C / 8) *reinterpret_cast<h8*>(owave + r * C + c * 8) = v;
    }
}

// The packed stream is the LDS image: piece k of this wave (1 KB pieces dealt round-robin to the 4 waves) is a lane-linear copy.
// ub = stream + wave * 1024 (wave-uniform: stays in SGPRs, the lane offset is the instruction's VGPR offset), l = LDS destination +
// wave * 1024 (wave-uniform).
template <int NP>
__device__ __forceinline__ void rc_issue_piece(const char* ub, unsigned lane16, char* l, int k) {
    static_assert(NP % 4 == 0, "streams are padded to whole rounds of the 4 waves");
    if (k < NP / 4) __builtin_amdgcn_global_load_lds((gptr_t)(ub + k * 4096 + lane16), (lptr_t)(l + k * 4096), 16, 0, 0);
}
template <int NP>
__device__ __forceinline__ void rc_issue(const char* src, char* dst, int wave, int lane) {
    const char* ub = src + wave * 1024;
    char* l = dst + wave * 1024;
#pragma unroll
    for (int k = 0; k < NP / 4; ++k) rc_issue_piece<NP>(ub, (unsigned)lane * 16u, l, k);
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void rc_phase_sync() {          // my loads have landed; everybody's have; everybody left the previous phase
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// A phase is a fixed sequence of steps, one MFMA each.  The compiler, left alone, issues each MFMA's LDS fragment read right in front of it
// and waits (lgkmcnt(0)) — one exposed LDS latency per MFMA, 4x the MFMA time with one wave per SIMD (first GPU run: 291 us for the
// feed-forward chain).  So the order is written out: the A fragments run PF steps ahead through a register ring, the LDS-direct loads of
// the next phase's operands are dealt one per third step, the middle op's VALU work is cut into per-step slices, and a
// sched_barrier closes every step.
constexpr int kRcPF = 8;

// ---------------------------------------------------------------------------------------------------------------
// feed-forward chain.  Pack j (j = 0 .. nchunk) = [B1 value unit of chunk j | B1 gate unit of chunk j | B2 unit of chunk j - 1]
// (the B1 units of pack nchunk and the B2 unit of pack 0 are zeros and never read).  Iteration j: S(j) (40 MFMAs) with GEGLU(j - 1)
// sliced between them, then O += B2(j - 1) P(j - 1) (20 MFMAs).  Two score sets alternate (template parity), two pack buffers in LDS;
// pack j + 1 is issued during iteration j.
// ---------------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256, 1) void rowchain_ff_kernel(RowChainP p) {
    typedef RcGeo<C> G;
    constexpr int PACK = G::FFPACK, NP = PACK / 1024, PF = kRcPF;
    constexpr int NDC = G::NDC, NDB = G::NDB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, lq = lane & 31;
    const long row = (long)blockIdx.x * 128 + wave * 32 + lq;

    // prologue: the wave's rows through its staging region in the second pack buffer (free until pack 1 is fetched), then pack 0
    char* const xreg = smem + PACK + wave * G::B1UNIT;
    static_assert(PACK + 4 * G::B1UNIT <= 160 * 1024, "row staging beside pack buffer 0");
    rc_stage_rows<C>(p.x + ((long)blockIdx.x * 128 + wave * 32) * C, xreg, lane, p.zero);
    rc_issue<NP>(p.packs, smem, wave, lane);
    wait_vm<NP / 4>();                                       // the rows have landed (loads retire in order); pack 0 may still be in flight
    __builtin_amdgcn_wave_barrier();
    h8 xf[NDC];
    rc_load_ln<C>(xreg, lq, p.gamma, p.beta, p.eps, half, xf);

    f16v o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    f16v sv[2], sg[2];                       // (value, gate) scores of chunks of even / odd index
    const int krow = (lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1);
    const int ka_off = krow * G::B1STR + half * 16;
    const int va_off = lq * G::B2STR + half * 16;
    const int nch = p.nchunk;
    const char* const gsrc = p.packs + wave * 1024;        // wave-uniform
    const unsigned lane16 = (unsigned)lane * 16u;

    // iteration j.  S1: chunk j exists (stage 1, and pack j + 1 is fetched); S2: chunk j - 1 exists (GEGLU + stage 2); PAR = j & 1
    constexpr int LE = 3;                                    // one operand piece every LE steps
    auto phase = [&](int j, auto s1c, auto s2c, auto parc) {
        constexpr bool S1 = decltype(s1c)::value, S2 = decltype(s2c)::value;
        constexpr int PAR = decltype(parc)::value;
        constexpr int N1 = S1 ? 2 * NDC : 0, N2 = S2 ? 2 * NDB : 0, NS = N1 + N2;
        rc_phase_sync();
        const char* reg = smem + (j & 1) * PACK;
        const char* b1 = reg + ka_off;
        const char* b2 = reg + 2 * G::B1UNIT + va_off;
        const char* gnext = gsrc + (long)(j + 1) * PACK;
        char* lnext = smem + ((j + 1) & 1) * PACK + wave * 1024;
        auto addr = [&](auto ic) -> const char* {
            constexpr int i = decltype(ic)::value;
            // consecutive MFMAs never share an accumulator (a filler between two MFMAs on the SAME accumulator costs +43 cycles,
            // MI355X_MICROARCH.md): stage 1 alternates the value / gate blocks, stage 2 walks the 10 output blocks per 16-column slice
            if constexpr (i < N1) return b1 + (i % 2) * G::B1UNIT + (i / 2) * 32;
            else return b2 + ((i - N1) % NDB) * 32 * G::B2STR + ((i - N1) / NDB) * 32;
        };
        h8 ring[PF];
        if constexpr (S1) {
            sv[PAR] = rc_init<C>(reg, half);
            sg[PAR] = rc_init<C>(reg + G::B1UNIT, half);
        }
        rc_static_for<PF>([&](auto ic) { if constexpr (decltype(ic)::value < NS) ring[decltype(ic)::value] = rc_lds(addr(ic)); });
        __builtin_amdgcn_sched_barrier(0);
        h8 pb[2];
        if constexpr (S2 && !S1) {           // last iteration: nothing to hide the GEGLU under
#pragma unroll
            for (int r = 0; r < 16; ++r) pb[r >> 3][r & 7] = (half_t)(SDMI_GELU_SIG ? geglu_gate(sv[PAR ^ 1][r], sg[PAR ^ 1][r]) : sv[PAR ^ 1][r] * rc_gelu_erf(sg[PAR ^ 1][r]));
        }
        rc_static_for<NS>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const h8 a = ring[i % PF];
            if constexpr (i < N1 && i % 2 == 0) sv[PAR] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, xf[i / 2], sv[PAR], 0, 0, 0);
            else if constexpr (i < N1) sg[PAR] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, xf[i / 2], sg[PAR], 0, 0, 0);
            else o[(i - N1) % NDB] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb[(i - N1) / NDB], o[(i - N1) % NDB], 0, 0, 0);
            if constexpr (i + PF < NS) ring[i % PF] = rc_lds(addr(std::integral_constant<int, (i + PF < NS ? i + PF : 0)>{}));
            if constexpr (S1 && i % LE == LE - 1) rc_issue_piece<NP>(gnext, lane16, lnext, i / LE);
            // GEGLU of chunk j - 1, element r = (i - 1) / 2 at the odd stage-1 steps 1 .. 31
            if constexpr (S1 && S2 && i % 2 == 1 && i / 2 < 16) {
                constexpr int r = i / 2;
                pb[r >> 3][r & 7] = (half_t)(SDMI_GELU_SIG ? geglu_gate(sv[PAR ^ 1][r], sg[PAR ^ 1][r]) : sv[PAR ^ 1][r] * rc_gelu_erf(sg[PAR ^ 1][r]));
            }
            __builtin_amdgcn_sched_barrier(0);               // a sched_barrier closes every step
        });
        if constexpr (S1) {                  // pieces the step loop had no slot for (NS / LE slots)
#pragma unroll
            for (int k = NS / LE; k < NP / 4; ++k) rc_issue_piece<NP>(gnext, lane16, lnext, k);
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    phase(0, T{}, F{}, P0{});
    int j = 1;
    for (; j + 1 < nch; j += 2) {            // two chunks per trip: no join between the parities, so no register shuffling
        phase(j, T{}, T{}, P1{});
        phase(j + 1, T{}, T{}, P0{});
    }
    if (j < nch) { phase(j, T{}, T{}, P1{}); ++j; }
    if (nch & 1) phase(nch, F{}, T{}, P1{});
    else phase(nch, F{}, T{}, P0{});
    rc_phase_sync();                                         // every wave is past its last operand read
    rc_store<C>(p, (long)blockIdx.x * 128 + wave * 32, smem + wave * G::B1UNIT, lane, o);
}

// ---------------------------------------------------------------------------------------------------------------
// pack kernels (one thread per 16-byte chunk of the stream image)
// ---------------------------------------------------------------------------------------------------------------
// feed-forward: w1 [2 * hidden][C] (rows of the value half, then of the gate half — or, `permuted`, the engine's GEGLU packing: groups
// of 64 rows = 32 value + 32 gate rows of one chunk, elementwise.hip geglu_row), b1 likewise, w2 [C][hidden] -> (hidden / 32 + 1) packs
template <int C>
__global__ __launch_bounds__(256) void rowchain_ff_pack_kernel(const half_t* w1, const float* b1, const half_t* w2, char* packs,
                                                               int hidden, int permuted) {
    typedef RcGeo<C> G;
    constexpr int PACK = G::FFPACK, NCHUNK16 = PACK / 16, U16 = G::B1UNIT / 16, SLOTS = G::B1STR / 16, V16 = G::B2UNIT / 16;
    const int nch = hidden / 32;
    const long total = (long)(nch + 1) * NCHUNK16;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int j = (int)(idx / NCHUNK16), ci = (int)(idx - (long)j * NCHUNK16);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (ci < 2 * U16) {
            const int u = ci / U16, cu = ci - u * U16;           // unit 0 = value rows, 1 = gate rows of the chunk's 32 hidden units
            const int r = cu / SLOTS, c = cu - r * SLOTS;
            // unit row rr -> (gate?, hidden unit of the chunk)
            auto src_of = [&](int rr) -> long {
                const int gate = u;
                const int hu = j * 32 + rr;
                return permuted ? (long)(hu >> 5) * 64 + (hu & 31) + 32 * gate : (long)gate * hidden + hu;
            };
            if (j < nch) {
                if (r < 32 && c < C / 8) {
                    v = *reinterpret_cast<const uint4*>(w1 + src_of(r) * C + c * 8);
                } else if (r >= 32 && cu * 16 >= G::B1TAB && cu * 16 < G::B1TAB + 128) {
                    const int t0 = (cu * 16 - G::B1TAB) / 4;       // table entries t0 .. t0 + 3
                    float f[4];
                    for (int e = 0; e < 4; ++e) f[e] = b1 ? b1[src_of(t0 + e)] : 0.f;
                    v = __builtin_bit_cast(uint4, f4{f[0], f[1], f[2], f[3]});
                }
            }
        } else {
            const int cb = ci - 2 * U16;
            const int n = cb / 5, c = cb - n * 5;
            const int jj = j - 1;                                // the B2 unit is skewed by one chunk (rowchain_ff_kernel)
            if (jj >= 0 && jj < nch && c < 4 && cb < V16) v = *reinterpret_cast<const uint4*>(w2 + (long)n * hidden + jj * 32 + c * 8);
        }
        *reinterpret_cast<uint4*>(packs + idx * 16) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------
bool rowchain_supports(int C) { return C == 320; }

size_t rowchain_ff_pack_bytes(int C, int hidden) {
    if (C != 320 || hidden % 32) return 0;
    typedef RcGeo<320> G;
    return (size_t)(hidden / 32 + 1) * G::FFPACK;
}

int launch_rowchain_ff_pack(const half_t* w1, const float* b1, const half_t* w2, void* packs, int C, int hidden, bool permuted,
                            hipStream_t s) {
    SDMI_REQUIRE(rowchain_supports(C) && hidden % 32 == 0 && hidden > 0, "rowchain feed-forward: C = 320, hidden % 32 == 0");
    const long total = (long)rowchain_ff_pack_bytes(C, hidden) / 16;
    hipLaunchKernelGGL(rowchain_ff_pack_kernel<320>, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, w1, b1,
                       w2, (char*)packs, hidden, permuted ? 1 : 0);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

template <typename K>
static int rc_set_smem(K kern, int bytes) {
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return 0;
}

int launch_rowchain_ff(const half_t* x, half_t* out, const float* gamma, const float* beta, const void* packs, const float* bias_out,
                       long rows, int C, int hidden, float eps, hipStream_t s) {
    SDMI_REQUIRE(rowchain_supports(C) && hidden % 32 == 0 && rows > 0 && rows % 128 == 0, "rowchain feed-forward: C = 320, rows % 128 == 0");
    typedef RcGeo<320> G;
    constexpr int SMEM = std::max(2 * G::FFPACK, G::FFPACK + 4 * G::B1UNIT);      // two pack buffers | pack 0 + the prologue's row staging
    void (*kern)(RowChainP) = rowchain_ff_kernel<320>;
    static PerDeviceOnce attr;                               // (per device: one process may drive several — common.h)
    if (attr.need() && rc_set_smem(kern, SMEM)) return 1;
    RowChainP p{};
    p.x = x; p.out = out; p.gamma = gamma; p.beta = beta; p.packs = (const char*)packs; p.bias_out = bias_out ? bias_out : reinterpret_cast<const float*>(zero_page()); p.zero = zero_page();
    p.M = (int)rows; p.nchunk = hidden / 32; p.eps = eps;
    ProfScope ps("rowchain_ff", 2.0 * rows * C * (3.0 * hidden), 4.0 * rows * C, s);
    hipLaunchKernelGGL(kern, dim3((unsigned)(rows / 128)), dim3(256), SMEM, s, p);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace sdmi
