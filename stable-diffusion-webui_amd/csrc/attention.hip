// attention.hip — fused QK^T*scale -> softmax -> PV for gfx950 (MI355X); fp16 in/out, fp32 scores and accumulators.
//
// Replaces the attention math of every CrossAttention.forward variant the webui can select
// (/root/reference/modules/sd_hijack_optimizations.py:221-281 Doggettx slicing, :508-546 sdp, :480-503 xformers,
// /root/reference/modules/hypernetworks/hypernetwork.py:382-407 baseline) with one LDS-tiled online-softmax kernel —
// the recurrence of /root/reference/modules/sub_quadratic_attention.py:54-113 with the KV chunk held in LDS and the
// running (max, sum, acc) in registers; the (B*H, N, M) score tensor never exists in HBM.
//
// Formulation chosen for CDNA4 (everything per-query stays in ONE lane, so no cross-lane softmax traffic):
//   S^T = K Q^T   : v_mfma_f32_32x32x16_f16, A = K rows from LDS (ds_read_b128), B = Q^T kept in registers.
//                   A lane (q = lane&31, half = lane>>5) ends up with 16+16 scores of its own query row.
//   O^T = V^T P^T : A = V^T rows from LDS, B = P^T = the lane's own exponentiated scores packed to fp16 —
//                   no LDS round trip and no shuffle for P.  The K rows are read with bits 2/3 of the row index
//                   swapped so the 8 scores a lane packs for one MFMA are 8 CONSECUTIVE keys, i.e. one aligned
//                   16-byte read of a V^T row.
//   V^T comes from the producer (the V projection GEMM writes [C][tokens] directly), so no transpose happens here.
//   Running max / sum / O^T rescale are per-lane scalars; the two half-waves exchange only the tile max (1 bpermute).
// Block = 4 waves x 32 queries = 128 queries; KV tile = 64 keys staged through registers into padded LDS rows
// (row stride = odd number of 16-byte slots => conflict-free ds_read_b128), next tile prefetched during compute.
#include "common.h"
#include "prof.h"
#include <cstdlib>

namespace sdmi {

typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

// Maximum of the NKB x 16 scores a lane holds, as a TREE of v_max3_f32 (depth 4 for 32 values) instead of the 16-deep dependent chain
// the running form `mx = fmaxf(mx, s[i])` compiles to: the resident waves of the flash kernels overlap each other so little (section
// timers, profiles/r03_attn_pp_sections.txt) that a wave's serial latency, not the issue slots, is what the softmax section costs.
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
template <int NKB>
__device__ __forceinline__ float tree_max(const f16v (&sc)[NKB]) {
    constexpr int N = NKB * 16;
    float v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = sc[i >> 4][i & 15];
    int n = N;
#pragma unroll
    for (int level = 0; level < 6; ++level) {
        if (n <= 1) break;
        const int m = (n + 2) / 3;
#pragma unroll
        for (int i = 0; i < m; ++i) {
            const int a = 3 * i, b = a + 1 < n ? a + 1 : a, c = a + 2 < n ? a + 2 : a;
            v[i] = max3f(v[a], v[b], v[c]);
        }
        n = m;
    }
    return v[0];
}

// VAR = forms of the d = 40 level-0 self-attention loop (docs/DESIGN_experiments.md, profiles/r02_attention_experiments.md); selected
// with SDMI_ATTN_OCC=<VAR> / sdmi_debug_set("attn_occ", VAR); 15 is the default for d = 40 (see g_attn_occ), handed on to 17 for key
// sequences of g_attn_fold_min_m (1024) and more since round 6:
//   0  round-1 kernel: register budget for 2 workgroups per CU (D <= 80) or 1; one ds_read -> wait -> MFMA chain per MFMA
//   5  128 VGPRs (4 workgroups per CU) + lazy rescale: the O accumulators are multiplied by alpha only when some lane's running
//      max moved (alpha == 1 otherwise: the result is unchanged; after the first few KV tiles the max rarely moves)
//  15  lazy rescale + all MFMA operand fragments of a phase read from LDS up front (the 6 K fragments before the first S^T MFMA, the
//      8 V^T fragments right after the last one, landing under the softmax): 14 exposed LDS latencies per KV tile become 2; the
//      half-wave max exchange is one v_permlane32_swap instead of a ds_bpermute round trip; 160 VGPRs, 3 workgroups per CU.
//      Same products in the same order per accumulator: bit-identical to 0 and 5 (tests/test_gpu_ops.py).
//  10..14, 18  (-DSDMI_ATTN_PARTS builds only, tools/gpu/attn_parts.py) 5 with one component removed / 15 with section timers
template <int D, int KVT, int VAR = 0>
__global__ __launch_bounds__(256, ((VAR == 15 || VAR == 16 || VAR == 17 || VAR == 18) ? 3 : (VAR == 5 || VAR >= 10) ? 4 : D <= 80 ? 2 : 1)) void attn_mfma_kernel(AttnP p) {
    constexpr bool LAZY_RESCALE = VAR >= 5;
    constexpr bool PREF = VAR == 15 || VAR == 16 || VAR == 17 || VAR == 18;
    // 17 = 15 with the softmax scale and shift folded into the S^T MFMA (head sizes with a spare contraction column: d = 40 -> 48), as in
    // attn_pp_kernel's FOLD form: Q is multiplied by scale * log2(e) when its fragments are loaded, K's padding column holds 1.0 and Q's
    // padding element -shift, so the accumulators ARE s * c - shift and exp2 applies to them directly: the 16 v_pk_fma_f32 per tile and
    // the per-tile alpha bookkeeping disappear (20 of the 92 VALU instructions of a tile).  shift is an fp16 number >= every score seen
    // so far (P <= 1); it is raised — scores re-based, O rescaled, Q's padding element rewritten — only when a tile's maximum exceeds it.
    constexpr bool FOLD = VAR == 17 && ((D + 15) / 16 * 16) > D;
    constexpr bool TREEMAX = VAR == 16;      // 16 = 15 with the tile maximum taken as a v_max3 tree (serial depth 4 instead of 16)
    constexpr bool TIMING = VAR == 18;       // s_memtime stamps around the sections of an iteration (AttnP::dbg)
    long long tm[6] = {0, 0, 0, 0, 0, 0};
    auto stamp = [&]() -> long long {
        if constexpr (!TIMING) return 0;
        const long long t = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return t;
    };
    // timing-only variants: results are wrong
    constexpr bool NO_EXP = VAR == 10, NO_S = VAR == 11, NO_PV = VAR == 12, NO_MAX = VAR == 13, NO_STAGE = VAR == 14;
    constexpr int DK = (D + 15) / 16 * 16;   // contraction length of S^T, padded to the MFMA K step
    constexpr int NDC = DK / 16;
    constexpr int DV = (D + 31) / 32 * 32;   // rows of O^T, padded to the MFMA M
    constexpr int NDB = DV / 32;
    constexpr int NKB = KVT / 32;            // 32-key blocks per KV tile
    constexpr int KSTR = DK * 2 + 16;        // bytes; (DK/8 + 1) slots of 16 B -> odd
    constexpr int VSTR = KVT * 2 + 16;       // (KVT/8 + 1) slots -> odd
    constexpr int K_BYTES = KVT * KSTR, V_BYTES = DV * VSTR;
    constexpr int TILE_BYTES = K_BYTES + V_BYTES;
    constexpr int KCPR = DK / 8;             // 16-byte chunks per K row
    constexpr int VCPR = KVT / 8;            // 16-byte chunks per V^T row
    constexpr int KCH = KVT * KCPR, VCH = DV * VCPR;
    constexpr int K_IT = (KCH + 255) / 256, V_IT = (VCH + 255) / 256;
    // When the head size leaves a spare (padding) row in O^T, V^T row D is held at 1.0 in LDS: the PV MFMA then accumulates
    // sum_k P[q][k] in O^T row D — the softmax denominator, rescaled together with O — and the VALU row sum disappears.
    constexpr bool SUMROW = DV > D;
    constexpr int L_RR = D - (D / 32) * 32, L_DB = D / 32;       // where that row lives in the 32x32 accumulator
    constexpr int L_HALF = (L_RR >> 2) & 1, L_R = (L_RR & 3) + 4 * (L_RR >> 3);
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 x (K tile | V^T tile)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, lq = lane & 31;
    // Workgroup -> (batch*head, query block).  Linear ids are dealt round-robin to the 8 XCDs; K/V of one head are read by
    // all of its query blocks, so every query block of a head is mapped to the SAME XCD (ids with equal id % 8) and the
    // head's K/V stay in that XCD's 4 MB L2 instead of being fetched by all eight (PMC: 3x algorithmic HBM reads before).
    int bh, qb;
    {
        const int nqb = gridDim.x, nbh = gridDim.y;
        const int id = blockIdx.y * nqb + blockIdx.x;
        if ((nbh & 7) == 0) {
            const int xcd = id & 7, slot = id >> 3;
            qb = slot % nqb;
            bh = (slot / nqb) * 8 + xcd;
        } else {
            bh = blockIdx.y; qb = blockIdx.x;
        }
    }
    const int b = bh / p.H, h = bh - b * p.H;
    const int q = qb * 128 + wave * 32 + lq;
    const bool qok = q < p.N;

    // ---- Q^T fragments (B operand of S^T): lane holds Q[q][dc*16 + half*8 .. +8) -----------------------------
    h8 qf[NDC];
    {
        const half_t* qptr = p.q + ((long)b * p.N + (qok ? q : 0)) * p.ldq + h * D;
#pragma unroll
        for (int dc = 0; dc < NDC; ++dc) {
            const int d = dc * 16 + half * 8;
            h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (qok && d < D) v = *reinterpret_cast<const h8*>(qptr + d);
            if constexpr (FOLD) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * p.scale_log2);
            }
            qf[dc] = v;
        }
    }
    // FOLD: element (D & 7) of the fragment that covers column D carries -shift (lanes of the half-wave that holds that column)
    constexpr int F_DC = D / 16, F_HALF = (D >> 3) & 1, F_E = D & 7;
    float shift = 0.f;

    const half_t* kbase = p.k + (long)b * p.M * p.ldk + h * D;
    const half_t* vbase = p.vt + ((long)b * p.H + h) * D * (long)p.vt_ld;

    // Staging is branch-free in the steady state: addresses are clamped instead of predicating the loads with a zero fill
    // (the select / phi copies of the predicated form cost more VALU issue slots than the softmax itself).
    //   K rows >= M  : re-read row M-1; their scores are masked in the (peeled) tail tile.
    //   K cols >= D  : re-read chunk 0; Q is zero there, so any finite value contributes 0.
    //   V^T rows >= D: loaded from a clamped row and not written to LDS; they only feed O^T rows that are not stored
    //                  (row D is the ones row, see SUMROW).
    //   V^T cols     : key0 + 8c < vt_ld holds for 64-key tiles (vt_ld >= M rounded up to 64); clamped for wider tiles.
    u4v kr[K_IT], vr[V_IT];
    auto load_tile = [&](int t) {
        const int key0 = t * KVT;
#pragma unroll
        for (int it = 0; it < K_IT; ++it) {
            const int idx = it * 256 + tid;
            const int row = idx / KCPR, c = idx - row * KCPR;
            // threads past the end of the tile (idx >= KCH) load a clamped row too; only their LDS write is skipped
            const int rr = min(key0 + row, p.M - 1);
            const int cc = (DK == D || c * 8 < D) ? c : 0;
            kr[it] = *reinterpret_cast<const u4v*>(kbase + (long)rr * p.ldk + cc * 8);
        }
#pragma unroll
        for (int it = 0; it < V_IT; ++it) {
            const int idx = it * 256 + tid;
            const int row = idx / VCPR, c = idx - row * VCPR;
            int col = key0 + c * 8;
            if (KVT > 64) col = min(col, p.vt_ld - 8);
            vr[it] = *reinterpret_cast<const u4v*>(vbase + (long)min(row, D - 1) * p.vt_ld + col);
        }
    };
    auto write_tile = [&](int buf) {
        char* Ks = smem + buf * TILE_BYTES;
        char* Vs = Ks + K_BYTES;
#pragma unroll
        for (int it = 0; it < K_IT; ++it) {
            const int idx = it * 256 + tid;
            const int row = idx / KCPR, c = idx - row * KCPR;
            if ((KCH % 256 == 0 || it + 1 < K_IT || idx < KCH) && !(FOLD && c * 8 >= D)) *reinterpret_cast<u4v*>(Ks + row * KSTR + c * 16) = kr[it];
        }
#pragma unroll
        for (int it = 0; it < V_IT; ++it) {
            const int idx = it * 256 + tid;
            const int row = idx / VCPR, c = idx - row * VCPR;
            if (row < D) *reinterpret_cast<u4v*>(Vs + row * VSTR + c * 16) = vr[it];
        }
    };

    f16v o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    // running max is kept in the scaled (log2) domain: t = s * scale_log2
    float m_run = -1e30f;
    f2v l_run = {0.f, 0.f};

    // K row read by this lane as MFMA row (lane&31): bits 2 and 3 swapped (see header)
    const int krow = (lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1);
    const int ka_off = krow * KSTR + half * 16;
    const int va_off = K_BYTES + lq * VSTR + half * 16;

    const int ntiles = (p.M + KVT - 1) / KVT;
    const int nfull = p.M / KVT;
    load_tile(0);
    write_tile(0);
    if constexpr (FOLD) {
        // K columns D..DK-1 of both buffers: column D = 1.0, the rest 0 — written once (write_tile skips these chunks)
        static_assert(!FOLD || (D % 8 == 0 && DK - D == 8), "one padding chunk per K row");
        for (int i = tid; i < 2 * KVT; i += 256) {
            const int buf = i / KVT, row = i - buf * KVT;
            *reinterpret_cast<uint4*>(smem + buf * TILE_BYTES + row * KSTR + (D / 8) * 16) = make_uint4(0x00003C00u, 0u, 0u, 0u);
        }
    }
    if (SUMROW) {
        // V^T rows D..DV-1 of both buffers: row D = 1.0 (fp16 0x3C00), the rest 0 — written once, never overwritten
        constexpr int PADCH = (DV - D) * VCPR;
        for (int i = tid; i < 2 * PADCH; i += 256) {
            const int buf = i / PADCH, j = i - buf * PADCH;
            const int row = D + j / VCPR, c = j % VCPR;
            const unsigned w = row == D ? 0x3C003C00u : 0u;
            *reinterpret_cast<uint4*>(smem + buf * TILE_BYTES + K_BYTES + row * VSTR + c * 16) = make_uint4(w, w, w, w);
        }
    }
    __syncthreads();

    const f2v sl2 = {p.scale_log2, p.scale_log2};
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        const bool more = !NO_STAGE && t + 1 < ntiles;
        const long long ta = stamp();
        if (more) load_tile(t + 1);
        const char* tile = smem + (NO_STAGE ? 0 : cur) * TILE_BYTES;

        // ---- S^T for the NKB 32-key blocks of this tile --------------------------------------------------------
        f16v sc[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kb][r] = 0.f;
        h8 vaf[PREF ? NDB : 1][PREF ? NKB : 1][2];
        if constexpr (PREF) {
            h8 kaf[NDC][NKB];
#pragma unroll
            for (int dc = 0; dc < NDC; ++dc)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) kaf[dc][kb] = *reinterpret_cast<const h8*>(tile + ka_off + kb * 32 * KSTR + dc * 32);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dc = 0; dc < NDC; ++dc)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kaf[dc][kb], qf[dc], sc[kb], 0, 0, 0);
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
                        vaf[db][kb][sb] = *reinterpret_cast<const h8*>(tile + va_off + db * 32 * VSTR + (kb * 32 + sb * 16) * 2);
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
        for (int dc = 0; dc < NDC; ++dc) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const h8 ka = *reinterpret_cast<const h8*>(tile + ka_off + kb * 32 * KSTR + dc * 32);
                if constexpr (NO_S) { asm volatile("" : "+v"(sc[kb]) : "v"(ka), "v"(qf[dc])); }
                else sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka, qf[dc], sc[kb], 0, 0, 0);
            }
        }
        }

        const long long tb = stamp();
        // ---- online softmax (per lane: one query row, KVT/2 of the tile's keys) -----------------------------------
        // register r of block kb <-> local key 32*kb + 16*(r>>3) + 8*half + (r&7)
        if (t >= nfull || p.causal) {
            // ragged last tile (cross-attention, M = 77) or causal mask (CLIP).  The empty volatile asm keeps this a real
            // (wave-uniform) branch: if-converted, the 32 compare+select pairs would run on every tile of every call.
            asm volatile("");
            int lim = p.M - t * KVT - 8 * half;                // local keys >= lim are past the end
            if (p.causal) lim = min(lim, q + 1 - t * KVT - 8 * half);      // ... or in this query's future
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (32 * kb + 16 * (r >> 3) + (r & 7) >= lim) sc[kb][r] = -INFINITY;
        }
        h8 pb[NKB][2];
        long long tc = 0;
        if constexpr (FOLD) {
            // the accumulators are s * c - shift already
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kb][r]);
            float mlo = mx, mhi = mx;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(mlo), "+v"(mhi));
            mx = fmaxf(mlo, mhi);
            // per query; the first tile always sets the shift (it may be negative); later tiles raise it only when a score exceeds it by
            // more than tau (AttnP::tau: P <= 2^tau until then)
            const bool up = t == 0 || mx > fmaxf(p.tau, 0.f);
            if (__builtin_amdgcn_ballot_w64(up) != 0) {
                asm volatile("");                                 // a real (wave-uniform) branch: rare after the first few tiles
                // new shift = the smallest fp16 number >= shift + mx (both halves of the wave compute the same value for a query)
                const float want = fmaxf(shift + mx, -60000.f);
                half_t hs = (half_t)want;
                float f = (float)hs;
                if (f < want) {
                    unsigned short bits = __builtin_bit_cast(unsigned short, hs);
                    bits = (f >= 0.f) ? (unsigned short)(bits + 1) : (unsigned short)(bits - 1);
                    hs = __builtin_bit_cast(half_t, bits);
                    f = (float)hs;
                }
                const float delta = up ? f - shift : 0.f;         // exact: both are fp16 numbers
                if (up) shift = f;
                const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[kb][r] -= delta;
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
                if (half == F_HALF) qf[F_DC][F_E] = (half_t)(-shift);
            }
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    pb[kb][r >> 3][r & 7] = (half_t)__builtin_amdgcn_exp2f(sc[kb][r]);
                    pb[kb][r >> 3][(r & 7) + 1] = (half_t)__builtin_amdgcn_exp2f(sc[kb][r + 1]);
                }
        } else {
        float mx = -INFINITY;
        if constexpr (NO_MAX) { mx = sc[0][0] * p.scale_log2; }
        else {
        if constexpr (TREEMAX) mx = tree_max<NKB>(sc);
        else {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kb][r]);
        }
        if constexpr (PREF) {
            // the other half-wave's maximum without the LDS round trip of ds_bpermute: v_permlane32_swap_b32 on two copies leaves
            // (lower-half value, upper-half value) of the same query in every lane
            // (inline asm: with both operands the same value the builtin's two results were folded into one by the compiler)
            float mlo = mx, mhi = mx;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(mlo), "+v"(mhi));
            mx = fmaxf(mlo, mhi) * p.scale_log2;
        } else
        mx = fmaxf(mx, __shfl_xor(mx, 32)) * p.scale_log2;      // scale > 0: max commutes with the scaling
        }
        // Exponent base m_run: with tau = 0 the running maximum (P <= 1).  The lazy forms re-base — all 64 lanes together, O^T rescaled —
        // only when some query's tile maximum exceeds its base by more than tau; until then the base stands and P <= 2^tau (fp16 P, fp32
        // sums: any tau <= 15 is exact up to the rounding of P, whose relative precision does not depend on its scale).  With a wave-wide
        // test and i.i.d. scores a plain running maximum moves in ~65 % of the 64 tiles of a 4096-key row; with tau = 8 only in the first.
        // (the round-1 form, VAR 0 — head sizes 64 / 80 / 128 / 160 and the 77-key launches — rescales in every tile only under knob attn_tau -1)
        const bool max_moved = (!LAZY_RESCALE && p.tau < 0.f) || __builtin_amdgcn_ballot_w64(mx > m_run + fmaxf(p.tau, 0.f)) != 0;   // wave-uniform
        const float m_new = max_moved ? fmaxf(m_run, mx) : m_run;
        if constexpr (TIMING) { asm volatile("" :: "v"(m_new)); tc = stamp(); }
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        const f2v mneg = {-m_new, -m_new};
        f2v rs = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f2v s2 = {sc[kb][r], sc[kb][r + 1]};
                const f2v y = __builtin_elementwise_fma(s2, sl2, mneg);          // v_pk_fma_f32
                f2v e;
                if constexpr (NO_EXP) e = y; else e = f2v{__builtin_amdgcn_exp2f(y.x), __builtin_amdgcn_exp2f(y.y)};
                if (!SUMROW) rs += e;                                            // v_pk_add_f32
                pb[kb][r >> 3][r & 7] = (half_t)e.x;
                pb[kb][r >> 3][(r & 7) + 1] = (half_t)e.y;
            }
        if (!SUMROW) l_run = l_run * alpha + rs;
        if (max_moved) {
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
        }   // !FOLD

        // ---- O^T += V^T P^T ---------------------------------------------------------------------------------
        if constexpr (PREF) {
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vaf[db][kb][sb], pb[kb][sb], o[db], 0, 0, 0);
        } else {
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb) {
                        const h8 va = *reinterpret_cast<const h8*>(tile + va_off + db * 32 * VSTR + (kb * 32 + sb * 16) * 2);
                        if constexpr (NO_PV) { asm volatile("" : "+v"(o[db]) : "v"(va), "v"(pb[kb][sb])); }
                        else
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pb[kb][sb], o[db], 0, 0, 0);
                    }
                }
            }
        }
        // double-buffered LDS: the other buffer was last read in iteration t-1 (all waves passed its barrier)
        long long td = 0, te = 0;
        if constexpr (TIMING) { __builtin_amdgcn_sched_barrier(0); td = stamp(); }
        if (more) write_tile(cur ^ 1);
        if constexpr (TIMING) { __builtin_amdgcn_sched_barrier(0); te = stamp(); }
        if constexpr (!NO_STAGE) __syncthreads();
        if constexpr (TIMING) {
            const long long tf = stamp();
            tm[0] += tb - ta; tm[1] += tc - tb; tm[2] += td - tc; tm[3] += te - td; tm[4] += tf - te; tm[5] += 1;
        }
    }
    if constexpr (TIMING) {
        if (p.dbg && lane == 0) {
            long long* dst = p.dbg + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
#pragma unroll
            for (int i = 0; i < 6; ++i) dst[i] = tm[i];
        }
    }

    // ---- normalise and store: o[db][r] is O[q][db*32 + (r&3) + 8*(r>>2) + 4*half] ----------------------------
    float l_tot;
    if (SUMROW) l_tot = __shfl(o[L_DB][L_R], lq + 32 * L_HALF);
    else l_tot = (l_run.x + l_run.y) + __shfl_xor(l_run.x + l_run.y, 32);
    const float inv = 1.0f / l_tot;
    if (qok) {
        half_t* optr = p.out + ((long)b * p.N + q) * p.ldo + h * D;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = db * 32 + g * 8 + half * 4;
                if (d0 < D) {
                    h4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (half_t)(o[db][g * 4 + e] * inv);
                    *reinterpret_cast<h4*>(optr + d0) = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Role-offset ("ping-pong") form of the flash kernel for long key sequences (round 3).
//
// Round 2's section timers (profiles/r02_attention_experiments.md) showed a wave-tile costing 921 SIMD cycles against 448 cycles of
// MFMA issue and ~450 of VALU issue: with three independent workgroups per CU the MFMA and VALU phases of the resident waves
// do not arrange themselves to coincide.  Here they are MADE to: 8 waves = two groups of 4 (one wave of each group per SIMD, as in
// the ping-pong GEMM) run the same two-section program ONE SECTION APART,
//     V(t): softmax of S(t) -> P(t) (VALU only), then the LDS reads of the MFMA operands of M(t)
//     M(t): O += V^T(t) P^T(t)  and  S(t+1) = K(t+1) Q^T   (14 MFMAs of d = 40, four independent accumulator chains), the LDS
//           writes of tile pair t+2 and the global loads of pair t+3 (register staged, two sections of flight time)
// with one s_barrier per section boundary for all 8 waves: whenever one wave of a SIMD is in its MFMA section the other one is in
// its VALU section.  The loop is software-pipelined by one tile (S(t+1) is issued before softmax(t+1) needs it), so the score tile
// of a wave is live across one barrier only.  LDS holds three {V^T(u), K(u+1)} pairs: a pair is written two sections or more before
// its first read and overwritten two sections or more after its last (schedule in the body).
//
// FOLD (head sizes with a spare contraction column, d = 40 -> 48): the softmax shift costs no VALU.  Q is multiplied by
// scale * log2(e) when its fragments are loaded (one fp16 rounding of q * c instead of q), K's padding column holds 1.0 and Q's padding
// element holds -shift, so the S^T MFMA itself delivers s * c - shift: the 16 v_pk_fma_f32 per tile (a fifth of the VALU section; the
// most expensive VALU form beside another wave's MFMAs, profiles/r02_valu_rates.txt) disappear and exp2 is applied to the accumulators
// directly.  The shift is an fp16 number (softmax is invariant to it as long as O, the denominator and P use the same one); it is
// raised — scores re-based, O rescaled, Q's padding element rewritten — only when a tile's maximum exceeds it (lazy, wave-uniform).
// ---------------------------------------------------------------------------------------------------------------
template <int D, bool FOLD, int NG, bool TIMING = false>
__global__ __launch_bounds__(NG * 256, NG) void attn_pp_kernel(AttnP p) {
    // TIMING (attn_occ 28 / 38, tools/gpu/attn_pp_sections.py): s_memtime stamps on both sides of every section barrier; per wave the
    // cycle sums of (V1 work, barrier wait, V2 work, wait, M work, wait) and the iteration count go to AttnP::dbg.  The stamp's own
    // lgkmcnt(0) makes the V sections wait for their operand reads: read the V2 / M split with that in mind.
    long long tm[7] = {0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&]() -> long long {
        if constexpr (!TIMING) return 0;
        const long long t = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return t;
    };
    constexpr int KVT = 64;
    constexpr int NT = NG * 256;             // NG groups of 4 waves (one wave of each group per SIMD)
    static_assert(NG == 2 || NG == 3, "two or three wave groups");
    constexpr int DK = (D + 15) / 16 * 16, NDC = DK / 16;
    constexpr int DV = (D + 31) / 32 * 32, NDB = DV / 32;
    constexpr int NKB = KVT / 32;
    constexpr int KSTR = DK * 2 + 16, VSTR = KVT * 2 + 16;       // odd numbers of 16-byte slots: conflict-free ds_read_b128
    constexpr int K_BYTES = KVT * KSTR, V_BYTES = DV * VSTR, SLOT = K_BYTES + V_BYTES;
    constexpr int NSLOT = 3;
    constexpr int KCR = D / 8;               // 16-byte chunks of a K row that exist in memory (D % 8 == 0)
    constexpr int KCPR = DK / 8;             // ... of its LDS image (padding chunks are constants written once)
    constexpr int VCPR = KVT / 8;
    constexpr int KCH = KVT * KCR, VCH = D * VCPR;
    constexpr int K_IT = (KCH + NT - 1) / NT, V_IT = (VCH + NT - 1) / NT;
    constexpr bool SUMROW = DV > D;          // see attn_mfma_kernel: row D of V^T is all ones, the PV MFMA accumulates the softmax denominator
    constexpr int L_RR = D - (D / 32) * 32, L_DB = D / 32;
    constexpr int L_HALF = (L_RR >> 2) & 1, L_R = (L_RR & 3) + 4 * (L_RR >> 3);
    static_assert(D % 8 == 0, "head size must be a multiple of 8");
    static_assert(!FOLD || DK > D, "FOLD needs a padding column in the contraction");
    constexpr int PDC = D / 16, PHALF = (D % 16) / 8;            // where column D sits in the Q^T / K fragments (element 0 of that chunk)
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NSLOT x (K tile | V^T tile), then one more K tile (K(0))

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;               // 0: leading group; group g runs g sections behind
    const int half = lane >> 5, lq = lane & 31;
    int bh, qb;
    {
        const int nqb = gridDim.x, nbh = gridDim.y;
        const int id = blockIdx.y * nqb + blockIdx.x;
        if ((nbh & 7) == 0) {                // all query blocks of a head on one XCD (see attn_mfma_kernel)
            const int xcd = id & 7, slot = id >> 3;
            qb = slot % nqb;
            bh = (slot / nqb) * 8 + xcd;
        } else {
            bh = blockIdx.y; qb = blockIdx.x;
        }
    }
    const int b = bh / p.H, h = bh - b * p.H;
    const int q = qb * (NG * 128) + wave * 32 + lq;
    const bool qok = q < p.N;

    // ---- Q^T fragments (B operand of S^T): lane holds Q[q][dc*16 + half*8 .. +8); FOLD: times scale * log2(e) ------------------
    h8 qf[NDC];
    {
        const half_t* qptr = p.q + ((long)b * p.N + (qok ? q : 0)) * p.ldq + h * D;
#pragma unroll
        for (int dc = 0; dc < NDC; ++dc) {
            const int d = dc * 16 + half * 8;
            h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (qok && d < D) v = *reinterpret_cast<const h8*>(qptr + d);
            if constexpr (FOLD) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (half_t)((float)v[j] * p.scale_log2);
            }
            qf[dc] = v;
        }
    }
    const half_t* kbase = p.k + (long)b * p.M * p.ldk + h * D;
    const half_t* vbase = p.vt + ((long)b * p.H + h) * D * (long)p.vt_ld;

    // Register-staged tiles, addresses clamped instead of predicated (attn_mfma_kernel's rules): K rows >= M re-read row M-1 (masked in
    // the ragged tile), V^T columns stay below vt_ld for every tile that exists.
    u4v kr[K_IT], vr[V_IT];
    auto load_k = [&](int key0) {
#pragma unroll
        for (int it = 0; it < K_IT; ++it) {
            const int idx = it * NT + tid;
            const int row = idx / KCR, c = idx - row * KCR;
            const int rr = min(key0 + row, p.M - 1);
            kr[it] = *reinterpret_cast<const u4v*>(kbase + (long)rr * p.ldk + c * 8);
        }
    };
    auto load_v = [&](int key0) {
#pragma unroll
        for (int it = 0; it < V_IT; ++it) {
            const int idx = it * NT + tid;
            const int row = idx / VCPR, c = idx - row * VCPR;
            const int col = min(key0 + c * 8, p.vt_ld - 8);
            vr[it] = *reinterpret_cast<const u4v*>(vbase + (long)min(row, D - 1) * p.vt_ld + col);
        }
    };
    char* const dump = smem + NSLOT * SLOT + K_BYTES + tid * 16;     // NT x 16 bytes behind the tile images
    auto write_k = [&](char* Ks) {
#pragma unroll
        for (int it = 0; it < K_IT; ++it) {
            const int idx = it * NT + tid;
            const int row = idx / KCR, c = idx - row * KCR;
            // (threads past the end of the tile store into a per-thread dump slot instead of being predicated off: an exec-mask branch
            // here splits the M section into basic blocks, and the compiler then sinks the softmax VALU work across the barrier into them)
            char* dst = (KCH % NT == 0 || idx < KCH) ? Ks + row * KSTR + c * 16 : dump;
            *reinterpret_cast<u4v*>(dst) = kr[it];
        }
    };
    auto write_v = [&](char* Vs) {
#pragma unroll
        for (int it = 0; it < V_IT; ++it) {
            const int idx = it * NT + tid;
            const int row = idx / VCPR, c = idx - row * VCPR;
            char* dst = (VCH % NT == 0 || idx < VCH) ? Vs + row * VSTR + c * 16 : dump;
            *reinterpret_cast<u4v*>(dst) = vr[it];
        }
    };
    // wait_lds: this wave's LDS writes must have landed before the others pass (end of an M section).  The V section ends with the
    // operand READS of the following M section still in flight: they return under the barrier wait (their slot is overwritten two
    // barriers later at the earliest), the compiler's own counted lgkmcnt waits precede the MFMAs that use them.
    auto section_barrier = [&](bool wait_lds = true) {
        __builtin_amdgcn_sched_barrier(0);
        if (wait_lds) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    const int T = (p.M + KVT - 1) / KVT;
    const int nfull = p.M / KVT;
    char* const K0s = smem + NSLOT * SLOT;

    // ---- prologue: constants, K(0), pairs 0 and 1 ----------------------------------------------------------------------------
    {
        // padding chunks of every K image (columns D .. DK-1): zeros — FOLD: column D = 1.0 (fp16 0x3C00), so Q's padding element is
        // added to every score; written once, the staging writes never touch them
        constexpr int KPAD = KCPR - KCR;
        if constexpr (KPAD > 0) {
            for (int i = tid; i < (NSLOT + 1) * KVT * KPAD; i += NT) {
                const int img = i / (KVT * KPAD), j = i - img * (KVT * KPAD);
                const int row = j / KPAD, c = KCR + j % KPAD;
                char* base = img < NSLOT ? smem + img * SLOT : K0s;
                const unsigned w0 = (FOLD && c == KCR) ? 0x00003C00u : 0u;
                *reinterpret_cast<uint4*>(base + row * KSTR + c * 16) = make_uint4(w0, 0u, 0u, 0u);
            }
        }
        if constexpr (SUMROW) {              // V^T rows D .. DV-1 of every slot: row D = 1.0, the rest 0
            constexpr int PADCH = (DV - D) * VCPR;
            for (int i = tid; i < NSLOT * PADCH; i += NT) {
                const int img = i / PADCH, j = i - img * PADCH;
                const int row = D + j / VCPR, c = j % VCPR;
                const unsigned w = row == D ? 0x3C003C00u : 0u;
                *reinterpret_cast<uint4*>(smem + img * SLOT + K_BYTES + row * VSTR + c * 16) = make_uint4(w, w, w, w);
            }
        }
        load_k(0);
        write_k(K0s);
        load_v(0); load_k(KVT);
        write_v(smem + K_BYTES); write_k(smem);
        if (T > 1) {
            load_v(KVT); load_k(2 * KVT);
            write_v(smem + SLOT + K_BYTES); write_k(smem + SLOT);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // K row read by this lane as MFMA row (lane & 31): bits 2 and 3 swapped, so that the 8 scores a lane packs for one PV MFMA are 8
    // consecutive keys (attn_mfma_kernel's header)
    const int krow = (lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1);
    const int ka_off = krow * KSTR + half * 16;
    const int va_off = K_BYTES + lq * VSTR + half * 16;

    f16v o[NDB], sc[NKB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[kb][r] = 0.f;
    // S(0) from the extra K image
    {
        h8 kaf0[NDC][NKB];
#pragma unroll
        for (int dc = 0; dc < NDC; ++dc)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) kaf0[dc][kb] = *reinterpret_cast<const h8*>(K0s + ka_off + kb * 32 * KSTR + dc * 32);
#pragma unroll
        for (int dc = 0; dc < NDC; ++dc)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kaf0[dc][kb], qf[dc], sc[kb], 0, 0, 0);
    }
    load_v(2 * KVT); load_k(3 * KVT);                         // pair 2: written in M(0) (clamped addresses when it does not exist)

    // FOLD: `shift` is the fp16 number currently subtracted by the MFMA (Q's padding element holds -shift); else the running maximum
    // in the scaled (log2) domain
    float shift = FOLD ? 0.f : -1e30f;
    f2v l_run = {0.f, 0.f};
    const f2v sl2 = {p.scale_log2, p.scale_log2};
    h8 pb[NKB][2];
    h8 kaf[NDC][NKB], vaf[NDB][NKB][2];

    section_barrier();
    for (int i = 0; i < grp; ++i) section_barrier();          // group g runs g sections behind from here on
    // barriers of the last iteration that have no partner: group g skips its final g (of NB per iteration)
    constexpr int NB = NG;

    for (int t = 0; t < T; ++t) {
        // ================================ V section: softmax of S(t), operand reads for M(t) ================================
        const long long ta = stamp();
        long long tb = ta, tc = ta;
        if (t >= nfull || p.causal) {
            asm volatile("");                                 // a real (wave-uniform) branch, see attn_mfma_kernel
            int lim = p.M - t * KVT - 8 * half;
            if (p.causal) lim = min(lim, q + 1 - t * KVT - 8 * half);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (32 * kb + 16 * (r >> 3) + (r & 7) >= lim) sc[kb][r] = -INFINITY;
        }
        float mx = tree_max<NKB>(sc);
        {
            float mlo = mx, mhi = mx;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(mlo), "+v"(mhi));
            mx = fmaxf(mlo, mhi);
        }
        if constexpr (FOLD) {
            // mx is relative to the current shift.  Raise the shift when some query's maximum exceeds it (first tile: set it).
            const bool raise = t == 0 || __builtin_amdgcn_ballot_w64(mx > 0.f) != 0;
            if (raise) {
                asm volatile("");
                const float want = shift + mx;
                const float ns = (t == 0 || mx > 0.f) ? (float)(half_t)fminf(fmaxf(want, -60000.f), 60000.f) : shift;
                const float delta = ns - shift;               // exact: both are fp16 numbers
                shift = ns;
                if (t > 0) {
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int db = 0; db < NDB; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
                    if (!SUMROW) l_run = l_run * alpha;
                }
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[kb][r] -= delta;
                if (half == PHALF) qf[PDC][0] = (half_t)(-ns);
            }
        } else {
            mx *= p.scale_log2;                               // scale > 0: max commutes with the scaling
            const float m_new = fmaxf(shift, mx);
            const float alpha = __builtin_amdgcn_exp2f(shift - m_new);
            const bool moved = __builtin_amdgcn_ballot_w64(m_new > shift) != 0;
            shift = m_new;
            if (!SUMROW) l_run = l_run * alpha;
            if (moved) {
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
            }
        }
        // exponentials of one 32-key block -> P (fp16, packed for the PV MFMA).  FOLD: the accumulators already are s * c - shift
        const f2v mneg = {-shift, -shift};
        auto exp_block = [&](int kb) {
            f2v rs = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f2v e;
                if constexpr (FOLD) {
                    e = f2v{__builtin_amdgcn_exp2f(sc[kb][r]), __builtin_amdgcn_exp2f(sc[kb][r + 1])};
                } else {
                    const f2v s2 = {sc[kb][r], sc[kb][r + 1]};
                    const f2v y = __builtin_elementwise_fma(s2, sl2, mneg);
                    e = f2v{__builtin_amdgcn_exp2f(y.x), __builtin_amdgcn_exp2f(y.y)};
                }
                if (!SUMROW) rs += e;
                pb[kb][r >> 3][r & 7] = (half_t)e.x;
                pb[kb][r >> 3][(r & 7) + 1] = (half_t)e.y;
            }
            if (!SUMROW) l_run += rs;
        };
        // barrier k (0 .. NB-1) of this iteration; in the last iteration group g skips its final g barriers (they have no partner)
        auto bar = [&](int k, bool wait_lds) {
            if (!(t + 1 == T && k >= NB - grp)) section_barrier(wait_lds);
        };
        exp_block(0);
        if constexpr (NG == 3) {
            // three groups: the VALU work of a tile is split over two sections (max + first key block | second key block + operand
            // reads), so that TWO waves of a SIMD are in VALU sections while the third is in its MFMA section — one wave alone issues a
            // VALU instruction only every 6-9 cycles (profiles/r02_valu_rates.txt), which made the two-group form VALU-latency bound
            asm volatile("" :: "v"(pb[0][0]), "v"(pb[0][1]));
            tb = stamp();
            bar(0, false);
            tc = stamp();
        }
#pragma unroll
        for (int kb = 1; kb < NKB; ++kb) exp_block(kb);
        // operands of M(t) from pair t (slot t % 3: complete since the end of section M(t-2) of the second group)
        {
            const char* tile = smem + (t % NSLOT) * SLOT;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
                        vaf[db][kb][sb] = *reinterpret_cast<const h8*>(tile + va_off + db * 32 * VSTR + (kb * 32 + sb * 16) * 2);
#pragma unroll
            for (int dc = 0; dc < NDC; ++dc)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) kaf[dc][kb] = *reinterpret_cast<const h8*>(tile + ka_off + kb * 32 * KSTR + dc * 32);
        }
        // P(t) must exist BEFORE the barrier: without a use here the compiler sinks the exponentials into the M section
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) asm volatile("" :: "v"(pb[kb][0]), "v"(pb[kb][1]));
        const long long td = stamp();
        bar(NB - 2, false);
        const long long te = stamp();
        // ================================ M section: O += V^T(t) P^T(t), S(t+1) = K(t+1) Q^T ================================
        // (S(T) of the last iteration is computed from clamped rows and never used: no branch in the MFMA stream)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kb][r] = 0.f;
        // staging, in the shadow of the MFMAs (no branch: the scheduler is free to interleave): pair t+2 (loaded one M section ago)
        // into slot (t+2) % 3 — last read in V(t-1) of the second group, two barriers back —, then the loads of pair t+3.  Pairs past
        // the end are loaded from clamped addresses and written like the others: nothing reads them.
        {
            char* slot = smem + ((t + 2) % NSLOT) * SLOT;
            write_v(slot + K_BYTES); write_k(slot);
            load_v((t + 3) * KVT); load_k((t + 4) * KVT);
        }
        // four independent accumulator chains, round robin
#pragma unroll
        for (int step = 0; step < (NDC > 2 * NKB ? NDC : 2 * NKB); ++step) {
            if (step < 2 * NKB) {
                const int kb = step >> 1, sb = step & 1;
#pragma unroll
                for (int db = 0; db < NDB; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vaf[db][kb][sb], pb[kb][sb], o[db], 0, 0, 0);
            }
            if (step < NDC) {
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kaf[step][kb], qf[step], sc[kb], 0, 0, 0);
            }
        }
        const long long tf = stamp();
        bar(NB - 1, true);
        if constexpr (TIMING) {
            const long long tg = stamp();
            tm[0] += tb - ta; tm[1] += tc - tb; tm[2] += td - tc; tm[3] += te - td; tm[4] += tf - te; tm[5] += tg - tf; tm[6] += 1;
        }
    }
    if constexpr (TIMING) {
        if (p.dbg && lane == 0) {
            long long* dst = p.dbg + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * (4 * NG) + wave) * 8;
#pragma unroll
            for (int i = 0; i < 7; ++i) dst[i] = tm[i];
        }
    }

    // ---- normalise and store: o[db][r] is O[q][db*32 + (r&3) + 8*(r>>2) + 4*half] ----------------------------
    float l_tot;
    if (SUMROW) l_tot = __shfl(o[L_DB][L_R], lq + 32 * L_HALF);
    else l_tot = (l_run.x + l_run.y) + __shfl_xor(l_run.x + l_run.y, 32);
    const float inv = 1.0f / l_tot;
    if (qok) {
        half_t* optr = p.out + ((long)b * p.N + q) * p.ldo + h * D;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = db * 32 + g * 8 + half * 4;
                if (d0 < D) {
                    h4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (half_t)(o[db][g * 4 + e] * inv);
                    *reinterpret_cast<h4*>(optr + d0) = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Generic kernel (any D <= 512, M <= 16384): one wave per query row, scores kept in LDS.  Slow; used for head sizes the
// MFMA kernel is not instantiated for and as the independent HIP cross-check in the parity tests.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void attn_generic_kernel(AttnP p) {
    extern __shared__ float sc[];                 // M scores
    const int lane = threadIdx.x;
    const int q = blockIdx.x, bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const half_t* qptr = p.q + ((long)b * p.N + q) * p.ldq + h * p.D;
    const half_t* kbase = p.k + (long)b * p.M * p.ldk + h * p.D;
    const half_t* vbase = p.vt + ((long)b * p.H + h) * p.D * (long)p.vt_ld;
    float mx = -INFINITY;
    const int m_vis = p.causal ? min(p.M, q + 1) : p.M;       // keys visible to this query
    for (int key = lane; key < p.M; key += 64) {
        if (key >= m_vis) { sc[key] = -INFINITY; continue; }
        const half_t* kp = kbase + (long)key * p.ldk;
        float acc = 0.f;
        for (int d = 0; d < p.D; ++d) acc = fmaf((float)qptr[d], (float)kp[d], acc);
        acc *= p.scale_log2;
        sc[key] = acc;
        mx = fmaxf(mx, acc);
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.f;
    for (int key = lane; key < p.M; key += 64) {
        const float e = exp2f(sc[key] - mx);
        sc[key] = e;
        sum += e;
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    __syncthreads();
    const float inv = 1.0f / sum;
    for (int d = lane; d < p.D; d += 64) {
        const half_t* vp = vbase + (long)d * p.vt_ld;
        float acc = 0.f;
        for (int key = 0; key < p.M; ++key) acc = fmaf(sc[key], (float)vp[key], acc);
        p.out[((long)b * p.N + q) * p.ldo + h * p.D + d] = (half_t)(acc * inv);
    }
}

// v [B, M, ldv] (head h at columns h*D) -> vt [B, H*D, Mpad], zero padded.  32x32 LDS tile transpose.
__global__ __launch_bounds__(256) void transpose_v_kernel(const half_t* v, half_t* vt, int M, int C, int ldv, int Mpad) {
    __shared__ half_t tile[32][33];
    const int b = blockIdx.z;
    const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int m = m0 + i, c = c0 + tx;
        tile[i][tx] = (m < M && c < C) ? v[((long)b * M + m) * ldv + c] : (half_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, m = m0 + tx;
        if (c < C && m < Mpad) vt[((long)b * C + c) * Mpad + m] = tile[tx][i];
    }
}

int launch_transpose_v(const half_t* v, half_t* vt, int B, int H, int M, int D, int ldv, int Mpad, hipStream_t s) {
    const int C = H * D;
    dim3 grid(cdiv(Mpad, 32), cdiv(C, 32), B);
    hipLaunchKernelGGL(transpose_v_kernel, grid, dim3(256), 0, s, v, vt, M, C, ldv, Mpad);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

unsigned long long g_attn_dbg = 0;
int g_attn_kvt = [] { const char* e = getenv("SDMI_ATTN_KVT"); return e ? atoi(e) : 0; }();

// default 15 since round 2: same-box A/Bs on the C1 job — 0 -> 5: self-attention 72.4 -> 69.5 ms per job (profiles/r02_knob_sweep.md);
// 5 -> 15: 68.2 -> 66.6 and 70.0 -> 68.7 ms on two boxes (profiles/r02_attention_experiments.md)
int g_attn_occ = [] { const char* e = getenv("SDMI_ATTN_OCC"); return e ? atoi(e) : 15; }();
// slack of the lazy re-basing in log2 units (AttnP::tau); 0 = the round-2 behaviour (bit-identical to the non-lazy forms).  Default 8 since
// round 6: level-0 self-attention 545.4 -> 525.0 us isolated (-3.7 %; 4 and 12 the same), C1 forward 16.893 -> 16.812 ms in a same-box A/B,
// 3.8e-4 from the tau = 0 output = the distance of two fp16 realisations of P (profiles/r06_attn_ab_tau.txt, r06_fwd_ab_tau.txt)
int g_attn_tau = [] { const char* e = getenv("SDMI_ATTN_TAU"); return e ? atoi(e) : 8; }();
// shortest d = 40 self-attention the folded-shift form (17) takes; 0: never.  8192 (the hires pass only) until round 6; with the re-basing
// slack the form's bookkeeping block runs once per row instead of in most tiles and it leads at N = 4096 too: 525.3 -> 487.6 us per
// level-0 launch (profiles/r06_attn_ab_forms_with_slack.txt), attention error 2.9e-4 -> 3.6e-4 (cap 5e-4), UNet forward parity unchanged
int g_attn_fold_min_m = [] { const char* e = getenv("SDMI_ATTN_FOLD_MIN_M"); return e ? atoi(e) : 1024; }();
int g_attn_lds_pad = 0;      // tuning only: extra dynamic LDS per workgroup (bytes) = an occupancy limiter (32 KB + pad per workgroup of 160 KB)
int g_attn_pp_min_m = [] { const char* e = getenv("SDMI_ATTN_PP_MIN_M"); return e ? atoi(e) : 256; }();   // shortest key sequence the 8-wave kernel takes

template <int D, int KVT, int VAR = 0>
static int launch_attn_d(const AttnP& p, hipStream_t s) {
    constexpr int DK = (D + 15) / 16 * 16, DV = (D + 31) / 32 * 32;
    constexpr int SMEM = 2 * (KVT * (DK * 2 + 16) + DV * (KVT * 2 + 16));
    auto kern = attn_mfma_kernel<D, KVT, VAR>;
    static PerDeviceOnce attr;
    const int smem = SMEM + (g_attn_lds_pad > 0 ? g_attn_lds_pad : 0);
    if (attr.need(smem)) SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    dim3 grid(cdiv(p.N, 128), p.B * p.H);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int D, bool FOLD, int NG = 2, bool TIMING = false>
static int launch_attn_pp(const AttnP& p, hipStream_t s) {
    constexpr int DK = (D + 15) / 16 * 16, DV = (D + 31) / 32 * 32;
    constexpr int SMEM = 3 * (64 * (DK * 2 + 16) + DV * (64 * 2 + 16)) + 64 * (DK * 2 + 16) + NG * 256 * 16;   // 3 pairs, K(0), dump slots
    auto kern = attn_pp_kernel<D, FOLD, NG, TIMING>;
    static PerDeviceOnce attr;
    if (attr.need()) SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    dim3 grid(cdiv(p.N, NG * 128), p.B * p.H);
    hipLaunchKernelGGL(kern, grid, dim3(NG * 256), SMEM, s, p);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_attention(const AttnP& p_in, bool force_generic, hipStream_t s) {
    AttnP p = p_in;
    p.tau = (float)(g_attn_tau < 0 ? -1 : g_attn_tau > 12 ? 12 : g_attn_tau);
    SDMI_REQUIRE(p.B > 0 && p.H > 0 && p.N > 0 && p.M > 0 && p.D > 0, "empty attention");
    SDMI_REQUIRE(p.vt_ld >= (p.M + 63) / 64 * 64, "vt_ld must be >= M rounded up to 64");
    const double pf_flops = 4.0 * p.B * p.H * (double)p.N * p.M * p.D;
    const double pf_bytes = 2.0 * p.B * p.H * ((double)p.N * p.D * 2 + (double)p.M * p.D * 2);
    ProfScope ps(force_generic ? "attention_generic" : (p.M > 128 ? "attention_mfma_self" : "attention_mfma_cross"), pf_flops, pf_bytes, s);
    const bool aligned = (p.ldq % 8 == 0) && (p.ldk % 8 == 0) && (p.vt_ld % 8 == 0) && (p.ldo % 4 == 0) && (p.D % 8 == 0);
    const bool kvt128 = g_attn_kvt == 128;       // measured: 64-key tiles win for every head size (profiles/)
    if (!force_generic && aligned) {
        // knob attn_kvt = 96: the text context (77 keys; M in (64, 96]) as ONE 96-key tile — three 32-key blocks — instead of a 64-key tile
        // plus a 13-key one padded to 64: 21 instead of 28 MFMAs and 96 instead of 128 score columns per query block, no second staging /
        // barrier round
        if (g_attn_kvt == 96 && p.M > 64 && p.M <= 96 && !p.causal) {
            switch (p.D) {
                case 40: return launch_attn_d<40, 96, 0>(p, s);          // (the fragment-prefetch form 15 spills with three key blocks)
                case 64: return launch_attn_d<64, 96>(p, s);
                case 80: return launch_attn_d<80, 96>(p, s);
                case 160: return launch_attn_d<160, 96>(p, s);
                default: break;
            }
        }
        switch (p.D) {
            // KV tile: 128 keys where the register budget allows it (small heads: the per-tile barrier / staging
            // overhead is amortised over twice the MFMA work), 64 otherwise or when the key sequence is short
            case 40:
                // 20 / 21: the role-offset 8-wave kernel (21: softmax shift folded into the S^T MFMA) for the long self-attention launches
                if ((g_attn_occ == 20 || g_attn_occ == 21) && p.M >= g_attn_pp_min_m && p.N >= 256)
                    return g_attn_occ == 21 ? launch_attn_pp<40, true>(p, s) : launch_attn_pp<40, false>(p, s);
                if ((g_attn_occ == 28 || g_attn_occ == 38) && p.M >= g_attn_pp_min_m && p.N >= 384) {      // section timers
                    AttnP q = p; q.dbg = (long long*)g_attn_dbg;
                    return g_attn_occ == 38 ? launch_attn_pp<40, true, 3, true>(q, s) : launch_attn_pp<40, true, 2, true>(q, s);
                }
                // 30 / 31: the same with THREE groups (12 waves, 384 queries per workgroup; VALU work split over two sections)
                if ((g_attn_occ == 30 || g_attn_occ == 31) && p.M >= g_attn_pp_min_m && p.N >= 384)
                    return g_attn_occ == 31 ? launch_attn_pp<40, true, 3>(p, s) : launch_attn_pp<40, false, 3>(p, s);
                if (!(kvt128 && p.M > 64)) {
                    if (g_attn_occ == 5) return launch_attn_d<40, 64, 5>(p, s);
                    // long self-attention (the 128x128-latent hires pass: N = M = 16384) takes the folded-shift form 17 by default: -4.2 % on
                    // the launch, -1.3 % on the c4a job in a same-box A/B, parity at those shapes re-measured with it
                    // (profiles/r03_parity_fullsize.json); at N = 4096 (C1) the two forms are within 1 % and 15 keeps the measured numerics
                    if (g_attn_occ == 15 && g_attn_fold_min_m > 0 && p.M >= g_attn_fold_min_m && p.N >= g_attn_fold_min_m && !p.causal)
                        return launch_attn_d<40, 64, 17>(p, s);
                    if (g_attn_occ == 15) return launch_attn_d<40, 64, 15>(p, s);
                    if (g_attn_occ == 16) return launch_attn_d<40, 64, 16>(p, s);
                    if (g_attn_occ == 17) return launch_attn_d<40, 64, 17>(p, s);
#ifdef SDMI_ATTN_PARTS
                    if (g_attn_occ == 10) return launch_attn_d<40, 64, 10>(p, s);
                    if (g_attn_occ == 11) return launch_attn_d<40, 64, 11>(p, s);
                    if (g_attn_occ == 12) return launch_attn_d<40, 64, 12>(p, s);
                    if (g_attn_occ == 13) return launch_attn_d<40, 64, 13>(p, s);
                    if (g_attn_occ == 14) return launch_attn_d<40, 64, 14>(p, s);
                    if (g_attn_occ == 18) { AttnP q = p; q.dbg = (long long*)g_attn_dbg; return launch_attn_d<40, 64, 18>(q, s); }
#endif
                }
                return (kvt128 && p.M > 64) ? launch_attn_d<40, 128>(p, s) : launch_attn_d<40, 64>(p, s);
            case 64:
                if ((g_attn_occ == 20 || g_attn_occ == 21) && p.M >= g_attn_pp_min_m && p.N >= 256) return launch_attn_pp<64, false>(p, s);
                return (kvt128 && p.M > 64) ? launch_attn_d<64, 128>(p, s) : launch_attn_d<64, 64>(p, s);
            case 80: return launch_attn_d<80, 64>(p, s);
            case 128: return launch_attn_d<128, 64>(p, s);
            case 160: return launch_attn_d<160, 64>(p, s);
            default: break;
        }
    }
    SDMI_REQUIRE(p.D <= 512 && p.M <= 16384, "generic attention supports D <= 512 and M <= 16384");
    static PerDeviceOnce attr;
    if (attr.need()) SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)attn_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4));
    hipLaunchKernelGGL(attn_generic_kernel, dim3(p.N, p.B * p.H), dim3(64), p.M * sizeof(float), s, p);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace sdmi
