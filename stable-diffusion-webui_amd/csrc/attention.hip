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

namespace sdmi {

template <int D>
__global__ __launch_bounds__(256) void attn_mfma_kernel(AttnP p) {
    constexpr int DK = (D + 15) / 16 * 16;   // contraction length of S^T, padded to the MFMA K step
    constexpr int NDC = DK / 16;
    constexpr int DV = (D + 31) / 32 * 32;   // rows of O^T, padded to the MFMA M
    constexpr int NDB = DV / 32;
    constexpr int KSTR = DK * 2 + 16;        // bytes; (DK/8 + 1) slots of 16 B -> odd
    constexpr int VSTR = 64 * 2 + 16;        // 9 slots
    constexpr int K_BYTES = 64 * KSTR, V_BYTES = DV * VSTR;
    constexpr int KCPR = DK / 8;             // 16-byte chunks per K row
    constexpr int KCH = 64 * KCPR, VCH = DV * 8;
    constexpr int K_IT = (KCH + 255) / 256, V_IT = (VCH + 255) / 256;
    __shared__ __attribute__((aligned(16))) char smem[K_BYTES + V_BYTES];
    char* Ks = smem;
    char* Vs = smem + K_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, lq = lane & 31;
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int q = blockIdx.x * 128 + wave * 32 + lq;
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
            qf[dc] = v;
        }
    }

    const half_t* kbase = p.k + (long)b * p.M * p.ldk + h * D;
    const half_t* vbase = p.vt + ((long)b * p.H + h) * D * (long)p.vt_ld;

    uint4 kr[K_IT], vr[V_IT];
    auto load_tile = [&](int t) {
        const int key0 = t * 64;
#pragma unroll
        for (int it = 0; it < K_IT; ++it) {
            const int idx = it * 256 + tid;
            const int row = idx / KCPR, c = idx - row * KCPR;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < KCH && key0 + row < p.M && c * 8 < D)
                v = *reinterpret_cast<const uint4*>(kbase + (long)(key0 + row) * p.ldk + c * 8);
            kr[it] = v;
        }
#pragma unroll
        for (int it = 0; it < V_IT; ++it) {
            const int idx = it * 256 + tid;
            const int row = idx >> 3, c = idx & 7;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < VCH && row < D) v = *reinterpret_cast<const uint4*>(vbase + (long)row * p.vt_ld + key0 + c * 8);
            vr[it] = v;
        }
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int it = 0; it < K_IT; ++it) {
            const int idx = it * 256 + tid;
            const int row = idx / KCPR, c = idx - row * KCPR;
            if (idx < KCH) *reinterpret_cast<uint4*>(Ks + row * KSTR + c * 16) = kr[it];
        }
#pragma unroll
        for (int it = 0; it < V_IT; ++it) {
            const int idx = it * 256 + tid;
            const int row = idx >> 3, c = idx & 7;
            if (idx < VCH) *reinterpret_cast<uint4*>(Vs + row * VSTR + c * 16) = vr[it];
        }
    };

    f16v o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    // K row read by this lane as MFMA row (lane&31): bits 2 and 3 swapped (see header)
    const int krow = (lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1);
    const char* ka_ptr = Ks + krow * KSTR + half * 16;
    const char* va_ptr = Vs + lq * VSTR + half * 16;

    const int ntiles = (p.M + 63) / 64;
    load_tile(0);
    write_tile();
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const bool more = t + 1 < ntiles;
        if (more) load_tile(t + 1);

        // ---- S^T for the two 32-key blocks of this tile ------------------------------------------------------
        f16v s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
        for (int dc = 0; dc < NDC; ++dc) {
            const h8 ka0 = *reinterpret_cast<const h8*>(ka_ptr + dc * 32);
            const h8 ka1 = *reinterpret_cast<const h8*>(ka_ptr + 32 * KSTR + dc * 32);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka0, qf[dc], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka1, qf[dc], s1, 0, 0, 0);
        }

        // ---- online softmax (per lane: one query row, 32 of the tile's 64 keys) -------------------------------
        // register r of block kb <-> local key 32*kb + 16*(r>>3) + 8*half + (r&7)
        const int key0 = t * 64;
        const bool tail = key0 + 64 > p.M;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a = s0[r] * p.scale_log2, c = s1[r] * p.scale_log2;
            if (tail) {
                const int kl = 16 * (r >> 3) + 8 * half + (r & 7);
                if (key0 + kl >= p.M) a = -INFINITY;
                if (key0 + 32 + kl >= p.M) c = -INFINITY;
            }
            s0[r] = a; s1[r] = c;
            mx = fmaxf(mx, fmaxf(a, c));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float rs = 0.f;
        h8 pb[2][2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e0 = __builtin_amdgcn_exp2f(s0[r] - m_new);
            const float e1 = __builtin_amdgcn_exp2f(s1[r] - m_new);
            rs += e0 + e1;
            pb[0][r >> 3][r & 7] = (half_t)e0;
            pb[1][r >> 3][r & 7] = (half_t)e1;
        }
        l_run = l_run * alpha + rs;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;

        // ---- O^T += V^T P^T ---------------------------------------------------------------------------------
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    const h8 va = *reinterpret_cast<const h8*>(va_ptr + db * 32 * VSTR + (kb * 32 + sb * 16) * 2);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pb[kb][sb], o[db], 0, 0, 0);
                }
            }
        }
        __syncthreads();                 // every wave is done reading this tile
        if (more) write_tile();
        __syncthreads();
    }

    // ---- normalise and store: o[db][r] is O[q][db*32 + (r&3) + 8*(r>>2) + 4*half] ----------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32);
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
    for (int key = lane; key < p.M; key += 64) {
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

template <int D>
static int launch_attn_d(const AttnP& p, hipStream_t s) {
    dim3 grid(cdiv(p.N, 128), p.B * p.H);
    hipLaunchKernelGGL(attn_mfma_kernel<D>, grid, dim3(256), 0, s, p);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_attention(const AttnP& p, bool force_generic, hipStream_t s) {
    SDMI_REQUIRE(p.B > 0 && p.H > 0 && p.N > 0 && p.M > 0 && p.D > 0, "empty attention");
    SDMI_REQUIRE(p.vt_ld >= (p.M + 63) / 64 * 64, "vt_ld must be >= M rounded up to 64");
    const double pf_flops = 4.0 * p.B * p.H * (double)p.N * p.M * p.D;
    const double pf_bytes = 2.0 * p.B * p.H * ((double)p.N * p.D * 2 + (double)p.M * p.D * 2);
    ProfScope ps(force_generic ? "attention_generic" : (p.M > 128 ? "attention_mfma_self" : "attention_mfma_cross"), pf_flops, pf_bytes, s);
    const bool aligned = (p.ldq % 8 == 0) && (p.ldk % 8 == 0) && (p.vt_ld % 8 == 0) && (p.ldo % 4 == 0) && (p.D % 8 == 0);
    if (!force_generic && aligned) {
        switch (p.D) {
            case 40: return launch_attn_d<40>(p, s);
            case 64: return launch_attn_d<64>(p, s);
            case 80: return launch_attn_d<80>(p, s);
            case 128: return launch_attn_d<128>(p, s);
            case 160: return launch_attn_d<160>(p, s);
            default: break;
        }
    }
    SDMI_REQUIRE(p.D <= 512 && p.M <= 16384, "generic attention supports D <= 512 and M <= 16384");
    static bool attr_set = false;
    if (!attr_set) {
        SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)attn_generic_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4));
        attr_set = true;
    }
    hipLaunchKernelGGL(attn_generic_kernel, dim3(p.N, p.B * p.H), dim3(64), p.M * sizeof(float), s, p);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace sdmi
