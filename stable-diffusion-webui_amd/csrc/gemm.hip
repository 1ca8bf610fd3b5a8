// gemm.hip — implicit-GEMM convolution / linear layer for gfx950 (MI355X), fp16 in, fp32 accumulate on MFMA.
//
//   out[m, n] = epilogue( alpha * sum_k A(m, k) * W[n, k] ),   k = tap*Cin + c
//
// A(m, k) is gathered on the fly from one or two NHWC fp16 activation tensors (3x3 taps with zero padding,
// stride 1/2, optional fused nearest-x2 upsample, optional channel concatenation of two sources), so no im2col
// buffer, no torch.cat and no F.interpolate ever touch HBM.  W is packed [N][K] (K contiguous), which makes both MFMA
// operands K-contiguous ("B^T" form).
//
// Design for CDNA4:
//   * 256 threads = 4 wave64; block tile BM x BN x 64(K); each wave owns a (BM/WR) x (BN/WC) sub-tile built from
//     16x16x32 f16 MFMAs (v_mfma_f32_16x16x32_f16), accumulators in registers.
//   * operands staged global -> LDS with the LDS-direct load (global_load_lds_dwordx4, 16 B/lane, no VGPR round
//     trip); the LDS image is lane-linear, so the bank-conflict-free XOR swizzle (16-B chunk c of row r lives in slot
//     c ^ (r & 7)) is applied on the per-lane SOURCE address and undone on the ds_read_b128 side.  Zero padding comes
//     from pointing padded lanes at a zero page.  (A register-staged variant with the identical LDS image is kept
//     as template GLDS=false.)
//   * double-buffered LDS, one barrier per K step; loads of step k+1 are in flight while step k runs on the MFMAs.
//   * MFMA is issued as D = W_frag x A_frag (i.e. the transposed product) so each lane ends up with 4 CONSECUTIVE
//     output channels of one pixel: the epilogue does 8-byte packed stores and vector bias / residual loads.
//   * XCD-aware block remap: consecutive tile ids (which share the gathered A panel) are kept on one XCD's L2.
//
// Replaces the torch op sequences enumerated in SURVEY.md 2.3 rows K1, K4, K7, K8 (ldm ResBlock / Downsample /
// Upsample / SpatialTransformer projections / GEGLU feed-forward; names pinned at
// /root/reference/extensions-builtin/Lora/networks.py:43-98).
#include "common.h"
#include "prof.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>

namespace sdmi {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct RowInfo {
    int b, yo, xo;
    bool ok;
};

// Row bookkeeping of the gather, precomputed once per thread: top-left source coordinate of the 3x3 window (in the
// x2-upsampled grid when p.up) and the image base pixel.
struct GRow {
    int yb, xb;        // yi_raw = yb + dy, xi_raw = xb + dx
    int pixbase;       // b * Hi * Wi (pixel counts stay far below 2^31)
    bool ok;
};

// exact (erf) GELU as ldm's GEGLU uses (F.gelu default).  erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below
// the fp16 rounding of the result): the GEGLU epilogue evaluates it 64x per lane per tile, libm erff made it dominate
// the K = 320 feed-forward GEMMs.
__device__ __forceinline__ float gelu_erf(float g) {
    const float x = fabsf(g) * 0.70710678118654752f;
    // v_rcp_f32 (1 ulp) — __frcp_rn expands to the ten-instruction IEEE division sequence, and the GEGLU epilogue is VALU-bound
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(x * x * -1.4426950408889634f);     // exp(-x^2)
    const float erf_abs = fmaf(-(poly * t), e, 1.0f);
    const float erf_v = copysignf(erf_abs, g);
    const float hg = 0.5f * g;
    return fmaf(hg, erf_v, hg);
}

// 16-byte epilogue accesses out of the 8-byte accumulator layout (v_permlane16_swap_b32, gfx950).  The MFMA leaves a lane with
// 4 consecutive channels (lane>>4)*4.. of row (lane & 15) of a 16x16 tile: 8-byte stores, and the store (and residual load) ISSUE
// rate — not HBM — paces the epilogue of the short-K layers.  Take the packed fp16 quads x, y of the SAME column tile in the row
// tiles 2a and 2a+1 and exchange the odd 16-lane rows of x with the even rows of y (one instruction per dword):
//     x' = [x.row0, y.row0, x.row2, y.row2]     y' = [x.row1, y.row1, x.row3, y.row3]
// Lane L now holds, for row tile 2a + ((L>>4)&1) and row (L & 15), channels (L>>5)*8 + 0..3 in x' and + 4..7 in y': one 16-byte
// access.  The exchange is an involution, so the same call turns a residual loaded in the wide layout back into the accumulator
// layout.  Pure data movement: results are bit-identical to the 8-byte path (EP_NARROW, tests/test_gpu_ops.py).
typedef unsigned u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void swap16(h4& x, h4& y) {
    const u2v xa = __builtin_bit_cast(u2v, x), ya = __builtin_bit_cast(u2v, y);
    const u2v r0 = __builtin_amdgcn_permlane16_swap(xa[0], ya[0], false, false);
    const u2v r1 = __builtin_amdgcn_permlane16_swap(xa[1], ya[1], false, false);
    x = __builtin_bit_cast(h4, (u2v){r0[0], r1[0]});
    y = __builtin_bit_cast(h4, (u2v){r0[1], r1[1]});
}
__device__ __forceinline__ h8 join8(h4 lo, h4 hi) { return h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; }

// (hi, lo) stream tensors (EP_HILO, engine option "residual_fp32"): the epilogue of a launch whose output and / or residual is a pair of
// fp16 tensors, x = hi + lo (GemmP).  A SEPARATE region, entered from gemm_epilogue behind one block-uniform branch, so that the default
// path's code and register allocation stay what they were (round 6: the same handling written into the default path's loops cost the
// 256x320 instantiations 40-155 VGPR spills).  Row-tile pairs in the 16-byte layout (swap16), column tile outer so that the
// GroupNorm-statistics form keeps 8 live sums, the (hi, lo) residual of pair a + 1 requested before pair a is converted and stored.
// v = alpha * acc + bias_scale * bias [+ rowbias] [+ resid_hi + resid_lo];  hi = fp16(v), lo = fp16(v - hi);  STATS: per (image, row
// chunk, group) sums of hi + lo — what the consumer's GroupNorm reads — in gemm_epilogue's layout and (fixed) order.
template <int TM, int TN, int WTM, int WTN, int WR_, int BN_, bool STATS>
__device__ __forceinline__ void gemm_epilogue_hilo(const GemmP& p, f4 (&acc)[TM][TN], int m0, int n0, int wr, int wc, int lane, long z,
                                                   char* smem) {
    static_assert(TM % 2 == 0, "row-tile pairs");
    const half_t* rlo = gemm_resid_lo(p);
    half_t* olo = gemm_out_lo(p);
    const long ob = z * p.o_bs, rbs = z * p.r_bs;
    const int sel = (lane >> 4) & 1, nw = (lane >> 5) * 8, lr = lane & 15;
    const int mw = m0 + wr * WTM + sel * 16 + lr;            // the lane's store row in pair 0 (pair a: + 32 a)
    const int mrow = m0 + wr * WTM + lr;                     // the lane's accumulator row in tile 0 (tile i: + 16 i)
    auto ld = [&](const half_t* base, int a, int j) {
        return *reinterpret_cast<const h8*>(base + rbs + (long)min(mw + a * 32, p.M - 1) * p.ldr + n0 + wc * WTN + j * 16 + nw);
    };
    // one (pair a, column tile j) cell: v = alpha acc + bias [+ rowbias] [+ resid hi + lo]; returns the pair (hi, lo) in the 16-byte layout
    // and, for the statistics form, leaves the accumulator-layout values in vx / vy
    auto cell = [&](int a, int j, const f4& bb, const h8& ch, const h8& cl, f4& vx, f4& vy, h4& ox, h4& oy, h4& lx, h4& ly) {
        const int n = n0 + wc * WTN + j * 16 + (lane >> 4) * 4;
        vx = acc[2 * a][j]; vy = acc[2 * a + 1][j];
#pragma unroll
        for (int r = 0; r < 4; ++r) { vx[r] = fmaf(vx[r], p.alpha, bb[r]); vy[r] = fmaf(vy[r], p.alpha, bb[r]); }
        if (p.rowbias) {
            const int mx = mrow + 2 * a * 16;
            vx += *reinterpret_cast<const f4*>(p.rowbias + (long)(min(mx, p.M - 1) / p.rows_per_batch) * p.ldrb + n);
            vy += *reinterpret_cast<const f4*>(p.rowbias + (long)(min(mx + 16, p.M - 1) / p.rows_per_batch) * p.ldrb + n);
        }
        if (p.resid) {
            h4 rx = {ch[0], ch[1], ch[2], ch[3]}, ry = {ch[4], ch[5], ch[6], ch[7]};
            swap16(rx, ry);
#pragma unroll
            for (int r = 0; r < 4; ++r) { vx[r] += (float)rx[r]; vy[r] += (float)ry[r]; }
            if (rlo) {
                h4 sx = {cl[0], cl[1], cl[2], cl[3]}, sy = {cl[4], cl[5], cl[6], cl[7]};
                swap16(sx, sy);
#pragma unroll
                for (int r = 0; r < 4; ++r) { vx[r] += (float)sx[r]; vy[r] += (float)sy[r]; }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ox[r] = (half_t)vx[r]; oy[r] = (half_t)vy[r];
            lx[r] = (half_t)(vx[r] - (float)ox[r]); ly[r] = (half_t)(vy[r] - (float)oy[r]);
        }
    };
    auto store = [&](int a, int j, h4 ox, h4 oy, h4 lx, h4 ly) {
        swap16(ox, oy);
        swap16(lx, ly);
        const int ms = mw + a * 32;
        if (ms < p.M) {
            const long o = ob + (long)ms * p.ldo + n0 + wc * WTN + j * 16 + nw;
            *reinterpret_cast<h8*>((half_t*)p.out + o) = join8(ox, oy);
            if (olo) *reinterpret_cast<h8*>(olo + o) = join8(lx, ly);
        }
    };
    auto col_bias = [&](int j) {
        f4 bb = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
            bb = *reinterpret_cast<const f4*>(p.bias + n0 + wc * WTN + j * 16 + (lane >> 4) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) bb[r] *= p.bias_scale;
        }
        return bb;
    };
    const h8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (!STATS) {
        // plain form: row-tile pair OUTER, column tile inner, as the default path — consecutive column tiles of a row are written back to
        // back, so the 32-byte pieces of a line meet in the L2 (column-tile-outer measured 80.9 us against 35.4 us for the fp16 launch on
        // M65536 N320 K320: 2.6 TB/s).  The (hi, lo) residual of cell (a + 1, j) is requested as soon as cell (a, j) has consumed its
        // registers: a one-deep ring with a whole row pair of cover.
        constexpr bool RING = TN <= 5;
        constexpr bool HOIST = RING && TM * TN < 40;         // (256x320: 160 accumulator registers — the column bias is fetched per cell)
        f4 bcw[HOIST ? TN : 1];
        h8 rh[RING ? TN : 1], rl[RING ? TN : 1];
        if constexpr (RING) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (HOIST) bcw[j] = col_bias(j);
                rh[j] = p.resid ? ld(p.resid, 0, j) : z8;
                rl[j] = rlo ? ld(rlo, 0, j) : z8;
            }
        }
#pragma unroll
        for (int a = 0; a < TM / 2; ++a) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f4 vx, vy;
                h4 ox, oy, lx, ly;
                if constexpr (RING) {
                    const h8 ch = rh[j], cl = rl[j];
                    if (a + 1 < TM / 2) {
                        if (p.resid) rh[j] = ld(p.resid, a + 1, j);
                        if (rlo) rl[j] = ld(rlo, a + 1, j);
                    }
                    cell(a, j, HOIST ? bcw[HOIST ? j : 0] : col_bias(j), ch, cl, vx, vy, ox, oy, lx, ly);
                } else {
                    const h8 ch = p.resid ? ld(p.resid, a, j) : z8, cl = rlo ? ld(rlo, a, j) : z8;
                    cell(a, j, col_bias(j), ch, cl, vx, vy, ox, oy, lx, ly);
                }
                store(a, j, ox, oy, lx, ly);
            }
        }
        return;
    }
    // statistics form: column tile OUTER so that the lane's column sums over its TM rows are 8 live values (gemm_epilogue's STATS note)
    [[maybe_unused]] float* cs = reinterpret_cast<float*>(smem);
    [[maybe_unused]] float* cq = cs + WR_ * BN_;
    __syncthreads();                                         // every wave is done reading the operand tiles that lived here
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const f4 bb = col_bias(j);
        f4 sv = {0.f, 0.f, 0.f, 0.f}, qv = {0.f, 0.f, 0.f, 0.f};
        h8 rh = p.resid ? ld(p.resid, 0, j) : z8, rl = rlo ? ld(rlo, 0, j) : z8;
#pragma unroll
        for (int a = 0; a < TM / 2; ++a) {
            const h8 ch = rh, cl = rl;
            if (a + 1 < TM / 2) {
                if (p.resid) rh = ld(p.resid, a + 1, j);
                if (rlo) rl = ld(rlo, a + 1, j);
            }
            f4 vx, vy;
            h4 ox, oy, lx, ly;
            cell(a, j, bb, ch, cl, vx, vy, ox, oy, lx, ly);
            const bool okx = mrow + 2 * a * 16 < p.M, oky = mrow + (2 * a + 1) * 16 < p.M;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float f = okx ? (float)ox[r] + (float)lx[r] : 0.f;
                sv[r] += f;
                qv[r] = fmaf(f, f, qv[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float f = oky ? (float)oy[r] + (float)ly[r] : 0.f;
                sv[r] += f;
                qv[r] = fmaf(f, f, qv[r]);
            }
            store(a, j, ox, oy, lx, ly);
        }
#pragma unroll
        for (int off = 1; off < 16; off <<= 1)
#pragma unroll
            for (int r = 0; r < 4; ++r) { sv[r] += __shfl_xor(sv[r], off); qv[r] += __shfl_xor(qv[r], off); }
        if ((lane & 15) == 0) {
            const int c = wc * WTN + j * 16 + (lane >> 4) * 4;
            *reinterpret_cast<f4*>(cs + wr * BN_ + c) = sv;
            *reinterpret_cast<f4*>(cq + wr * BN_ + c) = qv;
        }
    }
    __syncthreads();
    const int tid = threadIdx.x, cpg = p.stats_cpg, ngl = BN_ / cpg;
    if (tid < ngl) {
        float a = 0.f, q = 0.f;
        for (int w = 0; w < WR_; ++w)
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a += cs[w * BN_ + c]; q += cq[w * BN_ + c]; }
        const int b = m0 / p.rows_per_batch;                 // (a tile lies inside one image: launch_gemm's admission rule)
        const int chunk = (m0 - b * p.rows_per_batch) / (TM * 16 * WR_);
        const int G = p.N / cpg, g = n0 / cpg + tid;
        float* dst = p.stats_out + (((long)b * p.stats_nchunk + chunk) * G + g) * 2;
        dst[0] = a; dst[1] = q;
    }
}

// Epilogue shared by the GEMM kernels.  Lane holds, for accumulator tile (i, j):
//   m = m0 + wr*WTM + i*16 + (lane & 15),  n = n0 + wc*WTN + j*16 + (lane>>4)*4 + r   (r = 0..3: 4 consecutive channels)
typedef float f2e __attribute__((ext_vector_type(2)));
// LNF (EP_LNFOLD launches): finish a LayerNorm whose affine part is folded into the weights — v = rstd[m] * (v - mean[m] * s[n]) on the
// alpha-scaled accumulator, before the (folded) bias is added (GemmP::ln_stats).
template <int TM, int TN, int WTM, int WTN, bool GEGLU, bool TR = false, int WR_ = 1, int BN_ = 64, bool STATS = false, int LNM = 0>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f4 (&acc)[TM][TN], int m0, int n0, int wr, int wc, int lane, long z,
                                              char* smem = nullptr) {
    constexpr bool LNF = LNM == 1;       // LayerNorm consumer (EP_LNFOLD)
    constexpr bool LNS = LNM == 2;       // LayerNorm producer: per-row partial sums of the output (GemmP::lnp_out)
    static_assert(!(STATS && LNM != 0), "the GroupNorm-statistics epilogue neither follows nor feeds a LayerNorm");
    static_assert(!(LNS && (GEGLU || TR)), "row partial sums are taken by the plain epilogue only");
    [[maybe_unused]] const f2e* lnst = reinterpret_cast<const f2e*>(p.ln_stats);
    // (mean, rstd) of input row m: stored as such, or finished here from the producer's per-tile partial sums (fixed order)
    [[maybe_unused]] auto row_stat = [&](int m) -> f2e {
        if (p.ln_np == 0) return lnst[m];
        const float* pp = p.ln_stats + (long)m * p.ln_np * 2;
        float sm = 0.f, q = 0.f;
        for (int t = 0; t < p.ln_np; ++t) { sm += pp[2 * t]; q += pp[2 * t + 1]; }
        const float mean = sm * p.ln_inv_c;
        return f2e{mean, rsqrtf(fmaxf(fmaf(-mean, mean, q * p.ln_inv_c), 0.f) + p.ln_eps)};
    };
    if constexpr (TR) {
        // transposed store (EP_TRANSPOSE): the MFMAs ran with swapped operand roles, so for tile (i, j) the lane holds
        //   n = n0 + wc*WTN + j*16 + (lane & 15),  m = m0 + wr*WTM + i*16 + (lane>>4)*4 + r   (r = 0..3: 4 consecutive tokens)
        // out[b][n][m - b*rows_per_batch], 8-byte packed stores along the token dimension (rows_per_batch % 4 == 0).
        const long ob = z * p.o_bs;
        if constexpr (TM % 2 == 0) {
            if (!(p.flags & EP_NARROW)) {
                // wide form (swap16): 8 consecutive tokens of one channel per lane, 16-byte stores
                const int sel = (lane >> 4) & 1, tw = (lane >> 5) * 8;
#pragma unroll
                for (int a = 0; a < TM / 2; ++a) {
                    const int m = m0 + wr * WTM + (2 * a + sel) * 16 + tw;
                    const int b = min(m, p.M - 1) / p.rows_per_batch;
                    const int ml = m - b * p.rows_per_batch;
                    f2e stx[LNF ? 4 : 1], sty[LNF ? 4 : 1];      // (mean, rstd) of the 4 + 4 tokens this lane holds before the exchange
                    if constexpr (LNF) {
                        const int tx = m0 + wr * WTM + 2 * a * 16 + (lane >> 4) * 4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { stx[r] = row_stat(min(tx + r, p.M - 1)); sty[r] = row_stat(min(tx + 16 + r, p.M - 1)); }
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int n = n0 + wc * WTN + j * 16 + (lane & 15);
                        const float bb = p.bias ? p.bias[n] * p.bias_scale : 0.f;
                        h4 ox, oy;
                        if constexpr (LNF) {
                            const float sn = p.ln_s[n];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                ox[r] = (half_t)(stx[r][1] * (acc[2 * a][j][r] * p.alpha - stx[r][0] * sn) + bb);
                                oy[r] = (half_t)(sty[r][1] * (acc[2 * a + 1][j][r] * p.alpha - sty[r][0] * sn) + bb);
                            }
                        } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            ox[r] = (half_t)fmaf(acc[2 * a][j][r], p.alpha, bb);
                            oy[r] = (half_t)fmaf(acc[2 * a + 1][j][r], p.alpha, bb);
                        }
                        }
                        swap16(ox, oy);
                        if (m < p.M) *reinterpret_cast<h8*>((half_t*)p.out + ob + ((long)b * p.N + n) * p.ldo + ml) = join8(ox, oy);
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wr * WTM + i * 16 + (lane >> 4) * 4;
            if (m >= p.M) continue;
            const int b = m / p.rows_per_batch;
            const int ml = m - b * p.rows_per_batch;
            f2e stn[LNF ? 4 : 1];
            if constexpr (LNF) {
#pragma unroll
                for (int r = 0; r < 4; ++r) stn[r] = row_stat(min(m + r, p.M - 1));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wc * WTN + j * 16 + (lane & 15);
                const float bb = p.bias ? p.bias[n] * p.bias_scale : 0.f;
                h4 o;
                if constexpr (LNF) {
                    const float sn = p.ln_s[n];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (half_t)(stn[r][1] * (acc[i][j][r] * p.alpha - stn[r][0] * sn) + bb);
                } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (half_t)fmaf(acc[i][j][r], p.alpha, bb);
                }
                *reinterpret_cast<h4*>((half_t*)p.out + ob + ((long)b * p.N + n) * p.ldo + ml) = o;
            }
        }
        return;
    }
    // ---- epilogue ----------------------------------------------------------------------------------------
    // lane holds, for tile (i, j):  m = m0 + wr*WTM + i*16 + (lane & 15),  n = n0 + wc*WTN + j*16 + (lane>>4)*4 + r
    const int flags = p.flags;
    // LNF: (mean, rstd) of the lane's TM rows, fetched ONCE, all loads in flight together (round 3: fetched per column tile inside the
    // j loops — TN dependent 8-byte loads per row pair — the folded layers ran 2x slower than LayerNorm + plain GEMM)
    [[maybe_unused]] f2e lst[LNF ? TM : 1];
    if constexpr (LNF) {
#pragma unroll
        for (int i = 0; i < TM; ++i) lst[i] = row_stat(min(m0 + wr * WTM + i * 16 + (lane & 15), p.M - 1));
    }
    const long ob = z * p.o_bs, rbs = z * p.r_bs;
    if (p.splitk > 1) {
        float* slab = p.splitk_ws + ((long)blockIdx.y * gridDim.z + z) * (long)p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wr * WTM + i * 16 + (lane & 15);
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wc * WTN + j * 16 + (lane >> 4) * 4;
                *reinterpret_cast<f4*>(slab + (long)m * p.N + n) = acc[i][j];
            }
        }
        // the slabs are summed by splitk_reduce_kernel, a second launch.  (Round 3 built the alternative — each tile's last-arriving slice
        // sums the slabs itself behind an agent-scope release / acquire pair and a ticket counter, same bits — and measured it on the C1 job:
        // 458.1 ms against 424.8 ms, profiles/r03_knob_sweep_run5.json; every split launch paid the vmcnt(0) drain + two workgroup barriers
        // + the L2 write-back of the fence, more than the 13 us reduce launches it saved.  Removed again: it also cost the production
        // 256x320 instantiation 6 SGPR spills.)
        return;
    }
    // (not the statistics form of the 256x320 tiles: 160 accumulator registers per lane leave no room for it — launch_gemm keeps the
    // consumer's own statistics pass for those launches)
    if constexpr (!GEGLU && LNM == 0 && TM % 2 == 0 && !(STATS && TM * TN >= 40)) {
        if ((flags & EP_HILO) && !(flags & EP_NARROW)) {     // (hi, lo) stream tensors: their own region (block-uniform branch)
            gemm_epilogue_hilo<TM, TN, WTM, WTN, WR_, BN_, STATS>(p, acc, m0, n0, wr, wc, lane, z, smem);
            return;
        }
    }
    if constexpr (STATS && !GEGLU) {
        // ---- GroupNorm-statistics variant (GemmP::stats_out; launch_gemm admits only fp16 row-major outputs with a column bias,
        // an optional [B][N] row bias and an optional residual).  Column tile OUTER, row tile inner: the lane's column sums over
        // its TM rows are then 8 live values (one column tile at a time) instead of 8 x TN next to the accumulators, and the
        // residual pipeline is a 4-deep ring of 8-byte loads.  After each column tile: 16-lane shuffle tree over the rows of the
        // wave (lanes sharing lane >> 4 own the same 4 channels), LDS [WR][BN] per wave row; at the end one thread per group adds
        // its WR x cpg slots in a fixed order and writes (sum, sum of squares) for (image, row chunk, group).
        // The sums are taken from the fp16-ROUNDED outputs: exactly what gn_stats would have read back.
        float* cs = reinterpret_cast<float*>(smem);          // [WR_][BN_] sums, then [WR_][BN_] sums of squares
        float* cq = cs + WR_ * BN_;
        __syncthreads();                                     // every wave is done reading the operand tiles that lived here
        const int mrow = m0 + wr * WTM + (lane & 15);
        const int b = m0 / p.rows_per_batch;                 // (a tile lies inside one image: launch_gemm's admission rule)
        constexpr int RD = TM < 4 ? TM : 4;
        const bool wide = TM % 2 == 0 && !(p.flags & EP_NARROW);
        const int sel = (lane >> 4) & 1, nw = (lane >> 5) * 8;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wc * WTN + j * 16 + (lane >> 4) * 4;
            f4 bb = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
                bb = *reinterpret_cast<const f4*>(p.bias + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) bb[r] *= p.bias_scale;
            }
            if (p.rowbias) bb += *reinterpret_cast<const f4*>(p.rowbias + (long)b * p.ldrb + n);
            f4 sv = {0.f, 0.f, 0.f, 0.f}, qv = {0.f, 0.f, 0.f, 0.f};
            if constexpr (TM % 2 == 0) {
                if (wide) {
                    // row-tile pairs, residual and output in the 16-byte layout (swap16); the sums are taken in the accumulator
                    // layout in the same order as below (tile 2a, then 2a+1)
                    constexpr int RP2 = TM / 2 < 2 ? TM / 2 : 2;
                    const int nn = n0 + wc * WTN + j * 16 + nw;
                    const int mw = m0 + wr * WTM + sel * 16 + (lane & 15);
                    h8 ringw[RP2];
                    if (p.resid) {
#pragma unroll
                        for (int a = 0; a < RP2; ++a)
                            ringw[a] = *reinterpret_cast<const h8*>(p.resid + rbs + (long)min(mw + a * 32, p.M - 1) * p.ldr + nn);
                    }
#pragma unroll
                    for (int a = 0; a < TM / 2; ++a) {
                        f4 vx = acc[2 * a][j], vy = acc[2 * a + 1][j];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { vx[r] = fmaf(vx[r], p.alpha, bb[r]); vy[r] = fmaf(vy[r], p.alpha, bb[r]); }
                        if (p.resid) {
                            const h8 rr = ringw[a % RP2];
                            if (a + RP2 < TM / 2)
                                ringw[a % RP2] = *reinterpret_cast<const h8*>(p.resid + rbs + (long)min(mw + (a + RP2) * 32, p.M - 1) * p.ldr + nn);
                            h4 rx = {rr[0], rr[1], rr[2], rr[3]}, ry = {rr[4], rr[5], rr[6], rr[7]};
                            swap16(rx, ry);
#pragma unroll
                            for (int r = 0; r < 4; ++r) { vx[r] += (float)rx[r]; vy[r] += (float)ry[r]; }
                        }
                        h4 ox, oy;
                        const bool okx = mrow + 2 * a * 16 < p.M, oky = mrow + (2 * a + 1) * 16 < p.M;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            ox[r] = (half_t)vx[r];
                            const float f = okx ? (float)ox[r] : 0.f;
                            sv[r] += f;
                            qv[r] = fmaf(f, f, qv[r]);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            oy[r] = (half_t)vy[r];
                            const float f = oky ? (float)oy[r] : 0.f;
                            sv[r] += f;
                            qv[r] = fmaf(f, f, qv[r]);
                        }
                        swap16(ox, oy);
                        const int ms = mw + a * 32;
                        if (ms < p.M) *reinterpret_cast<h8*>((half_t*)p.out + ob + (long)ms * p.ldo + nn) = join8(ox, oy);
                    }
                }
            }
            if (!wide) {
            h4 ring[RD];
            if (p.resid) {
#pragma unroll
                for (int i = 0; i < RD; ++i)
                    ring[i] = *reinterpret_cast<const h4*>(p.resid + rbs + (long)min(mrow + i * 16, p.M - 1) * p.ldr + n);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mrow + i * 16;
                f4 v = acc[i][j];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(v[r], p.alpha, bb[r]);
                if (p.resid) {
                    const h4 rr = ring[i % RD];
                    if (i + RD < TM)
                        ring[i % RD] = *reinterpret_cast<const h4*>(p.resid + rbs + (long)min(m + RD * 16, p.M - 1) * p.ldr + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
                }
                if (m < p.M) {
                    h4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o[r] = (half_t)v[r];
                        const float f = (float)o[r];
                        sv[r] += f;
                        qv[r] = fmaf(f, f, qv[r]);
                    }
                    *reinterpret_cast<h4*>((half_t*)p.out + ob + (long)m * p.ldo + n) = o;
                }
            }
            }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1)
#pragma unroll
                for (int r = 0; r < 4; ++r) { sv[r] += __shfl_xor(sv[r], off); qv[r] += __shfl_xor(qv[r], off); }
            if ((lane & 15) == 0) {
                const int c = wc * WTN + j * 16 + (lane >> 4) * 4;
                *reinterpret_cast<f4*>(cs + wr * BN_ + c) = sv;
                *reinterpret_cast<f4*>(cq + wr * BN_ + c) = qv;
            }
        }
        __syncthreads();
        const int tid = threadIdx.x, cpg = p.stats_cpg, ngl = BN_ / cpg;
        if (tid < ngl) {
            float a = 0.f, q = 0.f;
            for (int w = 0; w < WR_; ++w)
                for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a += cs[w * BN_ + c]; q += cq[w * BN_ + c]; }
            const int chunk = (m0 - b * p.rows_per_batch) / (TM * 16 * WR_);
            const int G = p.N / cpg, g = n0 / cpg + tid;
            float* dst = p.stats_out + (((long)b * p.stats_nchunk + chunk) * G + g) * 2;
            dst[0] = a; dst[1] = q;
        }
        return;
    }
    // Direct stores from the accumulator layout: each lane owns 4 consecutive output channels of one pixel (8-byte
    // packed stores).  An LDS-staged, fully row-coalesced variant was measured 5-15 % SLOWER on the memory-bound 1x1
    // layers (extra barriers + LDS round trip; L2 write-combining already merges the 8-byte pieces) — profiles/.
    // Column bias is the same for every row tile: loaded once.  The residual of row tile i+1 is
    // requested before row tile i is converted and stored, so its latency hides behind the stores instead of serialising
    // TM x TN dependent load -> add -> store chains (measured: the K = 320 projections with a residual took 49 us against 31 us
    // for the same shape without one).  Loads of rows past M are clamped to row M-1 and dropped.
    // (wide wave tiles, TN > 5 — the two-stage 256x320 fallback — keep the plain per-use loads: no registers to spare)
    if constexpr (TM % 2 == 0) {
        if (!(flags & (EP_NARROW | EP_NCHW | EP_OUT_F32))) {
            // ---- 16-byte form (swap16 above): row tiles in pairs, residual loads and output stores of 8 channels per lane.  The
            // arithmetic per element is the 8-byte path's, instruction for instruction.
            const int sel = (lane >> 4) & 1, nw = (lane >> 5) * 8, lr = lane & 15;
            const int mw = m0 + wr * WTM + sel * 16 + lr;            // the lane's store row in pair 0 (pair a: + 32 a)
            if constexpr (GEGLU) {
                if constexpr (WTN % 64 == 0) {
#pragma unroll
                    for (int a = 0; a < TM / 2; ++a) {
                        const int ms = mw + a * 32;
#pragma unroll
                        for (int jg = 0; jg < TN / 4; ++jg) {
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const int npk = n0 + wc * WTN + jg * 64 + j * 16 + (lane >> 4) * 4;    // packed column of the value
                                const int nout = (n0 + wc * WTN + jg * 64) / 2 + j * 16 + nw;           // output column (wide layout)
                                f4 ba = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
                                if (p.bias) {
                                    ba = *reinterpret_cast<const f4*>(p.bias + npk);
                                    bg = *reinterpret_cast<const f4*>(p.bias + npk + 32);
                                }
                                h4 o[2];
                                [[maybe_unused]] f4 sva, svg;
                                if constexpr (LNF) {
                                    sva = *reinterpret_cast<const f4*>(p.ln_s + npk);
                                    svg = *reinterpret_cast<const f4*>(p.ln_s + npk + 32);
                                }
#pragma unroll
                                for (int t = 0; t < 2; ++t) {
                                    const f4 va = acc[2 * a + t][jg * 4 + j], vg = acc[2 * a + t][jg * 4 + j + 2];
                                    [[maybe_unused]] f2e st;
                                    if constexpr (LNF) st = lst[2 * a + t];
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        float av, g;
                                        if constexpr (LNF) {
                                            av = st[1] * (va[r] * p.alpha - st[0] * sva[r]) + ba[r];
                                            g = st[1] * (vg[r] * p.alpha - st[0] * svg[r]) + bg[r];
                                        } else {
                                            av = fmaf(va[r], p.alpha, ba[r]); g = fmaf(vg[r], p.alpha, bg[r]);
                                        }
                                        o[t][r] = (half_t)(SDMI_GELU_SIG ? geglu_gate(av, g) : av * gelu_erf(g));
                                    }
                                }
                                swap16(o[0], o[1]);
                                if (ms < p.M) *reinterpret_cast<h8*>((half_t*)p.out + ob + (long)ms * p.ldo + nout) = join8(o[0], o[1]);
                            }
                        }
                    }
                }
                return;
            } else {
                constexpr bool PIPEW = TN <= 5 && !LNF;             // (LayerNorm-folded layers never carry a residual: no ring registers)
                const bool colb = p.bias && !(flags & EP_BIAS_ROW);
                // (the row-sum variant needs the 20 registers of the hoisted bias.  The LayerNorm consumer keeps its per-column vectors as
                // per-use loads: hoisting all of them — 56 registers with the row statistics — spills the 256x320 instantiation, GPU run 3
                // of round 3; and per-use loads behind stores wait on vmcnt for the STORES too, which is why the folded projections run
                // ~2x slower than LayerNorm + plain GEMM.  The option stays experimental: what it could save is ~5 ms per C1 job.)
                constexpr bool HOISTB = PIPEW && !LNS;
                f4 bcw[HOISTB ? TN : 1];
                if constexpr (HOISTB) {
                    if (colb) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) bcw[j] = *reinterpret_cast<const f4*>(p.bias + n0 + wc * WTN + j * 16 + (lane >> 4) * 4);
                    }
                }
                // (row-sum variant on the 5-column-tile wave tiles: a one-deep ring — the pair a+1 residual is requested once pair a's
                // accumulators are dead, or the 256x320 instantiation spills; the other variants prefetch one pair ahead)
                constexpr int RWD = (LNS && TN > 4) ? 1 : 2;
                h8 rw[RWD][PIPEW ? TN : 1];
                auto prefetch_w = [&](int a, int slot) {
                    if (!PIPEW || !p.resid) return;
                    const int m = min(mw + a * 32, p.M - 1);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        rw[slot][j] = *reinterpret_cast<const h8*>(p.resid + rbs + (long)m * p.ldr + n0 + wc * WTN + j * 16 + nw);
                };
                if constexpr (PIPEW) prefetch_w(0, 0);
                // LNS: per-row (sum, sum of squares) of the fp16-rounded outputs over this tile's columns, for the LayerNorm that
                // reads this tensor: lane -> 16-lane-stride shuffles (the 4 lanes of a row) -> LDS [wave column][row] -> one thread
                // per row adds the wave columns in a fixed order.  Deterministic.
                constexpr int BMt = WR_ * WTM, WCn = BN_ / WTN;
                [[maybe_unused]] float* red = reinterpret_cast<float*>(smem);
                if constexpr (LNS) __syncthreads();                   // every wave is done reading the operand tiles that lived here
#pragma unroll
                for (int a = 0; a < TM / 2; ++a) {
                    [[maybe_unused]] float lsx = 0.f, lqx = 0.f, lsy = 0.f, lqy = 0.f;
                    if constexpr (PIPEW) {
                        if (RWD == 2 && a + 1 < TM / 2) prefetch_w(a + 1, (a + 1) & 1);
                    }
                    const int ms = mw + a * 32;
                    const int mx = min(m0 + wr * WTM + 2 * a * 16 + lr, p.M - 1), my = min(m0 + wr * WTM + (2 * a + 1) * 16 + lr, p.M - 1);
                    const int bx = mx / p.rows_per_batch, by = my / p.rows_per_batch;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int n = n0 + wc * WTN + j * 16 + (lane >> 4) * 4;
                        f4 vx = acc[2 * a][j], vy = acc[2 * a + 1][j];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { vx[r] *= p.alpha; vy[r] *= p.alpha; }
                        if constexpr (LNF) {
                            const f2e sx = lst[2 * a], sy = lst[2 * a + 1];
                            const f4 sv = *reinterpret_cast<const f4*>(p.ln_s + n);
#pragma unroll
                            for (int r = 0; r < 4; ++r) { vx[r] = sx[1] * (vx[r] - sx[0] * sv[r]); vy[r] = sy[1] * (vy[r] - sy[0] * sv[r]); }
                        }
                        if (p.bias) {
                            if (flags & EP_BIAS_ROW) {
                                const float b0 = p.bias[mx], b1 = p.bias[my];
#pragma unroll
                                for (int r = 0; r < 4; ++r) { vx[r] += b0; vy[r] += b1; }
                            } else {
                                f4 bb;
                                if constexpr (HOISTB) bb = bcw[j];
                                else bb = *reinterpret_cast<const f4*>(p.bias + n);
#pragma unroll
                                for (int r = 0; r < 4; ++r) { vx[r] = fmaf(bb[r], p.bias_scale, vx[r]); vy[r] = fmaf(bb[r], p.bias_scale, vy[r]); }
                            }
                        }
                        if (p.rowbias) {
                            vx += *reinterpret_cast<const f4*>(p.rowbias + (long)bx * p.ldrb + n);
                            vy += *reinterpret_cast<const f4*>(p.rowbias + (long)by * p.ldrb + n);
                        }
                        if (flags & (EP_QUICK_GELU | EP_GELU)) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                vx[r] = (flags & EP_QUICK_GELU) ? vx[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * vx[r])) : gelu_erf(vx[r]);
                                vy[r] = (flags & EP_QUICK_GELU) ? vy[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * vy[r])) : gelu_erf(vy[r]);
                            }
                        }
                        if (!LNF && p.resid) {
                            h8 rr;
                            if constexpr (PIPEW) rr = rw[a & (RWD - 1)][j];
                            else rr = *reinterpret_cast<const h8*>(p.resid + rbs + (long)min(ms, p.M - 1) * p.ldr + n0 + wc * WTN + j * 16 + nw);
                            h4 rx = {rr[0], rr[1], rr[2], rr[3]}, ry = {rr[4], rr[5], rr[6], rr[7]};
                            swap16(rx, ry);
#pragma unroll
                            for (int r = 0; r < 4; ++r) { vx[r] += (float)rx[r]; vy[r] += (float)ry[r]; }
                        }
                        h4 ox, oy;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { ox[r] = (half_t)vx[r]; oy[r] = (half_t)vy[r]; }
                        if constexpr (LNS) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float fx = (float)ox[r], fy = (float)oy[r];
                                lsx += fx; lqx = fmaf(fx, fx, lqx); lsy += fy; lqy = fmaf(fy, fy, lqy);
                            }
                        }
                        swap16(ox, oy);
                        if (ms < p.M) *reinterpret_cast<h8*>((half_t*)p.out + ob + (long)ms * p.ldo + n0 + wc * WTN + j * 16 + nw) = join8(ox, oy);
                    }
                    if constexpr (PIPEW && RWD == 1) {
                        if (a + 1 < TM / 2) prefetch_w(a + 1, 0);
                    }
                    if constexpr (LNS) {
                        lsx += __shfl_xor(lsx, 16); lqx += __shfl_xor(lqx, 16); lsy += __shfl_xor(lsy, 16); lqy += __shfl_xor(lqy, 16);
                        lsx += __shfl_xor(lsx, 32); lqx += __shfl_xor(lqx, 32); lsy += __shfl_xor(lsy, 32); lqy += __shfl_xor(lqy, 32);
                        if ((lane >> 4) == 0) {
                            const int rx = wr * WTM + 2 * a * 16 + lr;
                            float* d = red + ((long)wc * BMt + rx) * 2;
                            d[0] = lsx; d[1] = lqx; d[32] = lsy; d[33] = lqy;          // row rx + 16
                        }
                    }
                }
                if constexpr (LNS) {
                    __syncthreads();
                    const int tid = threadIdx.x;
                    if (tid < BMt && m0 + tid < p.M) {
                        float sm = 0.f, q = 0.f;
#pragma unroll
                        for (int w = 0; w < WCn; ++w) { sm += red[((long)w * BMt + tid) * 2]; q += red[((long)w * BMt + tid) * 2 + 1]; }
                        float* dst = p.lnp_out + ((long)(m0 + tid) * p.lnp_np + n0 / BN_) * 2;
                        dst[0] = sm; dst[1] = q;
                    }
                }
                return;
            }
        }
    }
    constexpr bool PIPE = !GEGLU && TN <= 5 && !LNF;
    const bool col_bias = p.bias && !(flags & EP_BIAS_ROW);
    f4 bcol[PIPE ? TN : 1];
    if constexpr (PIPE) {
        if (col_bias) {
#pragma unroll
            for (int j = 0; j < TN; ++j) bcol[j] = *reinterpret_cast<const f4*>(p.bias + n0 + wc * WTN + j * 16 + (lane >> 4) * 4);
        }
    }
    h4 rres[2][PIPE ? TN : 1];
    auto prefetch = [&](int i, int slot) {
        if (!PIPE || !p.resid) return;
        const int m = min(m0 + wr * WTM + i * 16 + (lane & 15), p.M - 1);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            rres[slot][j] = *reinterpret_cast<const h4*>(p.resid + rbs + (long)m * p.ldr + n0 + wc * WTN + j * 16 + (lane >> 4) * 4);
    };
    if constexpr (PIPE) prefetch(0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wr * WTM + i * 16 + (lane & 15);
        if constexpr (PIPE) {
            if (i + 1 < TM) prefetch(i + 1, (i + 1) & 1);
        }
        if (m >= p.M) continue;
        const int b = m / p.rows_per_batch;
        if constexpr (GEGLU) {
            // every 64 packed columns = 32 values followed by their 32 gates: column tiles {4g, 4g+1} / {4g+2, 4g+3}
            if constexpr (WTN % 64 == 0) {
#pragma unroll
                for (int jg = 0; jg < TN / 4; ++jg) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int nloc = j * 16 + (lane >> 4) * 4;
                        const int npk = n0 + wc * WTN + jg * 64 + nloc;           // packed column of the value
                        const int nout = (n0 + wc * WTN + jg * 64) / 2 + nloc;    // output column
                        f4 va = acc[i][jg * 4 + j], vg = acc[i][jg * 4 + j + 2];
                        f4 ba = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
                        if (p.bias) {
                            ba = *reinterpret_cast<const f4*>(p.bias + npk);
                            bg = *reinterpret_cast<const f4*>(p.bias + npk + 32);
                        }
                        h4 o;
                        if constexpr (LNF) {
                            const f2e st = lst[i];
                            const f4 sva = *reinterpret_cast<const f4*>(p.ln_s + npk), svg = *reinterpret_cast<const f4*>(p.ln_s + npk + 32);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float a = st[1] * (va[r] * p.alpha - st[0] * sva[r]) + ba[r];
                                const float g = st[1] * (vg[r] * p.alpha - st[0] * svg[r]) + bg[r];
                                o[r] = (half_t)(SDMI_GELU_SIG ? geglu_gate(a, g) : a * gelu_erf(g));
                            }
                        } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float a = fmaf(va[r], p.alpha, ba[r]), g = fmaf(vg[r], p.alpha, bg[r]);
                            o[r] = (half_t)(SDMI_GELU_SIG ? geglu_gate(a, g) : a * gelu_erf(g));
                        }
                        }
                        *reinterpret_cast<h4*>((half_t*)p.out + ob + (long)m * p.ldo + nout) = o;
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wc * WTN + j * 16 + (lane >> 4) * 4;
            f4 v = acc[i][j];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= p.alpha;
            if constexpr (LNF) {
                const f2e st = lst[i];
                const f4 sv = *reinterpret_cast<const f4*>(p.ln_s + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = st[1] * (v[r] - st[0] * sv[r]);
            }
            if (p.bias) {
                if (flags & EP_BIAS_ROW) {
                    const float bb = p.bias[m];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += bb;
                } else {
                    f4 bb;
                    if constexpr (PIPE) bb = bcol[j];
                    else bb = *reinterpret_cast<const f4*>(p.bias + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(bb[r], p.bias_scale, v[r]);
                }
            }
            if (p.rowbias) v += *reinterpret_cast<const f4*>(p.rowbias + (long)b * p.ldrb + n);      // [B][N] fp32: cache-resident
            if (flags & (EP_QUICK_GELU | EP_GELU)) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[r] = (flags & EP_QUICK_GELU) ? v[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v[r])) : gelu_erf(v[r]);
            }
            if (!LNF && p.resid) {
                h4 rr;
                if constexpr (PIPE) rr = rres[i & 1][j];
                else rr = *reinterpret_cast<const h4*>(p.resid + rbs + (long)m * p.ldr + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
                if (const half_t* rlo = gemm_resid_lo(p)) {  // (hi, lo) residual (EP_HILO)
                    const h4 rl = *reinterpret_cast<const h4*>(rlo + rbs + (long)m * p.ldr + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)rl[r];
                }
            }
            if (flags & EP_NCHW) {
                const int pix = m - b * p.rows_per_batch;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.n_real)
                        ((float*)p.out)[ob + ((long)b * p.n_real + n + r) * p.rows_per_batch + pix] = v[r];
            } else if (flags & EP_OUT_F32) {
                *reinterpret_cast<f4*>((float*)p.out + ob + (long)m * p.ldo + n) = v;
            } else {
                h4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
                *reinterpret_cast<h4*>((half_t*)p.out + ob + (long)m * p.ldo + n) = o;
                if (half_t* olo = gemm_out_lo(p)) {          // what the fp16 rounding dropped, as a second fp16 tensor (EP_HILO)
                    h4 l;
#pragma unroll
                    for (int r = 0; r < 4; ++r) l[r] = (half_t)(v[r] - (float)o[r]);
                    *reinterpret_cast<h4*>(olo + ob + (long)m * p.ldo + n) = l;
                }
            }
        }
    }
}

// LDS swizzle: 16-byte chunk c of tile row r is stored in slot c ^ swz(r).  BK=64 (128-byte rows): r & 7.
// BK=32 (64-byte rows, 4 rows per 256-byte bank row): f((r>>2)&3) with f = {0,2,3,1}, which makes every ds_read_b128
// lane group {(rows 0-3,c),(rows 12-15,c),(rows 4-11,c^1)} land on 16 distinct 16-byte slots.
template <int BK>
__device__ __forceinline__ int swz(int r) {
    if constexpr (BK == 64) return r & 7;
    else return (0x78 >> (((r >> 2) & 3) * 2)) & 3;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LIN (round 5): the 1x1 / linear launches (one tap, stride 1, no upsampling, no wrap: row m of the GEMM is pixel m of the source).
// The general gather rebuilds every load's address from (tap, channel block, row) each K step — ~40 VALU / SALU instructions per
// 16-byte load, 315 instructions around the 16 MFMAs of a 128x64 tile's K step: the 4-wave tiles were ISSUE-bound on address
// arithmetic, not latency-bound (M4096 N1280 K1280: 2.5 workgroups per CU x 20 steps x ~1400 cycles = the 30 us every tile shape
// measured in rounds 3-4).  Here every load slot keeps a running pointer: one 64-bit add per load per K step.
template <int BM, int BN, int WR, int WC, int BK, bool GLDS, bool GEGLU, bool TR = false, bool KORD = false, bool STATS = false, int LNM = 0, int NS = 2,
          bool LIN = false, bool LIN3 = false>
__global__ __launch_bounds__(WR * WC * 64) void gemm_mfma_kernel(GemmP p) {
    // LIN3: the same idea for the plain 3x3 convolutions (stride 1, padding 1, no upsampling, no wrap, tap-major walk): per load slot the
    // pointer of the row's own pixel in the current source and a 9-bit mask of the taps that fall inside the image; per K step one
    // block-uniform offset ((dy - 1) Wi + dx - 1) lda + channel — a bit test, a 64-bit add and a select per load.
    static_assert(!(LIN || LIN3) || (GLDS && !KORD), "the lean walks exist for the LDS-direct path");
    static_assert(!(LIN && LIN3), "one walk");
    constexpr bool LEAN = LIN || LIN3;
    if (p.gate && *p.gate == 0) return;
    constexpr int NT = WR * WC * 64;
    constexpr int WTM = BM / WR, WTN = BN / WC;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int CPR = BK / 8;                          // 16-byte chunks per tile row
    constexpr int ROWB = BK * 2;                         // bytes per tile row
    constexpr int A_IT = BM * CPR / NT, B_IT = BN * CPR / NT;   // chunk loads per thread per stage
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int STAGE = A_BYTES + B_BYTES;
    static_assert((BM * CPR) % NT == 0 && (BN * CPR) % NT == 0, "tile must split evenly over the threads");
    static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile must be a multiple of the MFMA tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // wave-uniform: LDS bases and tile offsets stay scalar
    const int wr = wave / WC, wc = wave % WC;

    // ---- XCD-aware tile id remap (bijective for any grid size) -------------------------------------------
    const int tiles_n = p.N / BN;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, sub = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + sub;
    }
    int tile_m, tile_n;
    if (p.tile_order) {                                  // M first: an XCD's run of tiles shares the weight panel
        const int tiles_m = (p.M + BM - 1) / BM;
        tile_n = bid / tiles_m; tile_m = bid - tile_n * tiles_m;
    } else {
        tile_m = bid / tiles_n; tile_n = bid - tile_m * tiles_n;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const long z = blockIdx.z;
    const half_t* a0 = p.a0 + z * p.a_bs;
    const half_t* a1 = p.a1 ? p.a1 + z * p.a_bs : nullptr;
    const half_t* wbase = p.w + z * p.w_bs;

    // ---- per-thread row bookkeeping for the A gather ---------------------------------------------------------
    GRow rows[LEAN ? 1 : A_IT];
    const int ylim = p.up ? 2 * p.Hi : p.Hi, xlim = p.up ? 2 * p.Wi : p.Wi;
#pragma unroll
    for (int it = 0; it < (LEAN ? 0 : A_IT); ++it) {
        const int idx = it * NT + tid;
        const int m = m0 + idx / CPR;
        GRow gr;
        gr.ok = m < p.M;
        const int mm = gr.ok ? m : 0;
        const int b = mm / p.rows_per_batch;
        const int rem = mm - b * p.rows_per_batch;
        const int yo = rem / p.Wo;
        const int xo = rem - yo * p.Wo;
        gr.yb = p.up ? yo - 1 : yo * p.stride - p.pad;
        gr.xb = p.up ? xo - 1 : xo * p.stride - p.pad;
        gr.pixbase = b * p.Hi * p.Wi;
        rows[it] = gr;
    }

    f4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    int nk = p.K / BK;
    int k_first = 0;                                     // first BK-step of this block (split-K slice)
    if (p.splitk > 1) {
        k_first = blockIdx.y * p.splitk_steps;
        nk = min(nk - k_first, p.splitk_steps);
    }
    uint4 ra[GLDS ? 1 : A_IT], rb[GLDS ? 1 : B_IT];

    // LIN: running pointer + per-step increment of every load slot (rows beyond M / columns beyond n_valid: the zero page, increment 0);
    // adelta = what takes a slot from the end of source 0 to the start of source 1 (two-source K = a channel concatenation)
    const half_t* acur[LEAN ? A_IT : 1];
    const half_t* bcur[LEAN ? B_IT : 1];
    int ainc[LIN ? A_IT : 1], binc[LEAN ? B_IT : 1];
    long adelta[LEAN ? A_IT : 1];
    unsigned amask[LIN3 ? A_IT : 1];
    // (tap, cbase) of the NEXT stage to issue; advanced incrementally (no integer division in the K loop)
    int nx_tap = 0, nx_cbase = LIN ? k_first * BK : 0;
    if constexpr (LIN3) {
        const int k0 = k_first * BK;
        nx_tap = k0 / p.cin;
        nx_cbase = k0 - nx_tap * p.cin;
        const bool first = nx_cbase < p.c0;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int idx = it * NT + tid;
            const int r = idx / CPR, c = (idx % CPR) ^ swz<BK>(r);
            const int m = m0 + r;
            const bool ok = m < p.M;
            const int mm = ok ? m : 0;
            const int b = mm / p.rows_per_batch;
            const int rem = mm - b * p.rows_per_batch;
            const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = yo + t / 3 - 1, xx = xo + t % 3 - 1;
                mk |= (unsigned)(ok && (unsigned)yy < (unsigned)p.Hi && (unsigned)xx < (unsigned)p.Wi) << t;
            }
            amask[it] = mk;
            const half_t* q0 = a0 + (long)mm * p.lda0 + c * 8;                      // the row's own pixel, channel 0 of source 0
            const half_t* q1 = a1 ? a1 + (long)mm * p.lda1 + c * 8 : p.zero;
            acur[it] = first ? q0 : q1;
            adelta[it] = a1 ? (long)(reinterpret_cast<const char*>(q1) - reinterpret_cast<const char*>(q0)) : 0;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int idx = it * NT + tid;
            const int r = idx / CPR, c = (idx % CPR) ^ swz<BK>(r);
            const bool ok = n0 + r < p.n_valid;
            bcur[it] = ok ? wbase + (long)(n0 + r) * p.ldw + k0 + c * 8 : p.zero;
            binc[it] = ok ? BK : 0;
        }
    }
    if constexpr (LIN) {
        const int k0 = k_first * BK;
        const bool first = k0 < p.c0;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int idx = it * NT + tid;
            const int r = idx / CPR, c = (idx % CPR) ^ swz<BK>(r);
            const int m = m0 + r;
            const bool ok = m < p.M;
            const half_t* q0 = a0 + (long)m * p.lda0 + c * 8;                       // channel 0 of source 0, this slot's chunk
            const half_t* q1 = a1 ? a1 + (long)m * p.lda1 + c * 8 : p.zero;
            acur[it] = !ok ? p.zero : first ? q0 + k0 : q1 + (k0 - p.c0);
            ainc[it] = ok ? BK : 0;
            adelta[it] = ok && a1 ? (long)(reinterpret_cast<const char*>(q1) - reinterpret_cast<const char*>(q0 + p.c0)) : 0;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int idx = it * NT + tid;
            const int r = idx / CPR, c = (idx % CPR) ^ swz<BK>(r);
            const bool ok = n0 + r < p.n_valid;
            bcur[it] = ok ? wbase + (long)(n0 + r) * p.ldw + k0 + c * 8 : p.zero;
            binc[it] = ok ? BK : 0;
        }
    }

    if (!LEAN && k_first > 0) {
        if constexpr (KORD) {                            // channel block outer, tap inner (3x3 convs, GemmP::korder)
            const int blk = k_first / 9;
            nx_tap = k_first - blk * 9;
            nx_cbase = blk * BK;
        } else {
            const int k0 = k_first * BK;
            nx_tap = k0 / p.cin;
            nx_cbase = k0 - nx_tap * p.cin;
        }
    }
    auto stage_issue = [&](int sb) {
        if constexpr (LIN3) {
            char* abuf = smem + sb * STAGE;
            char* bbuf = abuf + A_BYTES;
            const int tap = nx_tap, cbase = nx_cbase;
            const bool first = cbase < p.c0;
            const int dy = (tap * 11) >> 5, dx = tap - dy * 3;
            const long uoff = (long)((dy - 1) * p.Wi + (dx - 1)) * (first ? p.lda0 : p.lda1) + (first ? cbase : cbase - p.c0);   // elements, block-uniform
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const half_t* g = ((amask[it] >> tap) & 1u) ? acur[it] + uoff : p.zero;
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(abuf + (it * NT + wave * 64) * 16), 16, 0, 0);
            }
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                __builtin_amdgcn_global_load_lds((gptr_t)bcur[it], (lptr_t)(bbuf + (it * NT + wave * 64) * 16), 16, 0, 0);
                bcur[it] += binc[it];
            }
            nx_cbase += BK;
            const bool wrapped = nx_cbase >= p.cin;
            if (wrapped) { nx_cbase = 0; ++nx_tap; }
            if (a1 && (wrapped || nx_cbase == p.c0)) {     // the next stage reads the other source (c0, c1 are multiples of BK: gemm_mfma_supported)
#pragma unroll
                for (int it = 0; it < A_IT; ++it)
                    acur[it] = reinterpret_cast<const half_t*>(reinterpret_cast<const char*>(acur[it]) + (wrapped ? -adelta[it] : adelta[it]));
            }
        } else if constexpr (LIN) {
            char* abuf = smem + sb * STAGE;
            char* bbuf = abuf + A_BYTES;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                __builtin_amdgcn_global_load_lds((gptr_t)acur[it], (lptr_t)(abuf + (it * NT + wave * 64) * 16), 16, 0, 0);
                acur[it] += ainc[it];
            }
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                __builtin_amdgcn_global_load_lds((gptr_t)bcur[it], (lptr_t)(bbuf + (it * NT + wave * 64) * 16), 16, 0, 0);
                bcur[it] += binc[it];
            }
            nx_cbase += BK;
            if (nx_cbase == p.c0 && a1) {                  // the next stage starts source 1 (c0 is a multiple of BK: gemm_mfma_supported)
#pragma unroll
                for (int it = 0; it < A_IT; ++it)
                    acur[it] = reinterpret_cast<const half_t*>(reinterpret_cast<const char*>(acur[it]) + adelta[it]);
            }
        } else {
        const int tap = nx_tap, cbase = nx_cbase, k0 = nx_tap * p.cin + nx_cbase;
        if constexpr (KORD) {
            if (++nx_tap == 9) { nx_tap = 0; nx_cbase += BK; }
        } else {
            nx_cbase += BK;
            if (nx_cbase >= p.cin) { nx_cbase = 0; ++nx_tap; }
        }
        // block-uniform part of the address
        const bool first = cbase < p.c0;
        const half_t* src = first ? a0 : a1;
        const int cch = first ? cbase : cbase - p.c0;
        const int lda = first ? p.lda0 : p.lda1;
        int dy = 0, dx = 0;
        if (p.taps == 9) { dy = (tap * 11) >> 5; dx = tap - dy * 3; }      // tap / 3 for tap in [0, 9)
        char* abuf = smem + sb * STAGE;
        char* bbuf = abuf + A_BYTES;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int idx = it * NT + tid;
            const int r = idx / CPR, c = (idx % CPR) ^ swz<BK>(r);
            const GRow gr = rows[it];
            int yr = gr.yb + dy, xr = gr.xb + dx;
            if (p.flags & EP_WRAP) {                       // circular padding: one wrap is enough for a 3x3 window
                yr = yr < 0 ? yr + ylim : (yr >= ylim ? yr - ylim : yr);
                xr = xr < 0 ? xr + xlim : (xr >= xlim ? xr - xlim : xr);
            }
            const bool ok = gr.ok && (unsigned)yr < (unsigned)ylim && (unsigned)xr < (unsigned)xlim;
            const int pix = gr.pixbase + (yr >> p.up) * p.Wi + (xr >> p.up);
            const half_t* g = ok ? src + (long)pix * lda + (cch + c * 8) : p.zero;
            if constexpr (GLDS) {
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(abuf + (it * NT + wave * 64) * 16), 16, 0, 0);
            } else {
                ra[it] = *reinterpret_cast<const uint4*>(g);
            }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int idx = it * NT + tid;
            const int r = idx / CPR, c = (idx % CPR) ^ swz<BK>(r);
            const half_t* g = (n0 + r < p.n_valid) ? wbase + (long)(n0 + r) * p.ldw + k0 + c * 8 : p.zero;
            if constexpr (GLDS) {
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(bbuf + (it * NT + wave * 64) * 16), 16, 0, 0);
            } else {
                rb[it] = *reinterpret_cast<const uint4*>(g);
            }
        }
        }
    };
    auto stage_commit = [&](int sb) {
        if constexpr (!GLDS) {
            char* abuf = smem + sb * STAGE;
            char* bbuf = abuf + A_BYTES;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) *reinterpret_cast<uint4*>(abuf + (it * NT + tid) * 16) = ra[it];
#pragma unroll
            for (int it = 0; it < B_IT; ++it) *reinterpret_cast<uint4*>(bbuf + (it * NT + tid) * 16) = rb[it];
        }
    };
    auto compute = [&](int sb) {
        const char* abuf = smem + sb * STAGE;
        const char* bbuf = abuf + A_BYTES;
        const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            const int kc = ks * 4 + lk;
            const int sw = (kc ^ swz<BK>(lr)) << 4;       // swz(row) == swz(lane&15): tile bases are multiples of 16
            h8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const h8*>(abuf + (wr * WTM + i * 16 + lr) * ROWB + sw);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const h8*>(bbuf + (wc * WTN + j * 16 + lr) * ROWB + sw);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    // LIN, ring tiles: one K step with the LDS-direct loads of the stage that refills the ring spread between the MFMAs (one 1 KiB piece
    // per GAP MFMAs) instead of in a burst ahead of them — a burst of LPS pieces from all four waves queues in the CU's address unit (17 cycles per piece at the
    // 60 B/clk the L2 delivers, tools/micro/fill_rate.hip) while the MFMA pipe waits; fragment reads of the second half-step go out
    // ahead of the first half's MFMAs.  Same MFMA order as compute(): same bits.
    auto compute_lin = [&](int sb, int fillsb, bool more) {
        constexpr int KS = BK / 32, LPS = A_IT + B_IT, NMMA = KS * TM * TN;
        constexpr int GAP = NMMA / LPS > 0 ? NMMA / LPS : 1;
        const char* abuf = smem + sb * STAGE;
        const char* bbuf = abuf + A_BYTES;
        char* fa = smem + fillsb * STAGE;
        char* fb = fa + A_BYTES;
        const int lr = lane & 15, lk = lane >> 4;
        h8 af[2][TM], bf[2][TN];
        auto frags = [&](int ks) {
            const int sw = ((ks * 4 + lk) ^ swz<BK>(lr)) << 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[ks & 1][i] = *reinterpret_cast<const h8*>(abuf + (wr * WTM + i * 16 + lr) * ROWB + sw);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[ks & 1][j] = *reinterpret_cast<const h8*>(bbuf + (wc * WTN + j * 16 + lr) * ROWB + sw);
        };
        frags(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ks & 1][j], af[ks & 1][i], acc[i][j], 0, 0, 0);
                    // the next half-step's fragments go out a quarter into this one's MFMAs: with an LDS-direct load pending the compiler
                    // waits for ALL outstanding LDS reads at the first use of any (lgkmcnt(0), never a partial count) — reads issued ahead
                    // of the MFMAs would all be waited for before the first one
                    if (ks + 1 < KS && i * TN + j == (TM * TN) / 4) {
                        __builtin_amdgcn_sched_barrier(0);
                        frags(ks + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const int m = (ks * TM + i) * TN + j;              // a constant once the loops are unrolled
                    if (m % GAP == GAP - 1 && m / GAP < LPS) {
                        const int q = m / GAP;
                        __builtin_amdgcn_sched_barrier(0);
                        if (more) {
                            if (q < A_IT) {
                                __builtin_amdgcn_global_load_lds((gptr_t)acur[q < A_IT ? q : 0], (lptr_t)(fa + (q * NT + wave * 64) * 16), 16, 0, 0);
                                acur[q < A_IT ? q : 0] += ainc[q < A_IT ? q : 0];
                            } else {
                                const int b = q >= A_IT ? q - A_IT : 0;
                                __builtin_amdgcn_global_load_lds((gptr_t)bcur[b], (lptr_t)(fb + (b * NT + wave * 64) * 16), 16, 0, 0);
                                bcur[b] += binc[b];
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        }
        static_assert(NMMA / GAP >= LPS, "every piece of a stage must find its slot between the MFMAs");
        if (more) {
            nx_cbase += BK;
            if (nx_cbase == p.c0 && a1) {
#pragma unroll
                for (int it = 0; it < A_IT; ++it)
                    acur[it] = reinterpret_cast<const half_t*>(reinterpret_cast<const char*>(acur[it]) + adelta[it]);
            }
        }
    };

    if constexpr (NS > 2) {
        // ---- ring of NS stages (round 4): NS - 1 K steps of LDS-direct loads in flight, counted vmcnt, one barrier per K step -----
        // The two-stage loop below waits for step k+1's loads at the end of step k: a workgroup's K loop is a chain of nk load
        // latencies (M4096 N1280 K1280 on 128x64 tiles: 20 steps x 1.6 us, MFMA pipe 23 % busy, profiles/r03_shape_tuning.md).  Here
        // the loads of step k + NS - 1 are issued at step k (into the slot step k - 1 just released), so a step waits for data issued
        // NS - 1 steps earlier.  Every thread issues LPS loads per stage and vmcnt retires in order, so "at most rem * LPS outstanding"
        // means stage k has landed.  Same accumulation order as the two-stage kernel: identical bits.
        static_assert(GLDS && NS <= 4, "the ring form exists for the LDS-direct path only");
        constexpr int LPS = A_IT + B_IT;
        static_assert((NS - 2) * LPS < 64, "vmcnt is 6 bits");
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (s < nk) stage_issue(s);
        int slot = 0, fill = NS - 1;                         // slot of step kt; slot the next refill goes to
        for (int kt = 0; kt < nk; ++kt) {
            const int rem = min(nk - 1 - kt, NS - 2);        // stages issued beyond step kt: they may stay in flight
            if (rem >= 2) wait_vmcnt<2 * LPS>();
            else if (rem == 1) wait_vmcnt<LPS>();
            else wait_vmcnt<0>();
            // stage kt visible to all; everyone is done with step kt - 1's slot (its fragment reads fed MFMAs already issued).  A bare
            // s_barrier: __syncthreads() is fence + barrier and the fence compiles to `s_waitcnt vmcnt(0)` — with it every step waited
            // for the NS - 2 stages just issued and the ring ran as a two-stage loop (round 4's "ring = no faster", r04_ring_check.txt,
            // measured THAT; found in the ISA in round 5)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (LIN) {
                compute_lin(slot, fill, kt + NS - 1 < nk);
            } else {
                if (kt + NS - 1 < nk) stage_issue(fill);
                compute(slot);
            }
            if (++slot == NS) slot = 0;
            if (++fill == NS) fill = 0;
        }
        __syncthreads();
    } else {
    // ---- main loop: double-buffered, one barrier per K step ------------------------------------------------
    stage_issue(0);
    stage_commit(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        // (LIN: the loads stay in a burst AHEAD of the MFMAs here — spread between them, compute_lin's way, they are issued later and this
        // loop waits for them at the end of the same step: 128x160 on M4096 N1280 K1280 24.9 -> 34.0 us, r05_ring_check_linear_walk.txt)
        if (more) stage_issue(cur ^ 1);
        compute(cur);
        if (more) stage_commit(cur ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    }

    gemm_epilogue<TM, TN, WTM, WTN, GEGLU, TR, WR, BN, STATS, LNM>(p, acc, m0, n0, wr, wc, lane, z, smem);
}

// ---------------------------------------------------------------------------------------------------------------
// Ping-pong kernel for the 256-row tiles: BK = 64, 8 waves as 2 (M) x 4 (N).  Every SIMD holds one wave of each
// wave-row group; the two groups run the same program ONE BARRIER APART, so that while one group is in its MFMA section
// the other one is in its load section (LDS -> register fragment reads, LDS-direct global loads, address arithmetic,
// waits).  With all waves in lockstep (the two-stage kernel above) the MFMA pipe idles through every load section.
//
// A K tile is computed in 4 phases; phase q multiplies rows [32q, 32q+32) of the wave's 128-row sub-tile with all of the
// wave's B fragments (read once per K tile, in phase 0, and kept in registers).  Each phase is
//     L: ds_reads of this phase's fragments | issue one group of LDS-direct loads | counted vmcnt wait | lgkmcnt(0) | s_barrier
//     M: s_setprio 1 | 2 x 2 x TN MFMAs | s_setprio 0 | s_barrier
// (measured, tools/micro/pingpong_steps.hip: bare ping-pong MFMA sections run at 98 % of the MFMA rate; LDS reads and
// LDS-direct loads issued from the L section cost ~7 % together, the same reads issued from the M section ~15 % alone).
// LDS is double-buffered per K tile but refilled PIECEWISE as soon as a piece has been read for the last time, which puts
// loads 5-7 phases (~1.5 K tiles) ahead of their use instead of less than one K tile:
//     piece        last read     refilled for tile T+2 in          needed in
//     B half 0/1   L(0) of T     L(1) / L(2) of T                  L(0) of T+2
//     A rows 0-63  L(0), L(1)    L(3) of T                         L(0), L(1) of T+2
//     A rows 64-   L(2), L(3)    L(0) of T+1                       L(2), L(3) of T+2
// ("A rows" of both wave-row halves; the half-h rows are staged by the waves of group h and read only by group h.)
// vmcnt retires in issue order and every thread issues the same number of loads per group, so two counted waits per tile
// order everything (no vmcnt(0) in the steady state):
//     W1 in L(3) of T : B + A rows 0-63 of tile T+1 landed  <=> at most N1 = 4 + BU younger loads outstanding
//     W2 in L(1) of T : A rows 64-127 of tile T landed      <=> at most N2 = 4 + BU + BU/2 younger loads outstanding
// RAW: a piece is read only after the issuing waves' counted wait AND a barrier (group 1 waits one slot after group 0;
// the shared B pieces are first read by group 0 two slots after its own W1, one slot after group 1's).  WAR: lgkmcnt(0)
// precedes the barrier that closes every L section, so a piece is refilled only after both groups' reads have returned.
// Accumulation order over k equals the two-stage kernel's: results are bit-identical to it (tests/test_gpu_ops.py).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ long long stamp() {           // s_memtime; callers sit at points where lgkmcnt is already 0
    const long long t = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return t;
}

template <int BM, int BN, bool GEGLU, bool TIMING = false, bool TR = false, bool KORD = false, bool STATS = false, int LNM = 0>
__global__ __launch_bounds__(512) void gemm_mfma_pingpong_kernel(GemmP p) {
    if (p.gate && *p.gate == 0) return;
    constexpr int BK = 64, WC = 4, ROWB = 128;
    constexpr int WTM = BM / 2, WTN = BN / WC, TM = WTM / 16, TN = WTN / 16;
    constexpr int RP = 32, TMP = 2;                          // rows / MFMA row-tiles of the wave tile per phase
    constexpr int PH = WTM / RP;                             // phases per K tile: 4 (BM = 256) or 2 (BM = 128, see below)
    static_assert(PH == 4 || PH == 2, "BM must be 256 or 128");
    constexpr int BU = BN / 64;                              // 64-row pieces of the weight tile (one chunk per thread each)
    constexpr int NB0 = BU / 2, NB1 = BU - NB0;              // pieces in B half 0 / 1
    constexpr int N1 = 4 + BU, N2 = 4 + BU + NB0;            // PH = 4 wait thresholds
    // PH = 2 (128-row tiles): phase 0 reads B + A rows 0-31, phase 1 reads A rows 32-63.  Only two refill slots per tile:
    //   L(1) of T : all of B and A rows 0-31 of tile T+2 (their slots were last read in L(0) of T)
    //   L(0) of T+1: A rows 32-63 of tile T+2
    // and one counted wait in each: at most N3 = BU + 2 younger loads outstanding (2 phases of flight time).
    constexpr int N3 = BU + 2;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    static_assert(BN % 64 == 0 && WTN % 16 == 0 && N2 < 64, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;

    const int tiles_n = p.N / BN;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, sub = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + sub;
    }
    int tile_m, tile_n;
    if (p.tile_order) {                                  // M first: an XCD's run of tiles shares the weight panel
        const int tiles_m = (p.M + BM - 1) / BM;
        tile_n = bid / tiles_m; tile_m = bid - tile_n * tiles_m;
    } else {
        tile_m = bid / tiles_n; tile_n = bid - tile_m * tiles_n;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const long z = blockIdx.z;
    const half_t* a0 = p.a0 + z * p.a_bs;
    const half_t* a1 = p.a1 ? p.a1 + z * p.a_bs : nullptr;
    const half_t* wbase = p.w + z * p.w_bs;

    // ---- staging roles of this thread -----------------------------------------------------------------------------
    // A unit q (rows [32q, 32q+32) of both halves): group-h waves stage the half-h rows; thread -> row 8*(wave&3) + lane/8
    // of the unit's 32 rows, chunk lane & 7.  B piece i (64 rows): thread -> row tid / 8, chunk tid & 7.
    const int ch = tid & 7;
    const int a_rin = (wave & 3) * 8 + (lane >> 3);
    const int b_row = tid >> 3;
    // all rows a thread stages are congruent to lane/8 mod 8, so one source-side swizzle serves every piece
    const int src_chunk = (ch ^ (lane >> 3)) * 8;            // logical chunk (in halfs) whose data lands in slot `ch`
    // Gather bookkeeping per staged row, ONE register: image index << 24 | (yb & 0xfff) << 12 | (xb & 0xfff) with (yb, xb)
    // the window origin (a row past M gets yb = -2048, which fails every bounds test) — the launcher guarantees
    // B < 128 and window coordinates within +-2047 — plus the row's source pointer for the CURRENT (tap, source) segment
    // of K, rebuilt only when a K tile starts a new segment, so a steady-state load costs one 64-bit add.
    // Out-of-image taps point at the zero page (32 KB of zeros: any channel offset of the segment stays inside it).
    // (the packed words live in LDS behind the tile buffers: they are needed once per segment, and VGPRs are the scarce
    // resource of the 256x320 instantiation — a compiler spill would put scratch loads, i.e. vmcnt traffic, in the loop)
    int* rinfo = reinterpret_cast<int*>(smem + 2 * STAGE) + tid;        // [q * 512]
    const half_t* rptr[PH];
    const int ylim = p.up ? 2 * p.Hi : p.Hi, xlim = p.up ? 2 * p.Wi : p.Wi;
#pragma unroll
    for (int q = 0; q < PH; ++q) {
        const int m = m0 + wr * WTM + q * RP + a_rin;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int b = mm / p.rows_per_batch;
        const int rem = mm - b * p.rows_per_batch;
        const int yo = rem / p.Wo;
        const int xo = rem - yo * p.Wo;
        const int yb = ok ? (p.up ? yo - 1 : yo * p.stride - p.pad) : -2048;
        const int xb = p.up ? xo - 1 : xo * p.stride - p.pad;
        rinfo[q * 512] = (b << 24) | ((yb & 0xfff) << 12) | (xb & 0xfff);
        rptr[q] = p.zero;
    }
    const int img_pix = p.Hi * p.Wi;
    const bool plain = p.taps == 1 && p.stride == 1 && !p.up && p.Ho == p.Hi && p.Wo == p.Wi;
    const int a_lds = (wr * WTM + (wave & 3) * 8) * ROWB;    // wave-uniform: the wave's 8 rows inside unit 0
    const int b_lds = A_BYTES + wave * 8 * ROWB;             // the wave's 8 rows inside piece 0
    const half_t* b_src = wbase + (long)(n0 + b_row) * p.ldw + src_chunk;
    const long b_piece = 64L * p.ldw;                        // elements between consecutive 64-row pieces

    f4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    int nk = p.K / BK;
    int k_first = 0;
    if (p.splitk > 1) {
        k_first = blockIdx.y * p.splitk_steps;
        nk = min(nk - k_first, p.splitk_steps);
    }
    // (tap, cbase) of K tiles T+1 and T+2; advanced once per tile, no division in the loop
    int tap1, cb1, tap2, cb2;
    // KORD (compile time; launched for 3x3 convs when GemmP::korder is set): channel block outer, tap inner
    auto advance = [&](int& tap, int& cb) {
        if constexpr (KORD) {
            if (++tap == 9) { tap = 0; cb += BK; }
        } else {
            cb += BK;
            if (cb >= p.cin) { cb = 0; ++tap; }
        }
    };
    if constexpr (KORD) {
        const int blk = k_first / 9;
        tap1 = k_first - blk * 9;
        cb1 = blk * BK;
    } else {
        const int k0 = k_first * BK;
        tap1 = k0 / p.cin;
        cb1 = k0 - tap1 * p.cin;
    }

    // KORD fast path (stride-1/2 3x3 convs without the fused upsample / circular padding: every UNet and VAE conv but the six
    // upsampling ones).  With the channel-block-major K order EVERY K tile is a new tap, and the per-tile rebuild below (an LDS read,
    // two quarter-rate integer multiplies and a 64-bit multiply-add per staged row) was the longest thing in the L(0) / L(3)
    // sections (section timers, profiles/r02_gemm_sections_korder_timing.txt; A/B in profiles/r02_conv_korder.md).  A tap only shifts the source pixel by a block-uniform offset
    // (dy * Wi + dx) * lda, so the row keeps ONE pointer for the whole K loop — the window-origin pixel of the current source, with
    // the 9 tap-validity bits packed into address bits 48..56 — and a tile costs a bit test, a 64-bit add and a select per row.
    // (launch_gemm keeps the tap-major order — and so the non-KORD instantiation — for the upsampling / wrapping convs.)
    const int chunk8 = ((tid & 7) ^ (lane >> 3)) * 8;
    // KORD: source addresses of A units [q0, q0+nq) of the K tile at (tap, cbase).  In the steady state the loop calls this from the
    // M section BEFORE the L section that issues the loads (the MFMA stream has VALU issue slots to spare; the L sections are the
    // long pole of a phase: every VALU instruction there showed up 1:1 in the K-tile time).
    auto prep_a = [&](int q0, int nq, int tap, int cbase, bool force, const half_t* (&out)[PH / 2]) {
        const bool first = cbase < p.c0;
        const int seg0 = first ? 0 : p.c0;
        const int dy = (tap * 11) >> 5, dx = tap - dy * 3;
        const half_t* src = first ? a0 : a1;
        const int lda = first ? p.lda0 : p.lda1;
        if (force || (tap == 0 && cbase == seg0)) {       // block-uniform: first tile of a source (or of this K slice)
            asm volatile("");                             // keep this a real (scalar) branch: if-converted, the rebuild would run on every tile
#pragma unroll
            for (int q = q0; q < q0 + nq; ++q) {
                const int ri = rinfo[q * 512];
                const int yb = (int)((unsigned)ri << 8) >> 20, xb = (int)((unsigned)ri << 20) >> 20;
                unsigned colm = 0, mask = 0;
#pragma unroll
                for (int d = 0; d < 3; ++d) colm |= ((unsigned)(xb + d) < (unsigned)xlim ? 1u : 0u) << d;
#pragma unroll
                for (int d = 0; d < 3; ++d)
                    if ((unsigned)(yb + d) < (unsigned)ylim) mask |= colm << (3 * d);      // rows past M: yb = -2048, no bit set
                const long pix = (long)(ri >> 24) * img_pix + (long)yb * p.Wi + xb;          // may lie outside the image: only
                const unsigned long long addr = (unsigned long long)(src + pix * lda + chunk8);   // dereferenced under its bit
                rptr[q] = reinterpret_cast<const half_t*>((addr & 0xFFFFFFFFFFFFull) | ((unsigned long long)mask << 48));
            }
        }
        const long toff = (long)(dy * p.Wi + dx) * lda + (cbase - seg0);      // uniform: tap shift + channel offset (elements)
        // padding lanes read zeros from a line of the 32 KB zero page that moves with the tile: one fixed line would be
        // a hot spot of a single L2 channel for every border row of every workgroup
        const half_t* zp = p.zero + (((cbase - seg0) + tap * 64) & 0x3FC0) + chunk8;
#pragma unroll
        for (int q = q0; q < q0 + nq; ++q) {
            const unsigned long long raw = reinterpret_cast<unsigned long long>(rptr[q]);
            const bool ok = ((unsigned)(raw >> 48) >> tap) & 1u;
            out[q - q0] = ok ? reinterpret_cast<const half_t*>(raw & 0xFFFFFFFFFFFFull) + toff : zp;
        }
    };
    auto issue_a_at = [&](int q0, int nq, int sb, const half_t* (&np)[PH / 2]) {
        char* base = smem + sb * STAGE + a_lds;
#pragma unroll
        for (int q = q0; q < q0 + nq; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)np[q - q0], (lptr_t)(base + q * RP * ROWB), 16, 0, 0);
    };
    // A units [q0, q0+nq) of the K tile at (tap, cbase) into LDS buffer sb
    auto issue_a = [&](int q0, int nq, int tap, int cbase, int sb, bool force) {
        char* base = smem + sb * STAGE + a_lds;
        const bool first = cbase < p.c0;
        const int seg0 = first ? 0 : p.c0;
        int dy = 0, dx = 0;
        if (p.taps == 9) { dy = (tap * 11) >> 5; dx = tap - dy * 3; }
        if constexpr (KORD) {
            const half_t* np[PH / 2];
            prep_a(q0, nq, tap, cbase, force, np);
#pragma unroll
            for (int q = q0; q < q0 + nq; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)np[q - q0], (lptr_t)(base + q * RP * ROWB), 16, 0, 0);
            return;
        }
        if (force || cbase == seg0) {            // block-uniform: the tile opens a new (tap, source) segment
            const half_t* src = first ? a0 : a1;
            const int lda = first ? p.lda0 : p.lda1;
#pragma unroll
            for (int q = q0; q < q0 + nq; ++q) {
                bool ok;
                int pix;
                if (plain) {                              // 1x1, stride 1: the GEMM row IS the pixel index
                    pix = m0 + wr * WTM + q * RP + a_rin;
                    ok = pix < p.M;
                } else {
                    const int ri = rinfo[q * 512];
                    int yr = ((int)((unsigned)ri << 8) >> 20) + dy, xr = ((int)((unsigned)ri << 20) >> 20) + dx;     // sign-extended 12-bit fields
                    if ((p.flags & EP_WRAP) && yr > -1024) {   // circular padding (rows past M keep yb = -2048 and stay invalid)
                        yr = yr < 0 ? yr + ylim : (yr >= ylim ? yr - ylim : yr);
                        xr = xr < 0 ? xr + xlim : (xr >= xlim ? xr - xlim : xr);
                    }
                    ok = (unsigned)yr < (unsigned)ylim && (unsigned)xr < (unsigned)xlim;
                    pix = (ri >> 24) * img_pix + (yr >> p.up) * p.Wi + (xr >> p.up);
                }
                rptr[q] = (ok ? src + (long)pix * lda : p.zero) + chunk8;
            }
        }
        const int coff = cbase - seg0;                   // uniform channel offset inside the segment
#pragma unroll
        for (int q = q0; q < q0 + nq; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(rptr[q] + coff), (lptr_t)(base + q * RP * ROWB), 16, 0, 0);
    };
    // B pieces [i0, i0+n) of the K tile at (tap, cbase) into LDS buffer sb
    auto issue_b = [&](int i0, int n, int tap, int cbase, int sb) {
        char* base = smem + sb * STAGE + b_lds;
        const half_t* bt = b_src + (long)tap * p.cin + cbase;
#pragma unroll
        for (int i = i0; i < i0 + n; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(bt + i * b_piece), (lptr_t)(base + i * 64 * ROWB), 16, 0, 0);
    };

    const int lr = lane & 15, lk = lane >> 4;
    const int sw0 = (lk ^ (lr & 7)) << 4, sw1 = ((4 + lk) ^ (lr & 7)) << 4;
    const int a_rd = (wr * WTM + lr) * ROWB, b_rd = A_BYTES + (wc * WTN + lr) * ROWB;
    h8 bf[2][TN], af[2][TMP];
    auto read_b = [&](const char* sbuf) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bf[0][j] = *reinterpret_cast<const h8*>(sbuf + b_rd + j * 16 * ROWB + sw0);
            bf[1][j] = *reinterpret_cast<const h8*>(sbuf + b_rd + j * 16 * ROWB + sw1);
        }
    };
    auto read_a = [&](const char* sbuf, int ph) {
#pragma unroll
        for (int i = 0; i < TMP; ++i) {
            af[0][i] = *reinterpret_cast<const h8*>(sbuf + a_rd + (ph * RP + i * 16) * ROWB + sw0);
            af[1][i] = *reinterpret_cast<const h8*>(sbuf + a_rd + (ph * RP + i * 16) * ROWB + sw1);
        }
    };

    // ---- prologue: all of tile 0; of tile 1 everything except the group the first L(0) issues -------------------------
    issue_b(0, NB0, tap1, cb1, 0);
    issue_b(NB0, NB1, tap1, cb1, 0);
    issue_a(0, PH / 2, tap1, cb1, 0, true);
    issue_a(PH / 2, PH / 2, tap1, cb1, 0, true);
    advance(tap1, cb1);                                  // (tap1, cb1) = tile 1
    tap2 = tap1; cb2 = cb1;
    if (nk > 1) {
        issue_b(0, NB0, tap1, cb1, 1);
        issue_b(NB0, NB1, tap1, cb1, 1);
        issue_a(0, PH / 2, tap1, cb1, 1, true);
        wait_vmcnt<(PH == 4 ? N1 : N3)>();
    } else {
        wait_vmcnt<0>();
    }
    const half_t* anext[PH / 2];
#pragma unroll
    for (int q = 0; q < PH / 2; ++q) anext[q] = p.zero;
    if constexpr (KORD && !TIMING) {
        if (nk > 1) prep_a(PH / 2, PH / 2, tap1, cb1, true, anext);      // what the first L(0) issues: the second half of tile 1
    }
    advance(tap2, cb2);                                  // (tap2, cb2) = tile 2
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wr == 1) {                                       // group 1 runs one barrier behind group 0 from here on
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    long long tm[17] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // TIMING: per phase index: cycles in L work, barrier a, M issue, barrier b; [16] K tiles
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        const char* sbuf = smem + cur * STAGE;
#pragma unroll
        for (int ph = 0; ph < PH; ++ph) {
            long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
            if (TIMING) t0 = stamp();
            // ================= L section: fragment reads of this phase, refill one group, counted waits =================
            // (LDS reads cost nothing here, beside the other group's MFMAs; issued from the M section — before or after
            // the MFMAs — the same reads add ~80 cycles per phase: tools/micro/pingpong_steps.hip)
            if (!(TIMING && (p.flags & EP_DBG_NO_DSREAD)) || t == 0) {
                if (ph == 0) read_b(sbuf);
                read_a(sbuf, ph);
            }
            if (!(TIMING && (p.flags & EP_DBG_NO_GLDS))) {
                if (PH == 4) {
                    if (ph == 0) {
                        if (t + 1 < nk) { if constexpr (KORD && !TIMING) issue_a_at(2, 2, cur ^ 1, anext); else issue_a(2, 2, tap1, cb1, cur ^ 1, false); }
                    } else if (ph == 1) {
                        if (t + 2 < nk) issue_b(0, NB0, tap2, cb2, cur);
                    } else if (ph == 2) {
                        if (t + 2 < nk) issue_b(NB0, NB1, tap2, cb2, cur);
                    } else {
                        if (t + 2 < nk) { if constexpr (KORD && !TIMING) issue_a_at(0, 2, cur, anext); else issue_a(0, 2, tap2, cb2, cur, false); }
                    }
                } else {
                    if (ph == 0) {
                        if (t + 1 < nk) { if constexpr (KORD && !TIMING) issue_a_at(1, 1, cur ^ 1, anext); else issue_a(1, 1, tap1, cb1, cur ^ 1, false); }
                    } else if (t + 2 < nk) {
                        issue_b(0, BU, tap2, cb2, cur);
                        if constexpr (KORD && !TIMING) issue_a_at(0, 1, cur, anext); else issue_a(0, 1, tap2, cb2, cur, false);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (TIMING && (p.flags & EP_DBG_NO_VMWAIT)) {
            } else if (PH == 4) {
                if (ph == 1) {
                    if (t + 2 < nk) wait_vmcnt<N2>(); else wait_vmcnt<0>();      // W2: A rows 64-127 of tile t
                } else if (ph == 3) {
                    if (t + 2 < nk) wait_vmcnt<N1>(); else wait_vmcnt<0>();      // W1: B and A rows 0-63 of tile t+1
                }
            } else {
                if (ph == 0) {
                    if (t + 1 < nk) wait_vmcnt<N3>(); else wait_vmcnt<0>();      // A rows 32-63 of tile t
                } else {
                    if (t + 2 < nk) wait_vmcnt<N3>(); else wait_vmcnt<0>();      // B and A rows 0-31 of tile t+1
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // this phase's fragments are in registers
            // (the 8-phase template's order — barrier first, LDS wait behind it — was measured: C1 job -0.35 %, conv class -1.2 %;
            // it is only provably WAR-safe with the weight refills one phase later: profiles/r02_attention_experiments.md section 5)
            if (TIMING) t1 = stamp();
            if (!(TIMING && (p.flags & EP_DBG_NO_BAR_A))) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (TIMING) t2 = stamp();
            // ================= M section: MFMAs only =================
            __builtin_amdgcn_s_setprio(1);
            if constexpr (KORD && !TIMING) {
                // addresses of the A loads the NEXT L section issues (both belong to tile t+2 = (tap2, cb2))
                if (PH == 4) {
                    if (ph == 2) { if (t + 2 < nk) prep_a(0, 2, tap2, cb2, false, anext); }
                    else if (ph == 3) { if (t + 2 < nk) prep_a(2, 2, tap2, cb2, false, anext); }
                } else {
                    if (ph == 0) { if (t + 2 < nk) prep_a(0, 1, tap2, cb2, false, anext); }
                    else { if (t + 2 < nk) prep_a(1, 1, tap2, cb2, false, anext); }
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < TMP; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[ph * TMP + i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks][i], bf[ks][j], acc[ph * TMP + i][j], 0, 0, 0)
                                                  : __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ks][j], af[ks][i], acc[ph * TMP + i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (TIMING) t3 = stamp();
            if (!(ph == PH - 1 && wr == 1 && t + 1 == nk) && !(TIMING && (p.flags & EP_DBG_NO_BAR_B))) {   // group 1's very last barrier would have no partner
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            if (TIMING) {
                const long long t4 = stamp();
                tm[ph * 4 + 0] += t1 - t0; tm[ph * 4 + 1] += t2 - t1; tm[ph * 4 + 2] += t3 - t2; tm[ph * 4 + 3] += t4 - t3;
                if (ph == 0) tm[16] += 1;
            }
        }
        tap1 = tap2; cb1 = cb2;
        advance(tap2, cb2);
    }
    if (TIMING) {
        if (lane == 0 && p.dbg) {
            long long* d = p.dbg + ((long)blockIdx.x * 8 + wave) * 17;
            for (int i = 0; i < 17; ++i) d[i] = tm[i];
        }
    }
    gemm_epilogue<TM, TN, WTM, WTN, GEGLU, TR, 2, BN, STATS, LNM>(p, acc, m0, n0, wr, wc, lane, z, smem);
}

// ---------------------------------------------------------------------------------------------------------------
// Row-shared ping-pong kernel for the stride-1 3x3 convolutions (GemmP::korder == 2, "dx" K order; round 4).
//
// The three taps of one kernel row (dy; dx = 0, 1, 2) read the SAME input rows shifted by one pixel.  When a tile consists of whole
// image rows (BM % W == 0: 4 rows of 64, 4 of 32, 8 of 16 ...) the shifted pixel of every GEMM row but the image-border ones is
// another row of the same LDS tile, so the activation tile is staged ONCE per (dy, 64-channel block) — at dx = 1, i.e. unshifted —
// and the dx = 0 / 2 K tiles read their MFMA fragments one LDS row up / down.  Every image row of the tile sits between two zeroed
// GUARD rows in LDS (pixel row r of the tile lives in LDS row r + r / W + 1; the guards are written once, the staging never touches
// them), so the lanes that read across the left / right image border (lane 0 of an x = 0 fragment, lane 15 of an x = W - 16
// fragment) get their zero padding from the same ds_read, without a select.  The XOR swizzle is a function of the PIXEL row, so the
// shifted reads stay conflict-free (16 consecutive rows whatever the offset) and the staging side is the tap-major kernel's.
// K order: dy outer, channel block, dx inner; weights stay [N][tap][Cin].  Against the tap-major kernel above this removes two
// thirds of the activation global_load_lds traffic (72 -> 50.7 KB per K tile on the 256x320 tile), two thirds of the A issue
// slots, and the gather rebuild runs 3 times per source instead of 9; the weight pipeline (B of tile T+2 refilled in L(1) / L(2) of
// tile T) is unchanged.  The fp32 summation order differs from the tap-major order: results agree to accumulation rounding, not
// bitwise (tests/test_gpu_ops.py).
//
// Schedule per group g of 3 K tiles (s = 0, 1, 2 = dx), loads per thread in brackets, A buffer g & 1, B buffer T & 1:
//     s = 0 : L(0) A units 0, 1 of group g+1 [2] | L(1) B half 0 of T+2 [NB0] | L(2) B half 1 of T+2 [NB1] | L(3) A units 2, 3 of g+1 [2]
//                                                                                                       | L(last) wait <= BU + 4
//     s = 1 : L(1), L(2) B of T+2                                                                       | L(last) wait <= BU + 2
//     s = 2 : L(1), L(2) B of T+2                                                                       | L(last) wait <= BU
// (PH = 2 tiles: one A unit per slot, B in L(1); waits BU + 2 / BU + 1 / BU.)  The wait of tile T leaves only loads younger than B of
// tile T+1 outstanding (in issue order the second A pair follows B of tile 3g+2, so the s = 1 wait may leave it in flight); the s = 2
// wait covers all of group g+1's activations, issued two K tiles earlier: the buffer they land in was read last by group g-1.  RAW / WAR as in
// the tap-major kernel: counted wait, then the barrier that closes the L section; a buffer is refilled only after the L section
// that read it last was closed by lgkmcnt(0) + barrier on both wave groups.
// ---------------------------------------------------------------------------------------------------------------
template <int BM, int BN, bool STATS = false>
__global__ __launch_bounds__(512) void gemm_mfma_pingpong_dx_kernel(GemmP p) {
    if (p.gate && *p.gate == 0) return;
    constexpr int WC = 4, ROWB = 128;
    constexpr int WTM = BM / 2, WTN = BN / WC, TM = WTM / 16, TN = WTN / 16;
    constexpr int RP = 32, TMP = 2;
    constexpr int PH = WTM / RP;
    static_assert(PH == 4 || PH == 2, "BM must be 256 or 128");
    constexpr int BU = BN / 64, NB0 = BU / 2, NB1 = BU - NB0;
    constexpr int NA = PH == 4 ? 4 : 2;                      // activation loads per thread and group
    constexpr int A_ROWS = BM + BM / 16 + 1;                 // pixel rows + one guard row per image row (W >= 16) + the leading guard
    constexpr int A_BYTES = A_ROWS * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    static_assert(BN % 64 == 0 && WTN % 16 == 0 && BU + NA < 64 && TM <= 8, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;

    const int tiles_n = p.N / BN;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, sub = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + sub;
    }
    int tile_m, tile_n;
    if (p.tile_order) {
        const int tiles_m = (p.M + BM - 1) / BM;
        tile_n = bid / tiles_m; tile_m = bid - tile_n * tiles_m;
    } else {
        tile_m = bid / tiles_n; tile_n = bid - tile_m * tiles_n;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const long z = blockIdx.z;
    const half_t* a0 = p.a0 + z * p.a_bs;
    const half_t* a1 = p.a1 ? p.a1 + z * p.a_bs : nullptr;
    const half_t* wbase = p.w + z * p.w_bs;

    // staging roles: as in gemm_mfma_pingpong_kernel
    const int ch = tid & 7;
    const int a_rin = (wave & 3) * 8 + (lane >> 3);
    const int b_row = tid >> 3;
    const int src_chunk = (ch ^ (lane >> 3)) * 8;
    int* rinfo = reinterpret_cast<int*>(smem + 2 * STAGE) + tid;        // [q * 512]
    const half_t* rptr[PH];
#pragma unroll
    for (int q = 0; q < PH; ++q) {
        const int m = m0 + wr * WTM + q * RP + a_rin;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int b = mm / p.rows_per_batch;
        const int rem = mm - b * p.rows_per_batch;
        const int yo = rem / p.Wo;
        const int xo = rem - yo * p.Wo;
        rinfo[q * 512] = (b << 24) | (((ok ? yo - 1 : -2048) & 0xfff) << 12) | (xo & 0xfff);     // window origin row, the pixel's own column
        rptr[q] = p.zero;
    }
    {   // guard rows of both activation buffers: LDS rows k * (W + 1), k = 0 .. BM / W
        const int ngr = BM / p.Wo + 1, k = tid >> 3;
        if (k < 2 * ngr) {
            const int sb = k >= ngr ? 1 : 0, kk = k - sb * ngr;
            *reinterpret_cast<uint4*>(smem + sb * STAGE + kk * (p.Wo + 1) * ROWB + (tid & 7) * 16) = uint4{0u, 0u, 0u, 0u};
        }
    }
    const int img_pix = p.Hi * p.Wi;
    const int chunk8 = ((tid & 7) ^ (lane >> 3)) * 8;
    // LDS row of the first of the 8 pixel rows this wave stages in unit q (8 aligned rows never straddle an image row), and the
    // guard rows in front of each of the wave's TM fragment tiles, 8 bits per tile
    int a_lds[PH];
    unsigned long long gpack = 0;
#pragma unroll
    for (int q = 0; q < PH; ++q) {
        const int r0 = wr * WTM + q * RP + (wave & 3) * 8;
        a_lds[q] = (r0 + r0 / p.Wo + 1) * ROWB;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) gpack |= (unsigned long long)((wr * WTM + i * 16) / p.Wo + 1) << (8 * i);
    const int b_lds = A_BYTES + wave * 8 * ROWB;
    const half_t* b_src = wbase + (long)(n0 + b_row) * p.ldw + src_chunk;
    const long b_piece = 64L * p.ldw;

    f4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    int nk = p.K / 64;
    int k_first = 0;
    if (p.splitk > 1) {
        k_first = blockIdx.y * p.splitk_steps;               // a multiple of 3 (launch_gemm)
        nk = min(nk - k_first, p.splitk_steps);
    }
    const int ng = nk / 3;
    // K position of the weight tile T+2 (dy2, cb2, dx2) and of the NEXT group's activations (dyA, cbA)
    int dy2, cb2, dx2 = 0, dyA, cbA;
    {
        const int cpb = p.cin >> 6, g0 = k_first / 3;
        dy2 = g0 / cpb;
        cb2 = (g0 - dy2 * cpb) << 6;
        dyA = dy2; cbA = cb2;
    }
    auto advance_b = [&]() {
        if (++dx2 == 3) { dx2 = 0; cb2 += 64; if (cb2 >= p.cin) { cb2 = 0; ++dy2; } }
    };
    auto advance_a = [&]() {
        cbA += 64;
        if (cbA >= p.cin) { cbA = 0; ++dyA; }
    };
    // A units [q0, q0+nq) of group (dy, cbase) into A buffer sb (the tap-major kernel's gather at dx = 1: the pixel's own column)
    auto issue_a = [&](int q0, int nq, int dy, int cbase, int sb, bool force) {
        char* base = smem + sb * STAGE;
        const bool first = cbase < p.c0;
        const int seg0 = first ? 0 : p.c0;
        if (force || cbase == seg0) {                        // block-uniform: the group opens a new (dy, source) segment
            const half_t* src = first ? a0 : a1;
            const int lda = first ? p.lda0 : p.lda1;
#pragma unroll
            for (int q = q0; q < q0 + nq; ++q) {
                const int ri = rinfo[q * 512];
                const int yr = ((int)((unsigned)ri << 8) >> 20) + dy, xr = ri & 0xfff;
                const bool ok = (unsigned)yr < (unsigned)p.Hi;            // rows past M carry yb = -2048
                const int pix = (ri >> 24) * img_pix + yr * p.Wi + xr;
                rptr[q] = (ok ? src + (long)pix * lda : p.zero) + chunk8;
            }
        }
        const int coff = cbase - seg0;
#pragma unroll
        for (int q = q0; q < q0 + nq; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(rptr[q] + coff), (lptr_t)(base + a_lds[q]), 16, 0, 0);
    };
    auto issue_b = [&](int i0, int n, int sb) {              // B pieces [i0, i0+n) of the K tile at (dy2, cb2, dx2)
        char* base = smem + sb * STAGE + b_lds;
        const half_t* bt = b_src + (long)(dy2 * 3 + dx2) * p.cin + cb2;
#pragma unroll
        for (int i = i0; i < i0 + n; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(bt + i * b_piece), (lptr_t)(base + i * 64 * ROWB), 16, 0, 0);
    };

    const int lr = lane & 15, lk = lane >> 4;
    const int sw0 = (lk ^ (lr & 7)) << 4, sw1 = ((4 + lk) ^ (lr & 7)) << 4;
    const int b_rd = A_BYTES + (wc * WTN + lr) * ROWB;
    h8 bf[2][TN], af[2][TMP];
    auto read_b = [&](const char* sbuf) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bf[0][j] = *reinterpret_cast<const h8*>(sbuf + b_rd + j * 16 * ROWB + sw0);
            bf[1][j] = *reinterpret_cast<const h8*>(sbuf + b_rd + j * 16 * ROWB + sw1);
        }
    };

    // ---- prologue: weights of tiles 0 and 1, activations of group 0 ---------------------------------------------------
    issue_b(0, NB0, 0);
    issue_b(NB0, NB1, 0);
    issue_a(0, PH / 2, dyA, cbA, 0, true);
    issue_a(PH / 2, PH / 2, dyA, cbA, 0, true);
    advance_b();
    advance_a();
    if (nk > 1) {
        issue_b(0, NB0, 1);
        issue_b(NB0, NB1, 1);
        wait_vmcnt<BU>();
    } else {
        wait_vmcnt<0>();
    }
    advance_b();                                             // (dy2, cb2, dx2) = tile 2
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the guard rows are written
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wr == 1) {                                           // group 1 runs one barrier behind group 0 from here on
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // dx (the tile's kernel column) and `more` (another group follows this one) are wave-uniform RUNTIME values: the three tiles of a
    // group differ in a few scalar branches only.  (Unrolled by 3 with dx a compile-time constant the loop body triples and the
    // register allocator spills ~200 VGPRs on the 256x320 tile.)
    int dx = 0, ga = 0;
    bool more = ng > 1;
#pragma nounroll
    for (int t = 0; t < nk; ++t) {
        asm volatile("" : "+s"(dx));                         // opaque: no unroll-by-3 / jump threading through the dx cycle (28 loop clones, spills)
        const int cur = t & 1;
        const char* sbufB = smem + cur * STAGE;
        const bool last_tile = t + 1 == nk;
        const bool issue_bt = dx == 0 || more;               // the weight tile t + 2 exists
        const bool issue_at = dx == 0 && more;               // this tile carries the next group's activation loads
        // lane part of the fragment addresses: pixel row (lr + dx - 1) of a 16-row tile, swizzled by the pixel row
        const int rs = lr + dx - 1;
        const int ao0 = (wr * WTM + rs) * ROWB + ((lk ^ (rs & 7)) << 4) + ga * STAGE;
        const int ao1 = (wr * WTM + rs) * ROWB + (((4 + lk) ^ (rs & 7)) << 4) + ga * STAGE;
#pragma unroll
        for (int ph = 0; ph < PH; ++ph) {
            // ================= L section =================
            if (ph == 0) read_b(sbufB);
#pragma unroll
            for (int i = 0; i < TMP; ++i) {
                const int ti = ph * TMP + i;
                const int gu = (int)((gpack >> (8 * ti)) & 0xffu) * ROWB;       // wave-uniform: the guard rows in front of this tile
                af[0][i] = *reinterpret_cast<const h8*>(smem + (gu + ao0) + ti * 16 * ROWB);
                af[1][i] = *reinterpret_cast<const h8*>(smem + (gu + ao1) + ti * 16 * ROWB);
            }
            if (PH == 4) {
                if (ph == 0) { if (issue_at) issue_a(0, 2, dyA, cbA, ga ^ 1, false); }
                else if (ph == 1) { if (issue_bt) issue_b(0, NB0, cur); }
                else if (ph == 2) { if (issue_bt) issue_b(NB0, NB1, cur); }
                else { if (issue_at) issue_a(2, 2, dyA, cbA, ga ^ 1, false); }
            } else {
                if (ph == 0) { if (issue_at) issue_a(0, 1, dyA, cbA, ga ^ 1, false); }
                else {
                    if (issue_bt) issue_b(0, BU, cur);
                    if (issue_at) issue_a(1, 1, dyA, cbA, ga ^ 1, false);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ph == PH - 1) {                              // the weight tile t + 1 has landed; what may still fly is younger (header)
                if (issue_at) wait_vmcnt<BU + NA>();
                else if (dx == 1 && more) wait_vmcnt<BU + NA / 2>();
                else if (issue_bt) wait_vmcnt<BU>();
                else wait_vmcnt<0>();
            }
            // (Tried, round 4: passing this barrier of phases >= 1 — which read activations only, a buffer not rewritten before the group
            // after next — BEFORE the fragment reads returned, the 8-phase template's order: 434.5 vs 434.2 ms on the C1 job, no gain, and
            // the flag cost the 256x320 instantiation 6 SGPR spills.  Removed.)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ================= M section =================
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < TMP; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[ph * TMP + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ks][j], af[ks][i], acc[ph * TMP + i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(ph == PH - 1 && wr == 1 && last_tile)) {   // group 1's very last barrier would have no partner
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        advance_b();
        if (++dx == 3) {
            dx = 0;
            advance_a();
            ga ^= 1;
            more = t + 4 < nk;                               // the group that starts at tile t + 1 has a successor
        }
    }
    gemm_epilogue<TM, TN, WTM, WTN, false, false, 2, BN, STATS, 0>(p, acc, m0, n0, wr, wc, lane, z, smem);
}

// ---------------------------------------------------------------------------------------------------------------
// Generic kernel: one thread per output element, same addressing rules, no shape restrictions beyond Cin % 8 == 0.
// Used for shapes the MFMA kernel does not cover and as the independent HIP cross-check in the parity tests.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot_row(const GemmP& p, const half_t* a0, const half_t* a1, const half_t* wrow,
                                         const RowInfo& ri) {
    float acc = 0.f;
    for (int tap = 0; tap < p.taps; ++tap) {
        for (int cb = 0; cb < p.cin; cb += 8) {
            // find the source of this 8-channel group
            const half_t* src;
            int cch, lda;
            if (cb < p.c0) { src = a0; cch = cb; lda = p.lda0; } else { src = a1; cch = cb - p.c0; lda = p.lda1; }
            int dy = 0, dx = 0;
            if (p.taps == 9) { dy = tap / 3; dx = tap - dy * 3; }
            int yi, xi;
            bool ok = true;
            const bool wrap = p.flags & EP_WRAP;
            if (p.up) {
                int yy = ri.yo + dy - 1, xx = ri.xo + dx - 1;
                if (wrap) { yy = (yy + 2 * p.Hi) % (2 * p.Hi); xx = (xx + 2 * p.Wi) % (2 * p.Wi); }
                ok = yy >= 0 && xx >= 0 && yy < 2 * p.Hi && xx < 2 * p.Wi;
                yi = yy >> 1; xi = xx >> 1;
            } else {
                yi = ri.yo * p.stride + dy - p.pad;
                xi = ri.xo * p.stride + dx - p.pad;
                if (wrap) { yi = (yi + p.Hi) % p.Hi; xi = (xi + p.Wi) % p.Wi; }
                ok = yi >= 0 && xi >= 0 && yi < p.Hi && xi < p.Wi;
            }
            if (!ok) continue;
            const long pix = ((long)ri.b * p.Hi + yi) * p.Wi + xi;
            const h8 av = *reinterpret_cast<const h8*>(src + pix * lda + cch);
            const h8 wv = *reinterpret_cast<const h8*>(wrow + (long)tap * p.cin + cb);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf((float)av[e], (float)wv[e], acc);
        }
    }
    return acc;
}

__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmP p) {
    if (p.gate && *p.gate == 0) return;
    const long z = blockIdx.z;
    const half_t* a0 = p.a0 + z * p.a_bs;
    const half_t* a1 = p.a1 ? p.a1 + z * p.a_bs : nullptr;
    const half_t* wbase = p.w + z * p.w_bs;
    const bool geglu = p.flags & EP_GEGLU;
    const int nout_cols = geglu ? p.N / 2 : ((p.flags & EP_NCHW) ? p.n_real : p.N);
    const long total = (long)p.M * nout_cols;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / nout_cols), no = (int)(idx - (long)m * nout_cols);
        RowInfo ri;
        ri.ok = true;
        ri.b = m / p.rows_per_batch;
        const int rem = m - ri.b * p.rows_per_batch;
        ri.yo = rem / p.Wo;
        ri.xo = rem - ri.yo * p.Wo;
        if (geglu) {
            const int g = no >> 5, r = no & 31;
            const int na = g * 64 + r, ng = na + 32;
            float a = dot_row(p, a0, a1, wbase + (long)na * p.ldw, ri) * p.alpha;
            float gt = dot_row(p, a0, a1, wbase + (long)ng * p.ldw, ri) * p.alpha;
            if (p.bias) { a += p.bias[na]; gt += p.bias[ng]; }
            ((half_t*)p.out)[z * p.o_bs + (long)m * p.ldo + no] = (half_t)(SDMI_GELU_SIG ? geglu_gate(a, gt) : a * gelu_erf(gt));
            continue;
        }
        float v = no < p.n_valid ? dot_row(p, a0, a1, wbase + (long)no * p.ldw, ri) * p.alpha : 0.f;
        if (p.bias) v += (p.flags & EP_BIAS_ROW) ? p.bias[m] : p.bias[no] * p.bias_scale;
        if (p.rowbias) v += p.rowbias[(long)ri.b * p.ldrb + no];
        if (p.flags & EP_QUICK_GELU) v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v));
        if (p.flags & EP_GELU) v = gelu_erf(v);
        if (p.resid) v += (float)p.resid[z * p.r_bs + (long)m * p.ldr + no];
        if (p.flags & EP_TRANSPOSE) {
            ((half_t*)p.out)[z * p.o_bs + ((long)ri.b * p.N + no) * p.ldo + rem] = (half_t)v;
        } else if (p.flags & EP_NCHW) {
            ((float*)p.out)[z * p.o_bs + ((long)ri.b * p.n_real + no) * p.rows_per_batch + rem] = v;
        } else if (p.flags & EP_OUT_F32) {
            ((float*)p.out)[z * p.o_bs + (long)m * p.ldo + no] = v;
        } else {
            ((half_t*)p.out)[z * p.o_bs + (long)m * p.ldo + no] = (half_t)v;
        }
    }
}

// Split-K second pass: out = epilogue(sum over slices, in slice order => bit-reproducible).  One thread per 4 columns.
template <bool LNF = false>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmP p, int batch) {
    if (p.gate && *p.gate == 0) return;
    // (row, column-quad) pairs walked by the grid stride's quotient / remainder: no 64-bit division per element (static ISA review,
    // docs/DESIGN_experiments.md)
    const int NQ = p.N / 4;
    const int stride = (int)gridDim.x * 256;                 // M * N / 4 < 2^31: split-K is admitted for the small-M layers only
    const int sp = stride / NQ, sr = stride - sp * NQ;
    const int i_init = (int)blockIdx.x * 256 + (int)threadIdx.x;
    for (int z = 0; z < batch; ++z) {
        int m = i_init / NQ, nq = i_init - m * NQ;
        for (; m < p.M; m += sp, nq += sr) {
            if (nq >= NQ) { nq -= NQ; if (++m >= p.M) break; }
            const int n = nq * 4;
            f4 v = {0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < p.splitk; ++s)
                v += *reinterpret_cast<const f4*>(p.splitk_ws + ((long)s * batch + z) * (long)p.M * p.N + (long)m * p.N + n);
            const int b = m / p.rows_per_batch;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= p.alpha;
            if constexpr (LNF) {
                float mean, rstd;
                if (p.ln_np == 0) { mean = p.ln_stats[2 * m]; rstd = p.ln_stats[2 * m + 1]; }
                else {
                    const float* pp = p.ln_stats + (long)m * p.ln_np * 2;
                    float sm = 0.f, q = 0.f;
                    for (int t = 0; t < p.ln_np; ++t) { sm += pp[2 * t]; q += pp[2 * t + 1]; }
                    mean = sm * p.ln_inv_c;
                    rstd = rsqrtf(fmaxf(fmaf(-mean, mean, q * p.ln_inv_c), 0.f) + p.ln_eps);
                }
                const f4 sv = *reinterpret_cast<const f4*>(p.ln_s + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = rstd * (v[r] - mean * sv[r]);
            }
            if (p.bias) {
                if (p.flags & EP_BIAS_ROW) {
                    const float bb = p.bias[m];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += bb;
                } else {
                    const f4 bb = *reinterpret_cast<const f4*>(p.bias + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(bb[r], p.bias_scale, v[r]);
                }
            }
            if (p.rowbias) v += *reinterpret_cast<const f4*>(p.rowbias + (long)b * p.ldrb + n);
            if (p.flags & (EP_QUICK_GELU | EP_GELU)) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[r] = (p.flags & EP_QUICK_GELU) ? v[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v[r])) : gelu_erf(v[r]);
            }
            if (p.resid) {
                const h4 rr = *reinterpret_cast<const h4*>(p.resid + z * p.r_bs + (long)m * p.ldr + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
                if constexpr (!LNF) {
                    if (const half_t* rlo = gemm_resid_lo(p)) {      // (hi, lo) residual (EP_HILO)
                        const h4 rl = *reinterpret_cast<const h4*>(rlo + z * p.r_bs + (long)m * p.ldr + n);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)rl[r];
                    }
                }
            }
            if (p.flags & EP_OUT_F32) {
                *reinterpret_cast<f4*>((float*)p.out + z * p.o_bs + (long)m * p.ldo + n) = v;
            } else {
                h4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
                *reinterpret_cast<h4*>((half_t*)p.out + z * p.o_bs + (long)m * p.ldo + n) = o;
                if constexpr (!LNF) {
                    if (half_t* olo = gemm_out_lo(p)) {              // what the fp16 rounding dropped (EP_HILO)
                        h4 l;
#pragma unroll
                        for (int r = 0; r < 4; ++r) l[r] = (half_t)(v[r] - (float)o[r]);
                        *reinterpret_cast<h4*>(olo + z * p.o_bs + (long)m * p.ldo + n) = l;
                    }
                }
            }
        }
    }
}

template <int BM, int BN, int WR, int WC, int BK, bool GLDS, bool GEGLU, bool TR = false, bool KORD = false, bool STATS = false, int LNM = 0, int NS = 2,
          bool LIN = false, bool LIN3 = false>
static int launch_cfg2(const GemmP& p, int batch, hipStream_t s) {
    constexpr int SMEM = NS * (BM + BN) * BK * 2;
    constexpr int NT = WR * WC * 64;
    auto kern = gemm_mfma_kernel<BM, BN, WR, WC, BK, GLDS, GEGLU, TR, KORD, STATS, LNM, NS, LIN, LIN3>;
    static PerDeviceOnce attr;
    if (attr.need()) SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    const int tiles = cdiv(p.M, BM) * (p.N / BN);
    hipLaunchKernelGGL(kern, dim3(tiles, p.splitk > 1 ? p.splitk : 1, batch), dim3(NT), SMEM, s, p);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int BM, int BN, bool GEGLU, bool TR = false, bool KORD = false, bool STATS = false, int LNM = 0>
static int launch_pingpong2(const GemmP& p, int batch, hipStream_t s) {
    constexpr int SMEM = 2 * (BM + BN) * 128 + 8192;      // tile buffers + packed gather words
    auto kern = gemm_mfma_pingpong_kernel<BM, BN, GEGLU, false, TR, KORD, STATS, LNM>;
    static PerDeviceOnce attr;
    if (attr.need()) SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    const int tiles = cdiv(p.M, BM) * (p.N / BN);
    hipLaunchKernelGGL(kern, dim3(tiles, p.splitk > 1 ? p.splitk : 1, batch), dim3(512), SMEM, s, p);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int BM, int BN>
static int launch_pingpong_dx(const GemmP& p, int batch, hipStream_t s) {
    constexpr int SMEM = 2 * (BM + BM / 16 + 1 + BN) * 128 + 8192;  // tile buffers (activations with guard rows) + packed gather words
    const bool stats = p.stats_nchunk > 0;
    auto kern = stats ? gemm_mfma_pingpong_dx_kernel<BM, BN, true> : gemm_mfma_pingpong_dx_kernel<BM, BN, false>;
    static PerDeviceOnce attr[2];
    if (attr[stats].need()) SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    const int tiles = cdiv(p.M, BM) * (p.N / BN);
    hipLaunchKernelGGL(kern, dim3(tiles, p.splitk > 1 ? p.splitk : 1, batch), dim3(512), SMEM, s, p);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}
unsigned long long g_gemm_dbg = 0;
int g_gemm_dbgflags = 0;

template <int BM, int BN>
static int launch_pingpong(const GemmP& p, int batch, hipStream_t s) {
    if (p.korder == 2) return launch_pingpong_dx<BM, BN>(p, batch, s);     // row-shared 3x3 walk (launch_gemm admitted the shape)
    // tuning: instrumented instantiation — never for a launch that promised GroupNorm / LayerNorm partial sums to its consumer (the
    // instrumented kernel has no such epilogue: the consumer would normalise with sums that were never written)
    if (g_gemm_dbg && !(p.flags & EP_GEGLU) && p.stats_nchunk == 0 && p.lnp_np == 0 && !(p.flags & EP_LNFOLD)) {
        GemmP q = p;
        q.dbg = (long long*)g_gemm_dbg;
        constexpr int SMEM = 2 * (BM + BN) * 128 + 8192;
        auto kern = (p.korder && p.taps == 9) ? gemm_mfma_pingpong_kernel<BM, BN, false, true, false, true>      // production K order of the 3x3 convs
                                              : gemm_mfma_pingpong_kernel<BM, BN, false, true>;
        SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        hipLaunchKernelGGL(kern, dim3(cdiv(q.M, BM) * (q.N / BN), q.splitk > 1 ? q.splitk : 1, batch), dim3(512), SMEM, s, q);
        SDMI_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (p.flags & EP_LNFOLD) {                             // LayerNorm folded into this GEMM
        if constexpr ((BN / 4) % 64 == 0) {
            if (p.flags & EP_GEGLU) return launch_pingpong2<BM, BN, true, false, false, false, 1>(p, batch, s);
        }
        if (p.flags & EP_TRANSPOSE) return launch_pingpong2<BM, BN, false, true, false, false, 1>(p, batch, s);
        return launch_pingpong2<BM, BN, false, false, false, false, 1>(p, batch, s);
    }
    if (p.lnp_np > 0) return launch_pingpong2<BM, BN, false, false, false, false, 2>(p, batch, s);
    if constexpr ((BN / 4) % 64 == 0) {
        if (p.flags & EP_GEGLU) return launch_pingpong2<BM, BN, true>(p, batch, s);
    }
    if (p.flags & EP_TRANSPOSE) return launch_pingpong2<BM, BN, false, true>(p, batch, s);
    if (p.stats_nchunk > 0)
        return (p.korder && p.taps == 9) ? launch_pingpong2<BM, BN, false, false, true, true>(p, batch, s)
                                         : launch_pingpong2<BM, BN, false, false, false, true>(p, batch, s);
    if (p.korder && p.taps == 9) return launch_pingpong2<BM, BN, false, false, true>(p, batch, s);
    return launch_pingpong2<BM, BN, false>(p, batch, s);
}

// 1 (default): 1x1 / linear launches on the LDS-direct 4-wave tiles take the running-pointer K walk (gemm_mfma_kernel LIN).  0: the
// general gather everywhere (rounds 1-4) — for same-box A/Bs (debug knob "gemm_lin").  Same loads, same accumulation order: same bits.
int g_gemm_lin = [] { const char* e = getenv("SDMI_GEMM_LIN"); return e ? atoi(e) : 1; }();
static bool gemm_is_linear(const GemmP& p) {
    return g_gemm_lin && p.taps == 1 && p.stride == 1 && p.pad == 0 && !p.up && !(p.flags & EP_WRAP) && !p.korder &&
           p.Wo == p.Wi && p.rows_per_batch == p.Hi * p.Wi;                              // row m of the GEMM is pixel m of either source
}

static bool gemm_is_plain3x3(const GemmP& p) {
    return g_gemm_lin && p.taps == 9 && p.stride == 1 && p.pad == 1 && !p.up && !(p.flags & EP_WRAP) && !p.korder &&
           p.Wo == p.Wi && p.Ho == p.Hi && p.rows_per_batch == p.Hi * p.Wi;
}

template <int BM, int BN, int WR, int WC, int BK, bool GLDS>
static int launch_cfg(const GemmP& p, int batch, hipStream_t s) {
    if constexpr (GLDS && BM < 256) {                      // lean 3x3 walk: plain / GroupNorm-statistics epilogues (the VAE's 128-channel convs, conv_in / conv_out)
        if (gemm_is_plain3x3(p) && !(p.flags & (EP_LNFOLD | EP_GEGLU | EP_TRANSPOSE)) && p.lnp_np == 0) {
            if (p.stats_nchunk > 0) return launch_cfg2<BM, BN, WR, WC, BK, true, false, false, false, true, 0, 2, false, true>(p, batch, s);
            return launch_cfg2<BM, BN, WR, WC, BK, true, false, false, false, false, 0, 2, false, true>(p, batch, s);
        }
    }
    if constexpr (GLDS && BM < 256) {                      // linear walk: the forms the 1x1 layers of the UNet / VAE / CLIP run in
        if (gemm_is_linear(p) && !(p.flags & EP_LNFOLD)) {
            if (p.lnp_np > 0) return launch_cfg2<BM, BN, WR, WC, BK, true, false, false, false, false, 2, 2, true>(p, batch, s);
            if constexpr ((BN / WC) % 64 == 0) {
                if (p.flags & EP_GEGLU) return launch_cfg2<BM, BN, WR, WC, BK, true, true, false, false, false, 0, 2, true>(p, batch, s);
            }
            if (p.flags & EP_TRANSPOSE) return launch_cfg2<BM, BN, WR, WC, BK, true, false, true, false, false, 0, 2, true>(p, batch, s);
            if (p.stats_nchunk > 0) return launch_cfg2<BM, BN, WR, WC, BK, true, false, false, false, true, 0, 2, true>(p, batch, s);
            return launch_cfg2<BM, BN, WR, WC, BK, true, false, false, false, false, 0, 2, true>(p, batch, s);
        }
    }
    if constexpr (BM >= 256) {                             // the two-stage 256-row tiles (the ping-pong kernel's fallback) carry no LayerNorm forms
        if (p.flags & EP_LNFOLD) { set_error("EP_LNFOLD on a 256-row tile needs the ping-pong kernel (gemm_pipe 3 / 4)"); return 1; }
    }
    if constexpr (GLDS && BM < 256) {                      // LayerNorm folded into this GEMM (LDS-direct path only; launch_gemm checks)
        if (p.flags & EP_LNFOLD) {
            if constexpr ((BN / WC) % 64 == 0) {
                if (p.flags & EP_GEGLU) return launch_cfg2<BM, BN, WR, WC, BK, GLDS, true, false, false, false, 1>(p, batch, s);
            }
            if (p.flags & EP_TRANSPOSE) return launch_cfg2<BM, BN, WR, WC, BK, GLDS, false, true, false, false, 1>(p, batch, s);
            return launch_cfg2<BM, BN, WR, WC, BK, GLDS, false, false, false, false, 1>(p, batch, s);
        }
        if (p.lnp_np > 0) return launch_cfg2<BM, BN, WR, WC, BK, GLDS, false, false, false, false, 2>(p, batch, s);
    }
    if constexpr ((BN / WC) % 64 == 0) {
        if (p.flags & EP_GEGLU) return launch_cfg2<BM, BN, WR, WC, BK, GLDS, true>(p, batch, s);
    }
    if (p.flags & EP_TRANSPOSE) return launch_cfg2<BM, BN, WR, WC, BK, GLDS, false, true>(p, batch, s);
    if constexpr (GLDS) {                                  // GroupNorm statistics epilogue (LDS-direct path only)
        if (p.stats_nchunk > 0)
            return (p.korder && p.taps == 9) ? launch_cfg2<BM, BN, WR, WC, BK, GLDS, false, false, true, true>(p, batch, s)
                                             : launch_cfg2<BM, BN, WR, WC, BK, GLDS, false, false, false, true>(p, batch, s);
    }
    if (p.korder && p.taps == 9) return launch_cfg2<BM, BN, WR, WC, BK, GLDS, false, false, true>(p, batch, s);
    return launch_cfg2<BM, BN, WR, WC, BK, GLDS, false>(p, batch, s);
}

// ring-buffered instantiations (NS stages): plain / GEGLU / transposed / GroupNorm-statistics epilogues, tap-major K walk
template <int BM, int BN, int WR, int WC, int NS>
static int launch_ring(const GemmP& p, int batch, hipStream_t s) {
    if (gemm_is_linear(p)) {
        if (p.lnp_np > 0) return launch_cfg2<BM, BN, WR, WC, 64, true, false, false, false, false, 2, NS, true>(p, batch, s);
        if constexpr ((BN / WC) % 64 == 0) {
            if (p.flags & EP_GEGLU) return launch_cfg2<BM, BN, WR, WC, 64, true, true, false, false, false, 0, NS, true>(p, batch, s);
        }
        if (p.flags & EP_TRANSPOSE) return launch_cfg2<BM, BN, WR, WC, 64, true, false, true, false, false, 0, NS, true>(p, batch, s);
        if (p.stats_nchunk > 0) return launch_cfg2<BM, BN, WR, WC, 64, true, false, false, false, true, 0, NS, true>(p, batch, s);
        return launch_cfg2<BM, BN, WR, WC, 64, true, false, false, false, false, 0, NS, true>(p, batch, s);
    }
    if (p.lnp_np > 0) return launch_cfg2<BM, BN, WR, WC, 64, true, false, false, false, false, 2, NS>(p, batch, s);
    if constexpr ((BN / WC) % 64 == 0) {
        if (p.flags & EP_GEGLU) return launch_cfg2<BM, BN, WR, WC, 64, true, true, false, false, false, 0, NS>(p, batch, s);
    }
    if (p.flags & EP_TRANSPOSE) return launch_cfg2<BM, BN, WR, WC, 64, true, false, true, false, false, 0, NS>(p, batch, s);
    if (p.stats_nchunk > 0) return launch_cfg2<BM, BN, WR, WC, 64, true, false, false, false, true, 0, NS>(p, batch, s);
    return launch_cfg2<BM, BN, WR, WC, 64, true, false, false, false, false, 0, NS>(p, batch, s);
}

bool gemm_mfma_supported(const GemmP& p) {
    if (p.N % 64 || p.cin % 64 || p.K % 64) return false;
    if (p.c1 > 0 && (p.c0 % 64)) return false;
    if ((p.flags & EP_GEGLU) && (p.N % 64)) return false;
    if (p.c0 % 8 || p.c1 % 8 || p.lda0 % 8 || (p.c1 > 0 && p.lda1 % 8) || p.ldw % 8) return false;
    return true;
}

// Tile configurations.  flop per byte staged into LDS = BM*BN/(BM+BN): the big 8-wave tiles exist because the
// 128x128 tile (64 flop/B) saturates the L2 -> LDS path long before the MFMA pipe.
enum GemmCfg {
    CFG_128x128 = 0,      // 4 waves, BK 64, 64 KB LDS  (2 blocks/CU)
    CFG_256x64 = 1,       // 4 waves, BK 64, 80 KB
    CFG_64x64 = 2,        // 4 waves, BK 64, 32 KB      (deep, small-M levels)
    CFG_128x128_K32 = 3,  // 4 waves, BK 32, 32 KB      (3+ blocks/CU)
    CFG_256x256 = 4,      // 8 waves, BK 64, 128 KB
    CFG_256x320 = 5,      // 8 waves, BK 64, 144 KB     (N = 320 / 640 / 1280 / 2560)
    CFG_256x128 = 6,      // 8 waves, BK 64, 96 KB
    CFG_128x64 = 7,       // 4 waves, BK 64, 48 KB
    CFG_128x320 = 8,      // 8 waves, BK 64, 112 KB     (mid levels: twice the tiles of 256x320 at 91 flop/B)
    CFG_128x160 = 9,      // 4 waves, BK 64, 72 KB      (2 blocks/CU; the HBM-bound short-K 1x1 layers with N = 320 / 640 / 1280:
                          //                             one block's epilogue / prologue overlaps the other's K loop)
    // ring-buffered forms of the 4-wave tiles (round 4; gemm_mfma_kernel NS > 2): the short-K / small-M 1x1 layers, whose two-stage K
    // loop is a chain of load latencies.  Never picked by the score model: per shape through gemm_tuned_shapes.inc / gemm_override.
    CFG_128x160_R4 = 10,  // 4 stages, 144 KB (1 block/CU)
    CFG_128x128_R4 = 11,  // 4 stages, 128 KB (1 block/CU)
    CFG_128x64_R3 = 12,   // 3 stages, 72 KB  (2 blocks/CU)
    CFG_128x160_R3 = 13,  // 3 stages, 108 KB (1 block/CU)
    CFG_COUNT = 14
};
static const int kCfgBM[CFG_COUNT] = {128, 256, 64, 128, 256, 256, 256, 128, 128, 128, 128, 128, 128, 128};
static const int kCfgBN[CFG_COUNT] = {128, 64, 64, 128, 256, 320, 128, 64, 320, 160, 160, 128, 64, 160};
static const char* kCfgName[CFG_COUNT] = {"gemm_mfma_128x128", "gemm_mfma_256x64", "gemm_mfma_64x64", "gemm_mfma_128x128k32",
                                          "gemm_mfma_256x256", "gemm_mfma_256x320", "gemm_mfma_256x128", "gemm_mfma_128x64",
                                          "gemm_mfma_128x320", "gemm_mfma_128x160", "gemm_mfma_128x160r4", "gemm_mfma_128x128r4",
                                          "gemm_mfma_128x64r3", "gemm_mfma_128x160r3"};
// relative MFMA efficiency of each tile once the chip is full (measured, profiles/): used only to rank candidates
static const float kCfgEff[CFG_COUNT] = {0.66f, 0.55f, 0.45f, 0.62f, 1.0f, 1.0f, 0.60f, 0.64f, 0.95f, 0.70f, 0.70f, 0.66f, 0.64f, 0.70f};

int g_force_gemm_cfg = [] { const char* e = getenv("SDMI_GEMM_CFG"); return e ? atoi(e) : -1; }();

int g_force_gemm_split = 0;
// Experiment knob (default off): tile config forced for the short-K launches only (K / 64 < 8: the q / k / v / out projections and
// proj_in / proj_out, ~5 % of the C1 job at 2-3x their HBM floor with one 256x320 tile per CU) — e.g. SDMI_SHORTK_CFG=<128x64 id>
// to try several co-resident workgroups per CU in a same-box A/B (docs/DESIGN_experiments.md, lead 2).
int g_shortk_gemm_cfg = [] { const char* e = getenv("SDMI_SHORTK_CFG"); return e ? atoi(e) : -1; }();
int g_shortk_max_k = [] { const char* e = getenv("SDMI_SHORTK_MAXK"); return e ? atoi(e) : 448; }();
int g_geglu_gemm_cfg = [] { const char* e = getenv("SDMI_GEGLU_CFG"); return e ? atoi(e) : -1; }();
// K walk of the 3x3 convolutions.  2 (default since round 4): row-shared walk (gemm_mfma_pingpong_dx_kernel: dy outer, channel block,
// dx inner; the activation tile staged once per three K tiles) wherever launch_gemm admits it — stride 1, ping-pong tile made of whole
// image rows, W % 16 == 0 — and tap-major elsewhere; same-box A/B on the C1 job 407.9 -> 405.1 ms, 3x3 class 150.1 -> 147.7 ms, every
// admitted shape faster or equal (profiles/r04_knob_dx.json).  0: tap-major everywhere (rounds 1-3).  1: channel-block-major
// (-31 % HBM traffic on the 3x3 convs, PMC profiles/r02_pmc_traffic.md, but slower: same-box two-build A/B profiles/r02_conv_korder.md —
// tap-major 161.4 ms of 3x3 convs per job, channel-block-major 165.5 ms with the per-source pointer + tap-mask addressing, 188.3 ms
// with the per-tile address rebuild it shipped with first).
int g_conv_korder_default = [] { const char* e = getenv("SDMI_CONV_KORDER"); return e ? atoi(e) : 2; }();
int g_conv_korder = g_conv_korder_default;
int g_ep_wide = [] { const char* e = getenv("SDMI_EP_WIDE"); return e ? atoi(e) : 1; }();
int g_gn_fuse = [] { const char* e = getenv("SDMI_GN_FUSE"); return e ? atoi(e) : 1; }();
int g_tile_order = [] { const char* e = getenv("SDMI_TILE_ORDER"); return e ? atoi(e) : -1; }();
int g_vt_mode = [] { const char* e = getenv("SDMI_VT_MODE"); return e ? atoi(e) : 1; }();
// 4 (default): ping-pong kernel for the 256-row tiles and the 128x320 tile (+5..22 % over the two-stage kernel per shape,
// bit-identical results), two-stage kernels elsewhere.  3: ping-pong for the 256-row tiles only.  0: two-stage kernels only.
// (A BK = 32 ring-buffer kernel was measured 7-25 % SLOWER than the two-stage kernels in round 1 — profiles/r01_microbench_pipe.txt —
// and has been removed.)
int g_gemm_pipe_default = [] { const char* e = getenv("SDMI_GEMM_PIPE"); return e ? atoi(e) : 4; }();
int g_gemm_pipe = g_gemm_pipe_default;

static bool cfg_valid(int cfg, const GemmP& p) {
    if (cfg < 0 || cfg >= CFG_COUNT) return false;
    if (p.N % kCfgBN[cfg]) return false;
    if (cfg == CFG_256x64) return false;                               // measured never best: not instantiated
    // 256x128: not in pick_cfg's candidate list (the two-stage form was never best in round 1); instantiated for the ping-pong kernel
    // so that the shape tuner / gemm_cfg=6 can try it on the N = 128 VAE convs (now 128x128 two-stage at ~750 TFLOP/s)
    if ((p.flags & EP_GEGLU) && (cfg == CFG_256x320 || cfg == CFG_128x320 || cfg == CFG_128x160 || cfg == CFG_256x128 || cfg == CFG_128x160_R4 || cfg == CFG_128x160_R3)) return false;   // wave tile not a multiple of 64
    if (cfg >= CFG_128x160_R4 && (p.korder || (p.flags & EP_LNFOLD))) return false;     // ring forms: tap-major, no folded LayerNorm
    return true;
}

// workgroups of each config that fit on one CU (LDS-limited)
static const int kCfgOcc[CFG_COUNT] = {2, 2, 4, 3, 1, 1, 1, 3, 1, 2, 1, 1, 2, 1};

// Expected relative throughput of (cfg, split): tile efficiency x chip fill / split-K overhead.  Fitted to the sweeps in
// profiles/ (tools/bench_kernels.py): the first workgroup per CU brings ~80 % of a config's rate, co-resident ones the
// rest; beyond one full residency the tail wave quantises; split-K slabs that fit the 256 MB Infinity Cache are cheap.
static float cfg_score(const GemmP& p, int batch, int cfg, int split) {
    const float tiles = (float)((long)cdiv(p.M, kCfgBM[cfg]) * (p.N / kCfgBN[cfg]) * batch * split);
    const float cus = 256.f, cap = cus * kCfgOcc[cfg];
    float fill;
    if (tiles <= cus) fill = 0.8f * tiles / cus;
    else if (tiles <= cap) fill = 0.8f + 0.2f * (tiles - cus) / (cap - cus + 1e-3f);
    else { const float w = tiles / cap; fill = w / ceilf(w); }
    if (kCfgOcc[cfg] == 1 && tiles <= cus) fill = tiles / cus;
    float score = kCfgEff[cfg] * fill;
    // short K loops are dominated by prologue / epilogue: favour the higher-occupancy 4-wave tiles there
    const int ksteps = p.K / 64 / split;
    if (ksteps < 8 && kCfgOcc[cfg] == 1) score *= 0.9f;
    // (Tried: preferring 128x320 over 256x320 for the K <= 384 projections — +13 % in the isolated sweep of
    // tools/geglu_sweep.py, but -1.5 % on the whole job in a same-box A/B: in the engine the inputs are cache-resident.)
    if (split > 1) {
        const double flops = 2.0 * p.M * (double)p.N * p.K * batch;
        const double t_mma = flops / (1.0e15 * (score > 0.05f ? score : 0.05f));
        const double slab_bytes = (double)split * p.M * p.N * batch * 4.0;
        const double bw = slab_bytes < 128.0e6 ? 6.0e12 : 3.0e12;
        const double t_red = 2.0 * slab_bytes / bw + 3.0e-6;
        score = (float)(score * t_mma / (t_mma + t_red));
    }
    return score;
}

// Per-shape choices.  (1) a runtime override list for in-engine autotuning (sdmi_debug_set_str("gemm_override", ...),
// tools/gpu/shape_tune.py: every candidate tile / split of a shape is timed INSIDE the whole job, where its operands sit in the
// cache state they really have — isolated micro-benchmarks mis-ranked tiles twice in round 1); (2) the table those runs produced
// for the shapes of the benchmarked workload.  Anything not listed falls through to the score model below.
struct ShapeKey { int M, N, K, taps, kind; };            // kind: 0 plain epilogue, 1 GEGLU, 2 transposed output; + 4: the (hi, lo) launch of that kind
struct ShapeChoice { ShapeKey k; int cfg, split; };
static std::vector<ShapeChoice> g_gemm_override;
static const ShapeChoice kTunedShapes[] = {
    // {M, N, K, taps, kind}, cfg, split      — filled from profiles/r02_shape_tuning.md
#include "gemm_tuned_shapes.inc"
    {{0, 0, 0, 0, 0}, -1, 1}
};
int gemm_set_override(const char* spec) {
    g_gemm_override.clear();
    if (!spec) return 0;
    const char* c = spec;
    while (*c) {
        ShapeChoice sc{};
        int n = 0;
        if (sscanf(c, "%d,%d,%d,%d,%d:%d:%d%n", &sc.k.M, &sc.k.N, &sc.k.K, &sc.k.taps, &sc.k.kind, &sc.cfg, &sc.split, &n) != 7) return 1;
        if (sc.split > 8) sc.split = 8;                      // gemm_splitk_ws_bytes sizes the slab workspace for 8 slices
        g_gemm_override.push_back(sc);
        c += n;
        if (*c == ';') ++c;
    }
    return 0;
}
static const ShapeChoice* find_shape(const GemmP& p) {
    // (hi, lo) launches of the accuracy mode (EP_HILO: twice the epilogue bytes) may carry their own entries, kind + 4; without one
    // they take the entry of the plain launch of the same shape
    const int base_kind = (p.flags & EP_GEGLU) ? 1 : (p.flags & EP_TRANSPOSE) ? 2 : 0;
    for (int kind : {(p.flags & EP_HILO) ? base_kind + 4 : base_kind, base_kind}) {
        for (const ShapeChoice& sc : g_gemm_override)
            if (sc.k.M == p.M && sc.k.N == p.N && sc.k.K == p.K && sc.k.taps == p.taps && sc.k.kind == kind) return &sc;
        for (const ShapeChoice& sc : kTunedShapes)
            if (sc.cfg >= 0 && sc.k.M == p.M && sc.k.N == p.N && sc.k.K == p.K && sc.k.taps == p.taps && sc.k.kind == kind) return &sc;
    }
    return nullptr;
}

static int pick_cfg(const GemmP& p, int batch, int* split_out, bool allow_split) {
    *split_out = 1;
    if (g_force_gemm_cfg >= 0 && cfg_valid(g_force_gemm_cfg, p)) {
        if (allow_split && g_force_gemm_split > 1 && p.K / 64 / g_force_gemm_split >= 4) *split_out = std::min(g_force_gemm_split, 8);   // workspace: 8 slices
        return g_force_gemm_cfg;
    }
    if (g_force_gemm_split == 1) allow_split = false;
    if (g_shortk_gemm_cfg >= 0 && p.taps == 1 && p.K <= g_shortk_max_k && !(p.flags & EP_GEGLU) && cfg_valid(g_shortk_gemm_cfg, p))
        return g_shortk_gemm_cfg;
    if (g_geglu_gemm_cfg >= 0 && (p.flags & EP_GEGLU) && cfg_valid(g_geglu_gemm_cfg, p)) return g_geglu_gemm_cfg;
    if (batch == 1) {
        if (const ShapeChoice* sc = find_shape(p)) {
            if (cfg_valid(sc->cfg, p) && (sc->split <= 1 || (allow_split && p.K / 64 / sc->split >= 2))) {
                *split_out = sc->split > 1 ? sc->split : 1;
                return sc->cfg;
            }
        }
    }
    const int cands[] = {CFG_256x320, CFG_256x256, CFG_128x320, CFG_128x128_K32, CFG_128x128, CFG_128x64, CFG_64x64};
    int best = -1;
    float best_score = -1.f;
    const int nk = p.K / 64;
    for (int cfg : cands) {
        if (!cfg_valid(cfg, p)) continue;
        for (int split : {1, 2, 3, 4, 6, 8}) {
            if (split > 1 && (!allow_split || nk / split < 12)) break;
            const float sc = cfg_score(p, batch, cfg, split);
            if (sc > best_score * 1.03f) { best_score = sc; best = cfg; *split_out = split; }
        }
    }
    return best >= 0 ? best : CFG_64x64;
}

size_t gemm_splitk_ws_bytes(int M, int N, int K, int batch) {
    if (K < 64 * 16 || N % 64) return 0;                    // the score model splits at >= 12 BK-steps per slice; tuned shapes (gemm_tuned_shapes.inc) may go down to 8
    if ((long)cdiv(M, 128) * cdiv(N, 128) * batch >= 512) return 0;   // chip already full with ordinary tiles
    return (size_t)8 * M * N * batch * sizeof(float);
}

int launch_gemm(const GemmP& p_in, int batch, bool force_generic, bool use_glds, hipStream_t s, int* stats_nchunk_out, int* lnp_np_out) {
    GemmP p = p_in;
    p.stats_nchunk = 0;
    p.lnp_np = 0;
    if (stats_nchunk_out) *stats_nchunk_out = 0;
    if (lnp_np_out) *lnp_np_out = 0;
    p.flags |= g_gemm_dbgflags;
    if (p.n_valid <= 0 || p.n_valid > p.N) p.n_valid = p.N;
    if (p.bias_scale == 0.f) p.bias_scale = 1.f;
    p.korder = (p.taps == 9 && g_conv_korder == 1 && !p.up && !(p.flags & EP_WRAP)) ? 1 : 0;     // (the ping-pong kernel's KORD form has no upsample / wrap addressing)
    p.zero = zero_page();
    SDMI_REQUIRE(p.zero != nullptr, "zero page allocation failed");
    SDMI_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "empty GEMM");
    SDMI_REQUIRE(p.cin % 8 == 0, "Cin must be a multiple of 8 (pad channels)");
    // algorithmic work of this launch: 2*M*N*K flops; bytes = activations read once + weights + output written once
    const double pf_flops = 2.0 * p.M * (double)p.N * p.K * batch;
    const double pf_in = (p.taps == 9 ? (double)p.M / (p.stride * p.stride) * (p.up ? 0.25 : 1.0) : (double)p.M) * p.cin * 2.0;
    const double pf_bytes = (pf_in + (double)p.N * p.K * 2.0 + (double)p.M * p.N * ((p.flags & EP_OUT_F32) ? 4.0 : 2.0)) * batch;
    if (p.flags & EP_LNFOLD) {
        SDMI_REQUIRE(p.ln_stats && p.ln_s && p.taps == 1 && !p.a1 && !p.stats_out && !(p.flags & (EP_NCHW | EP_OUT_F32 | EP_BIAS_ROW)),
                     "EP_LNFOLD: 1x1 / linear layers with fp16 row-major (or transposed) output only");
        SDMI_REQUIRE(!force_generic && use_glds && gemm_mfma_supported(p), "EP_LNFOLD runs on the LDS-direct MFMA kernels only");
    }
    if (force_generic || !gemm_mfma_supported(p)) {
        SDMI_REQUIRE(!(p.flags & EP_HILO), "(hi, lo) stream tensors need the MFMA kernels (shape not supported / force_generic)");
        const bool geglu = p.flags & EP_GEGLU;
        const long total = (long)p.M * (geglu ? p.N / 2 : p.N);
        int blocks = (int)std::min<long>((total + 255) / 256, 65535L * 8);
        ProfScope ps("gemm_generic", pf_flops, pf_bytes, s);
        hipLaunchKernelGGL(gemm_generic_kernel, dim3(blocks, 1, batch), dim3(256), 0, s, p);
        SDMI_CHECK_HIP(hipGetLastError());
        return 0;
    }
    {
        // 16-byte epilogue accesses need 16-byte aligned rows and bases (EP_TRANSPOSE: 8 consecutive tokens inside one image)
        auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
        bool wide = g_ep_wide != 0 && al16(p.out) && p.ldo % 8 == 0 && p.o_bs % 8 == 0 && p.N % 8 == 0;
        if (p.resid) wide = wide && al16(p.resid) && p.ldr % 8 == 0 && p.r_bs % 8 == 0;
        if (p.flags & EP_TRANSPOSE) wide = wide && p.rows_per_batch % 8 == 0 && p.M % 8 == 0;
        if (!wide) p.flags |= EP_NARROW;
    }
    // (hi, lo) stream tensors (EP_HILO, engine option "residual_fp32"): plain fp16 row-major launches.  Round 6: they take split-K (the
    // reduce pass carries the pair), the GroupNorm-statistics epilogue (sums of hi + lo: what the norm reads) and 16-byte accesses —
    // gemm_epilogue_hilo, a separate region of the epilogue on the tiles with row-tile pairs; elsewhere the 8-byte general path.
    // No LayerNorm row partials (the option and "ln_fold" are not combined).
    const bool hilo = (p.flags & EP_HILO) != 0;
    if (hilo) {
        SDMI_REQUIRE(!force_generic && !(p.flags & (EP_GEGLU | EP_NCHW | EP_TRANSPOSE | EP_OUT_F32 | EP_LNFOLD | EP_BIAS_ROW | EP_QUICK_GELU | EP_GELU)) &&
                         (gemm_resid_lo(p) == nullptr || p.resid != nullptr),
                     "(hi, lo) stream tensors: plain fp16 row-major epilogue on the MFMA kernels only");
        auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
        if (!al16(gemm_out_lo(p)) || !al16(gemm_resid_lo(p))) p.flags |= EP_NARROW;
        p.lnp_out = nullptr;
    }
    int split = 1;
    const bool can_split = p.splitk_ws != nullptr && !(p.flags & (EP_GEGLU | EP_NCHW | EP_TRANSPOSE)) && p.N % 4 == 0;
    SDMI_REQUIRE(!(p.flags & EP_TRANSPOSE) || (p.rows_per_batch % 4 == 0 && p.M % 4 == 0 && !(p.flags & (EP_GEGLU | EP_NCHW | EP_OUT_F32 | EP_BIAS_ROW)) &&
                                               !p.resid && !p.rowbias),
                 "EP_TRANSPOSE: rows per image must be a multiple of 4; no residual / GEGLU / fp32 output");
    const int cfg_picked = pick_cfg(p, batch, &split, can_split);
    // ((hi, lo) launches on the 256x320 tile — 160 accumulator registers per lane — take gemm_epilogue_hilo's plain form without the
    // hoisted bias, and no statistics form: the consumer keeps its own statistics pass there.  Measured alternatives, round 6
    // (profiles/r06_fwd_ab_accuracy*.txt): the 8-byte general path 80.9 us, the 128x320 tile 65.8 us on M65536 N320 K320 (fp16 launch:
    // 35.4 us); the level-0 convolutions 189 us on 128x320 with the statistics form against 123 us + a 17 us longer norm on 256x320.)
    const int cfg = cfg_picked;
    // ping-pong kernel: every weight row must exist (no n_valid masking) and a (tap, source) segment must fit the zero page
    const bool phase = use_glds && (g_gemm_pipe == 3 || g_gemm_pipe == 4) &&
                       (cfg == CFG_256x320 || cfg == CFG_256x256 || cfg == CFG_256x128 || (cfg == CFG_128x320 && g_gemm_pipe == 4)) && p.n_valid == p.N &&
                       p.cin + 64 <= kZeroPageHalfs &&
                       ((p.taps == 1 && p.stride == 1 && !p.up && p.Ho == p.Hi && p.Wo == p.Wi) ||
                        (p.M / p.rows_per_batch < 128 && (p.up ? 2 : 1) * p.Hi < 2040 && (p.up ? 2 : 1) * p.Wi < 2040 &&
                         p.stride * p.Ho < 2040 && p.stride * p.Wo < 2040));
    // Row-shared 3x3 walk (g_conv_korder == 2, gemm_mfma_pingpong_dx_kernel): stride-1 "same" convs on the ping-pong tiles whose BM rows
    // are whole image rows (then the dx = 0 / 2 taps are LDS-row shifts of the staged dx = 1 tile).  Everything else keeps the tap-major walk.
    const bool dxwalk = g_conv_korder == 2 && phase && p.taps == 9 && p.stride == 1 && p.pad == 1 && !p.up &&
                        !(p.flags & (EP_WRAP | EP_GEGLU | EP_TRANSPOSE | EP_LNFOLD)) && p.Ho == p.Hi && p.Wo == p.Wi && p.Wo % 16 == 0 &&
                        kCfgBM[cfg] % p.Wo == 0 && p.Wi < 2040 && p.Hi < 2040 && !p.lnp_out && !(g_gemm_dbg && p.stats_nchunk == 0);
    if (dxwalk) p.korder = 2;
    if (split > 1) {
        const int nk = p.K / (cfg == CFG_128x128_K32 ? 32 : 64);
        p.splitk_steps = cdiv(nk, split);
        if (dxwalk) p.splitk_steps = cdiv(p.splitk_steps, 3) * 3;       // a slice is a whole number of (dy, channel block) groups
        p.splitk = cdiv(nk, p.splitk_steps);
        if (p.splitk <= 1) { p.splitk = 0; split = 1; }
    } else {
        p.splitk = 0;
    }
    {
        // Which operand should an XCD-local run of consecutive tiles share?  One XCD (its own 4 MB L2) executes tiles/8 consecutive
        // logical tiles.  N first: the run covers ~run/tiles_n row panels of A and min(run, tiles_n) weight panels; M first the
        // other way round.  Bytes an XCD pulls across the fabric = panels x panel size (A panels count their RAW pixels: the 9 taps of
        // a 3x3 conv re-read the same rows); pick the order that moves less.  (M = 1024, N = 1280, K = 11520 — the 8x8 level — read
        // all 29.5 MB of weights into each of the 8 L2s under N-first: 236 MB per launch, fabric-bound at 4 TB/s.)
        const int BMc = kCfgBM[cfg], BNc = kCfgBN[cfg];
        const long tiles_m = cdiv(p.M, BMc), tiles_n = p.N / BNc;
        const double run = std::max(1.0, (double)tiles_m * tiles_n / 8.0);
        const double a_panel = (double)BMc * p.cin * 2.0 * (p.taps == 9 ? 1.3 : 1.0) / ((p.taps == 9 && p.up) ? 2.0 : 1.0);
        const double w_panel = (double)BNc * p.K * 2.0 / (split > 1 ? split : 1);
        const double n_first = std::ceil(run / tiles_n) * a_panel + std::min<double>(run, tiles_n) * w_panel;
        const double m_first = std::ceil(run / tiles_m) * w_panel + std::min<double>(run, tiles_m) * a_panel;
        p.tile_order = g_tile_order >= 0 ? g_tile_order : (m_first < 0.8 * n_first ? 1 : 0);
        if (tiles_n == 1 || tiles_m == 1) p.tile_order = 0;
    }
    if (p.stats_out && g_gn_fuse && use_glds && split <= 1 && batch == 1 && !(p.flags & (EP_GEGLU | EP_TRANSPOSE | EP_NCHW | EP_OUT_F32 | EP_BIAS_ROW | EP_QUICK_GELU | EP_GELU)) && p.stats_cpg > 0) {
        const int BMc = kCfgBM[cfg], BNc = kCfgBN[cfg];
        // a tile must lie inside one image, a group inside one column tile; few enough chunks that gn_apply's prologue stays short
        // ((hi, lo) launches: the statistics form lives in gemm_epilogue_hilo — row-tile pairs, 16-byte accesses)
        if (p.rows_per_batch % BMc == 0 && BNc % p.stats_cpg == 0 && p.N % p.stats_cpg == 0 && p.rows_per_batch / BMc <= 64 &&
            p.M % p.rows_per_batch == 0 && (!hilo || (cfg != CFG_64x64 && cfg != CFG_256x320 && !(p.flags & EP_NARROW)))) {
            p.stats_nchunk = p.rows_per_batch / BMc;
            if (stats_nchunk_out) *stats_nchunk_out = p.stats_nchunk;
        }
    }
    // LayerNorm row partials from this launch's epilogue (GemmP::lnp_out): the 16-byte plain epilogue of an unsplit fp16 launch only
    if (p.lnp_out && !hilo && use_glds && split <= 1 && batch == 1 && p.stats_nchunk == 0 &&
        !(p.flags & (EP_NARROW | EP_GEGLU | EP_TRANSPOSE | EP_NCHW | EP_OUT_F32 | EP_BIAS_ROW | EP_QUICK_GELU | EP_GELU | EP_LNFOLD)) &&
        kCfgBM[cfg] % 32 == 0 && cfg != CFG_64x64 && (phase || kCfgBM[cfg] < 256)) {   // (64x64: one 16-row tile per wave, no row-tile
                                                                                         // pairs; two-stage 256-row tiles: not instantiated)
        p.lnp_np = p.N / kCfgBN[cfg];
        if (lnp_np_out) *lnp_np_out = p.lnp_np;
    }
    std::string pname;
    if (prof_enabled()) {
        pname = std::string(kCfgName[cfg]) + (phase ? "pp" : "") + (split > 1 ? "_splitk" + std::to_string(p.splitk) : "") + (p.taps == 9 ? "_conv3x3" : "_1x1") +
                ((p.flags & EP_GEGLU) ? "_geglu" : "") + ((p.flags & EP_TRANSPOSE) ? "_tr" : "") + ((p.flags & EP_HILO) ? "_hl" : "") + (p.tile_order ? "_mf" : "") + (p.stats_nchunk ? "_gn" : "") + (p.korder == 2 ? "_dx" : "") + " M" + std::to_string(p.M) + " N" + std::to_string(p.N) + " K" + std::to_string(p.K) +
                ((p.flags & EP_LNFOLD) ? " ln" : "") + (batch > 1 ? " x" + std::to_string(batch) : "");
    }
    ProfScope ps(pname.c_str(), pf_flops, pf_bytes, s);
    struct Reduce {     // second pass of split-K runs when the main kernel has been enqueued (scope exit of the switch)
        const GemmP& p; int batch; hipStream_t s; bool on;
        int run() const {
            if (!on) return 0;
            const long total = (long)p.M * (p.N / 4);      // per batch element: the kernel loops over the batch itself
            const dim3 grid((unsigned)std::min<long>((total + 255) / 256, 8192));
            if (p.flags & EP_LNFOLD) hipLaunchKernelGGL(splitk_reduce_kernel<true>, grid, dim3(256), 0, s, p, batch);
            else hipLaunchKernelGGL(splitk_reduce_kernel<false>, grid, dim3(256), 0, s, p, batch);
            SDMI_CHECK_HIP(hipGetLastError());
            return 0;
        }
    } reduce{p, batch, s, split > 1};
#define SDMI_CASE(ID, BM, BN, WR, WC, BK)                                                                       \
    case ID:                                                                                                    \
        if (use_glds ? launch_cfg<BM, BN, WR, WC, BK, true>(p, batch, s) : launch_cfg<BM, BN, WR, WC, BK, false>(p, batch, s)) \
            return 1;                                                                                           \
        return reduce.run();
    if (phase) {
        int rc = 1;
        switch (cfg) {
            case CFG_256x320: rc = launch_pingpong<256, 320>(p, batch, s); break;
            case CFG_256x256: rc = launch_pingpong<256, 256>(p, batch, s); break;
            case CFG_256x128: rc = launch_pingpong<256, 128>(p, batch, s); break;
            case CFG_128x320: rc = launch_pingpong<128, 320>(p, batch, s); break;
            default: break;
        }
        if (rc) return 1;
        return reduce.run();
    }
    switch (cfg) {
        SDMI_CASE(CFG_128x128, 128, 128, 2, 2, 64)
        SDMI_CASE(CFG_64x64, 64, 64, 4, 1, 64)
        SDMI_CASE(CFG_128x128_K32, 128, 128, 2, 2, 32)
        SDMI_CASE(CFG_256x256, 256, 256, 4, 2, 64)
        SDMI_CASE(CFG_256x128, 256, 128, 4, 2, 64)
        SDMI_CASE(CFG_256x320, 256, 320, 4, 2, 64)
        SDMI_CASE(CFG_128x64, 128, 64, 4, 1, 64)
        SDMI_CASE(CFG_128x320, 128, 320, 2, 4, 64)
        SDMI_CASE(CFG_128x160, 128, 160, 2, 2, 64)
        // ring forms need the LDS-direct path; without it they run as their two-stage twins
        case CFG_128x160_R4: if (use_glds ? launch_ring<128, 160, 2, 2, 4>(p, batch, s) : launch_cfg<128, 160, 2, 2, 64, false>(p, batch, s)) return 1; return reduce.run();
        case CFG_128x160_R3: if (use_glds ? launch_ring<128, 160, 2, 2, 3>(p, batch, s) : launch_cfg<128, 160, 2, 2, 64, false>(p, batch, s)) return 1; return reduce.run();
        case CFG_128x128_R4: if (use_glds ? launch_ring<128, 128, 2, 2, 4>(p, batch, s) : launch_cfg<128, 128, 2, 2, 64, false>(p, batch, s)) return 1; return reduce.run();
        case CFG_128x64_R3: if (use_glds ? launch_ring<128, 64, 4, 1, 3>(p, batch, s) : launch_cfg<128, 64, 4, 1, 64, false>(p, batch, s)) return 1; return reduce.run();
    }
#undef SDMI_CASE
    set_error("bad gemm config");
    return 1;
}

}  // namespace sdmi
