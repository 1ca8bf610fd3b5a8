// elementwise.hip — the latency-/HBM-bound small kernels of the hot path (gfx950):
//   * Philox4x32-10 + Box-Muller noise, bit-compatible with /root/reference/modules/rng_philox.py:32-102
//   * CFG batch build / combine and sampler updates (fused replacements for the ~3B tiny torch kernels per step of
//     /root/reference/modules/sd_samplers_cfg_denoiser.py:74-82,203-205 and the k-diffusion sampler loops)
//   * layout / dtype conversion at the engine boundary, timestep embedding
//     (/root/reference/modules/sd_hijack_unet.py:58-78), small-M linear layers (time embedding MLP), row softmax (VAE
//     mid attention), weight repacking, final uint8 conversion (/root/reference/modules/processing.py:1004-1005,1034-1035).
#include "common.h"
#include "prof.h"
#include <algorithm>
#include <cstdio>

// The sampler / RNG arithmetic mirrors separately-rounded torch / numpy ops: never contract a*b+c into an fma here.
#pragma clang fp contract(off)

namespace sdmi {

static inline int ew_blocks(int64_t n, int per_thread = 1) {
    int64_t b = (n + 256LL * per_thread - 1) / (256LL * per_thread);
    return (int)std::max<int64_t>(1, std::min<int64_t>(b, 4096));
}

// ------------------------------------------------------------------------------------------------------------
// Philox4x32-10 (counter = [offset, 0, i, 0], key = seed) + Box-Muller on outputs 0/1.
// The reference evaluates Box-Muller in float64 with float32-rounded constants (numpy promotes uint32*float32 to
// float64) and rounds to float32 once; the same expression order is used here so results agree bit-for-bit except
// where the device libm's double log/sin differ from the host's by more than half a float32 ulp (measured: none in
// the golden fixtures).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void philox_randn_kernel(float* out, long n, unsigned k0i, unsigned k1i, unsigned offset) {
    const float two_pow32_inv_f = 2.3283064e-10f;
    const float two_pow32_inv_2pi_f = (float)(2.3283064e-10 * 6.2831855);
    const double c1 = (double)two_pow32_inv_f, c2 = (double)two_pow32_inv_2pi_f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        unsigned c0 = offset, cc1 = 0u, c2w = (unsigned)i, c3 = 0u;
        unsigned k0 = k0i, k1 = k1i;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const unsigned long long p0 = (unsigned long long)c0 * 0xD2511F53ull;
            const unsigned long long p1 = (unsigned long long)c2w * 0xCD9E8D57ull;
            const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0;
            const unsigned hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
            const unsigned n0 = hi1 ^ cc1 ^ k0, n2 = hi0 ^ c3 ^ k1;
            c0 = n0; cc1 = lo1; c2w = n2; c3 = lo0;
            if (r != 9) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        }
        const double u = (double)c0 * c1 + c1 / 2;
        const double v = (double)cc1 * c2 + c2 / 2;
        const double sq = sqrt(-2.0 * log(u));
        out[i] = (float)(sq * sin(v));
    }
}

int launch_philox(float* out, int64_t n, uint64_t seed, uint32_t offset, hipStream_t s) {
    SDMI_REQUIRE(n < (1LL << 32), "philox stream index is 32-bit in the reference (rng_philox.py:92)");
    hipLaunchKernelGGL(philox_randn_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, out, (long)n, (unsigned)(seed & 0xFFFFFFFFu),
                       (unsigned)(seed >> 32), offset);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// CFG / sampler arithmetic (fp32 state)
// ------------------------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void cfg_prepare_kernel(const float* x, const float* c_in, TO* xin, int B, int reps, long chw) {
    const long n = (long)B * chw;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int b = (int)(i / chw);
        const float v = x[i] * (c_in ? c_in[b] : 1.0f);
        const TO hv = (TO)v;
        for (int r = 0; r < reps; ++r) xin[(long)r * n + i] = hv;
    }
}
int launch_cfg_prepare(const float* x, const float* c_in, void* xin, int out_dtype, int B, int reps, int64_t chw, hipStream_t s) {
    if (out_dtype == 0)
        hipLaunchKernelGGL(cfg_prepare_kernel<half_t>, dim3(ew_blocks((int64_t)B * chw)), dim3(256), 0, s, x, c_in, (half_t*)xin, B, reps, (long)chw);
    else
        hipLaunchKernelGGL(cfg_prepare_kernel<float>, dim3(ew_blocks((int64_t)B * chw)), dim3(256), 0, s, x, c_in, (float*)xin, B, reps, (long)chw);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

// x_in rows of a UNet whose input is cat([x, c_concat], dim=1) (inpainting / InstructPix2Pix checkpoints): row r*B+b holds
// x[b]*c_in[b] in channels [0,C) and cond[b] in channels [C,C+Cc) — zeros there for the repetitions named in zero_reps.
template <typename TO>
__global__ __launch_bounds__(256) void cfg_prepare_concat_kernel(const float* x, const float* c_in, const float* cond, TO* xin,
                                                                int B, int reps, int C, int Cc, long hw, unsigned zero_reps) {
    const long per = (long)(C + Cc) * hw, n = (long)B * per;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int b = (int)(i / per);
        const long rem = i - (long)b * per;
        const int ch = (int)(rem / hw);
        const long px = rem - (long)ch * hw;
        const bool is_x = ch < C;
        const float v = is_x ? x[((long)b * C + ch) * hw + px] * (c_in ? c_in[b] : 1.0f) : cond[((long)b * Cc + (ch - C)) * hw + px];
        for (int r = 0; r < reps; ++r) xin[(long)r * n + i] = (TO)((!is_x && ((zero_reps >> r) & 1u)) ? 0.0f : v);
    }
}
int launch_cfg_prepare_concat(const float* x, const float* c_in, const float* cond, void* xin, int out_dtype, int B, int reps,
                              int C, int Cc, int64_t hw, unsigned zero_reps, hipStream_t s) {
    const int64_t n = (int64_t)B * (C + Cc) * hw;
    if (out_dtype == 0)
        hipLaunchKernelGGL(cfg_prepare_concat_kernel<half_t>, dim3(ew_blocks(n)), dim3(256), 0, s, x, c_in, cond, (half_t*)xin, B, reps, C, Cc,
                           (long)hw, zero_reps);
    else
        hipLaunchKernelGGL(cfg_prepare_concat_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, s, x, c_in, cond, (float*)xin, B, reps, C, Cc,
                           (long)hw, zero_reps);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void cfg_combine_kernel(const float* x, const float* eps, const float* c_out, float cond_scale,
                                                         int mode, const float* mask, const float* nmask, const float* init,
                                                         float* den, int B, long chw) {
    const long n = (long)B * chw;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int b = (int)(i / chw);
        const float ec = eps[i], eu = eps[n + i];
        float d;
        if (mode == 0) {
            // CompVisDenoiser: denoised = input + eps * c_out, evaluated for both halves, then combine_denoised:
            // denoised = u; denoised += (c - u) * (weight * cond_scale)
            const float xv = x[i], co = c_out[b];
            const float dc = xv + ec * co, du = xv + eu * co;
            d = du + (dc - du) * cond_scale;
        } else {
            d = eu + (ec - eu) * cond_scale;
        }
        if (mask) d = d * nmask[i] + init[i] * mask[i];
        den[i] = d;
    }
}
int launch_cfg_combine(const float* x, const float* eps, const float* c_out, float cond_scale, int mode, const float* mask,
                       const float* nmask, const float* init_latent, float* den, int B, int64_t chw, hipStream_t s) {
    hipLaunchKernelGGL(cfg_combine_kernel, dim3(ew_blocks((int64_t)B * chw)), dim3(256), 0, s, x, eps, c_out, cond_scale, mode,
                       mask, nmask, init_latent, den, B, (long)chw);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

// v-prediction models: each half is first mapped through an affine wrapper  d_h = out_h * c_out[b] + x * c_skip[b]
//   sigma space   : k-diffusion CompVisVDenoiser  (c_skip = 1/(s^2+1), c_out = -s/sqrt(s^2+1)) -> denoised
//   timestep space: CompVisTimestepsVDenoiser.predict_eps_from_z_and_v (modules/sd_samplers_timesteps.py:38-39:
//                   c_out = sqrt(alpha_t), c_skip = sqrt(1 - alpha_t)) -> eps
// then combined like cfg_combine_kernel.
__global__ __launch_bounds__(256) void cfg_combine_affine_kernel(const float* x, const float* out, const float* c_out,
                                                                const float* c_skip, float cond_scale, const float* mask,
                                                                const float* nmask, const float* init, float* den, int B, long chw) {
    const long n = (long)B * chw;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int b = (int)(i / chw);
        const float xv = x[i], co = c_out[b], cs = c_skip[b];
        const float dc = out[i] * co + xv * cs, du = out[n + i] * co + xv * cs;
        float d = du + (dc - du) * cond_scale;
        if (mask) d = d * nmask[i] + init[i] * mask[i];
        den[i] = d;
    }
}
int launch_cfg_combine_affine(const float* x, const float* out, const float* c_out, const float* c_skip, float cond_scale,
                              const float* mask, const float* nmask, const float* init_latent, float* den, int B, int64_t chw,
                              hipStream_t s) {
    hipLaunchKernelGGL(cfg_combine_affine_kernel, dim3(ew_blocks((int64_t)B * chw)), dim3(256), 0, s, x, out, c_out, c_skip,
                       cond_scale, mask, nmask, init_latent, den, B, (long)chw);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void euler_step_kernel(float* x, const float* den, const float* noise, float sigma,
                                                        float sigma_down, float sigma_up, float s_noise, long n) {
    const float dt = sigma_down - sigma;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float xv = x[i];
        const float d = (xv - den[i]) / sigma;     // to_d
        xv = xv + d * dt;
        if (noise) xv = xv + noise[i] * s_noise * sigma_up;
        x[i] = xv;
    }
}
int launch_euler_step(float* x, const float* den, const float* noise, float sigma, float sigma_down, float sigma_up,
                      float s_noise, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(euler_step_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, den, noise, sigma, sigma_down, sigma_up,
                       s_noise, (long)n);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void dpmpp2m_step_kernel(float* x, const float* den, const float* old, float ratio, float em1,
                                                          float c1, float c2, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float dd = den[i];
        if (old) dd = c1 * dd - c2 * old[i];       // denoised_d = (1 + 1/(2r)) * denoised - (1/(2r)) * old_denoised
        x[i] = ratio * x[i] - em1 * dd;            // x = (sigma_next/sigma) * x - expm1(-h) * denoised_d
    }
}
int launch_dpmpp2m_step(float* x, const float* den, const float* old, float ratio, float em1, float c1, float c2, int64_t n,
                        hipStream_t s) {
    hipLaunchKernelGGL(dpmpp2m_step_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, den, old, ratio, em1, c1, c2, (long)n);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void ddim_step_kernel(float* x, const float* e, const float* noise, float* pred_out, float a_t,
                                                       float a_prev, float sigma_t, float somat, long n) {
    // sd_samplers_timesteps_impl.py:25-36: `alphas_prev[index].item() * s_x` etc. are python-scalar * fp32-tensor
    // products, so every per-step coefficient and the whole update are fp32 (the float64 of :15 only affects how the
    // host computes a_prev / sigma_t before rounding them to fp32).
    const float sqrt_at = sqrtf(a_t);
    const float sqrt_aprev = sqrtf(a_prev);
    const float dir_c = sqrtf(1.0f - a_prev - sigma_t * sigma_t);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float ev = e[i];
        const float pred = (x[i] - somat * ev) / sqrt_at;
        const float dir_xt = dir_c * ev;
        float r = sqrt_aprev * pred + dir_xt;
        r = r + (noise ? sigma_t * noise[i] : 0.0f);
        if (pred_out) pred_out[i] = pred;
        x[i] = r;
    }
}
int launch_ddim_step(float* x, const float* e, const float* noise, float* pred_x0, float a_t, float a_prev, float sigma_t,
                     float somat, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(ddim_step_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, e, noise, pred_x0, a_t, a_prev, sigma_t, somat,
                       (long)n);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void axpby_kernel(float* y, const float* x, float a, const float* z, float b, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v = x[i] * a;
        if (z) v = v + z[i] * b;
        y[i] = v;
    }
}
int launch_axpby(float* y, const float* x, float a, const float* z, float b, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, y, x, a, z, b, (long)n);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

// DPM adaptive (k-diffusion DPMSolver.dpm_solver_adaptive): squared mixed-tolerance error of the embedded pair,
//   sum_i ((lo_i - hi_i) / max(atol, rtol * max(|lo_i|, |prev_i|)))^2,
// reduced to 256 per-block partial sums in a fixed order (thread-strided accumulation, LDS tree): bit-reproducible; the host adds
// the 256 partials in float64.
__global__ __launch_bounds__(256) void dpm_error_kernel(const float* lo, const float* hi, const float* prev, float atol, float rtol,
                                                        float* partial, long n) {
    __shared__ float red[256];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float a = lo[i];
        const float delta = fmaxf(atol, rtol * fmaxf(fabsf(a), fabsf(prev[i])));
        const float e = (a - hi[i]) / delta;
        acc = fmaf(e, e, acc);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
int launch_dpm_error(const float* lo, const float* hi, const float* prev, float atol, float rtol, float* partial256, int64_t n,
                     hipStream_t s) {
    hipLaunchKernelGGL(dpm_error_kernel, dim3(256), dim3(256), 0, s, lo, hi, prev, atol, rtol, partial256, (long)n);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

struct LinTerms {
    const float* t[6];
    float c[6];
    int n;
};
// out = sum_k c[k] * t[k], accumulated left to right in fp32 (out may alias any term: each element is read before written)
__global__ __launch_bounds__(256) void lincomb_kernel(float* out, LinTerms lt, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v = lt.t[0][i] * lt.c[0];
#pragma unroll
        for (int k = 1; k < 6; ++k)
            if (k < lt.n) v = v + lt.t[k][i] * lt.c[k];
        out[i] = v;
    }
}
int launch_lincomb(float* out, const float* const* terms, const float* coefs, int n_terms, int64_t n, hipStream_t s) {
    SDMI_REQUIRE(n_terms >= 1 && n_terms <= 6, "lincomb takes 1..6 terms");
    LinTerms lt{};
    lt.n = n_terms;
    for (int k = 0; k < n_terms; ++k) { lt.t[k] = terms[k]; lt.c[k] = coefs[k]; }
    hipLaunchKernelGGL(lincomb_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, out, lt, (long)n);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

// x = init * mask + nmask * x   (inpainting blend, modules/sd_samplers_cfg_denoiser.py:206-209 / :279-280)
__global__ __launch_bounds__(256) void mask_blend_kernel(float* x, const float* init, const float* mask, const float* nmask, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        x[i] = init[i] * mask[i] + nmask[i] * x[i];
}
int launch_mask_blend(float* x, const float* init, const float* mask, const float* nmask, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(mask_blend_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, init, mask, nmask, (long)n);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

// Latent resample of the hires-fix pass: torch.nn.functional.interpolate(samples, size, mode, antialias=False) on [B,4,h,w]
// fp32 (modules/processing.py:1392; upscaler table modules/shared.py:54-62).  Index arithmetic follows ATen's upsample
// kernels (align_corners = False): scale = in / out; nearest: floor(dst * scale); nearest-exact: floor((dst + 0.5) * scale);
// bilinear: src = max(scale * (dst + 0.5) - 0.5, 0); bicubic: same src unclamped, cubic convolution with A = -0.75 and
// border-clamped taps.  mode: 0 nearest, 1 nearest-exact, 2 bilinear, 3 bicubic.
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__global__ __launch_bounds__(256) void latent_resize_kernel(const float* in, float* out, int planes, int hi, int wi, int ho,
                                                            int wo, int mode) {
    const long n = (long)planes * ho * wo;
    const float sh = (float)hi / (float)ho, sw = (float)wi / (float)wo;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int x = (int)(i % wo), y = (int)((i / wo) % ho);
        const long pl = i / ((long)wo * ho);
        const float* src = in + pl * (long)hi * wi;
        float v;
        if (mode <= 1) {
            const float off = mode == 1 ? 0.5f : 0.f;
            const int ys = min((int)floorf(((float)y + off) * sh), hi - 1), xs = min((int)floorf(((float)x + off) * sw), wi - 1);
            v = src[(long)ys * wi + xs];
        } else if (mode == 2) {
            const float fy = fmaxf(sh * ((float)y + 0.5f) - 0.5f, 0.f), fx = fmaxf(sw * ((float)x + 0.5f) - 0.5f, 0.f);
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 + (y0 < hi - 1 ? 1 : 0), x1 = x0 + (x0 < wi - 1 ? 1 : 0);
            const float ly = fy - (float)y0, lx = fx - (float)x0;
            const float hy = 1.f - ly, hx = 1.f - lx;
            v = hy * (hx * src[(long)y0 * wi + x0] + lx * src[(long)y0 * wi + x1]) +
                ly * (hx * src[(long)y1 * wi + x0] + lx * src[(long)y1 * wi + x1]);
        } else if (mode >= 4) {
            // antialias = True (modes 4 bilinear, 5 bicubic): torch's separable area filter — per axis the window of source samples within
            // `support` = (interp size / 2) * max(scale, 1) of the output sample's centre scale * (i + 0.5), weights filter((j + 0.5 - centre)
            // / max(scale, 1)) renormalised to sum 1 over the part of the window inside the image (no border replication); triangle
            // filter for bilinear, Keys cubic with a = -0.5 for bicubic (the plain mode's is -0.75).  Shrinking widens the window with
            // the scale; enlarging it is 2 / 4-5 samples.
            const float half = mode == 4 ? 1.f : 2.f;
            const float suy = half * fmaxf(sh, 1.f), sux = half * fmaxf(sw, 1.f);
            const float ivy = sh >= 1.f ? 1.f / sh : 1.f, ivx = sw >= 1.f ? 1.f / sw : 1.f;
            const float cyc = sh * ((float)y + 0.5f), cxc = sw * ((float)x + 0.5f);
            const int ymin = max((int)(cyc - suy + 0.5f), 0), xmin = max((int)(cxc - sux + 0.5f), 0);
            const int ysize = min((int)(cyc + suy + 0.5f), hi) - ymin, xsize = min((int)(cxc + sux + 0.5f), wi) - xmin;
            auto filt = [&](float t) {
                t = fabsf(t);
                if (mode == 4) return t < 1.f ? 1.f - t : 0.f;
                return t < 1.f ? cubic1(t, -0.5f) : (t < 2.f ? cubic2(t, -0.5f) : 0.f);
            };
            float wys = 0.f, wxs = 0.f;
            for (int a = 0; a < ysize; ++a) wys += filt(((float)(a + ymin) - cyc + 0.5f) * ivy);
            for (int b = 0; b < xsize; ++b) wxs += filt(((float)(b + xmin) - cxc + 0.5f) * ivx);
            v = 0.f;
            for (int a = 0; a < ysize; ++a) {
                const float wy = filt(((float)(a + ymin) - cyc + 0.5f) * ivy) / wys;
                float row = 0.f;
                for (int b = 0; b < xsize; ++b)
                    row = row + (filt(((float)(b + xmin) - cxc + 0.5f) * ivx) / wxs) * src[(long)(a + ymin) * wi + b + xmin];
                v = v + wy * row;
            }
        } else {
            const float A = -0.75f;
            const float fy = sh * ((float)y + 0.5f) - 0.5f, fx = sw * ((float)x + 0.5f) - 0.5f;
            const int iy = (int)floorf(fy), ix = (int)floorf(fx);
            const float ty = fy - (float)iy, tx = fx - (float)ix;
            const float cy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
            const float cx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
            v = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int yy = min(max(iy - 1 + a, 0), hi - 1);
                float row = 0.f;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int xx = min(max(ix - 1 + b, 0), wi - 1);
                    row = row + cx[b] * src[(long)yy * wi + xx];
                }
                v = v + cy[a] * row;
            }
        }
        out[i] = v;
    }
}
int launch_latent_resize(const float* in, float* out, int planes, int hi, int wi, int ho, int wo, int mode, hipStream_t s) {
    SDMI_REQUIRE(mode >= 0 && mode <= 5 && planes > 0 && hi > 0 && wi > 0 && ho > 0 && wo > 0, "latent_resize: bad arguments");
    const long n = (long)planes * ho * wo;
    hipLaunchKernelGGL(latent_resize_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, in, out, planes, hi, wi, ho, wo, mode);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

// CLIP text embeddings: out[b,l,:] = (inputs_embeds[b,l,:] if given else tok_emb[token[b,l],:]) + pos_emb[l,:]  -> fp16 rows
// (transformers CLIPTextEmbeddings; in-repo twin modules/models/sd3/other_impls.py:117-125)
template <typename TT>
__global__ __launch_bounds__(256) void clip_embed_kernel(const int* tokens, const TT* tok_emb, const float* pos_emb,
                                                         const float* inputs_embeds, half_t* out, int L, int C, int vocab, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long row = i / C;
        const int c = (int)(i - row * C), l = (int)(row % L);
        float v;
        if (inputs_embeds) {
            v = inputs_embeds[i];
        } else {
            int t = tokens[row];
            t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
            v = (float)tok_emb[(long)t * C + c];
        }
        out[i] = (half_t)(v + pos_emb[(long)l * C + c]);
    }
}
int launch_clip_embed(const int* tokens, const void* tok_emb, int tok_dtype, const float* pos_emb, const float* inputs_embeds,
                      half_t* out, int B, int L, int C, int vocab, hipStream_t s) {
    const long n = (long)B * L * C;
    if (tok_dtype == 0)
        hipLaunchKernelGGL((clip_embed_kernel<half_t>), dim3(ew_blocks(n)), dim3(256), 0, s, tokens, (const half_t*)tok_emb, pos_emb,
                           inputs_embeds, out, L, C, vocab, n);
    else
        hipLaunchKernelGGL((clip_embed_kernel<float>), dim3(ew_blocks(n)), dim3(256), 0, s, tokens, (const float*)tok_emb, pos_emb,
                           inputs_embeds, out, L, C, vocab, n);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

// pooled[b,:] = hidden[b, argmax_l tokens[b,l], :]   (the EOS position: CLIP's EOS id is the largest id; other_impls.py:146)
__global__ __launch_bounds__(256) void clip_pool_kernel(const int* tokens, const half_t* hidden, float* pooled, int L, int C) {
    const int b = blockIdx.x;
    int best = 0, bv = tokens[(long)b * L];
    for (int l = 1; l < L; ++l) {
        const int t = tokens[(long)b * L + l];
        if (t > bv) { bv = t; best = l; }                // first maximum, like torch.argmax
    }
    for (int c = threadIdx.x; c < C; c += 256) pooled[(long)b * C + c] = (float)hidden[((long)b * L + best) * C + c];
}
int launch_clip_pool(const int* tokens, const half_t* hidden, float* pooled, int B, int L, int C, hipStream_t s) {
    hipLaunchKernelGGL(clip_pool_kernel, dim3(B), dim3(256), 0, s, tokens, hidden, pooled, L, C);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

// W' = W + scale * (up @ down): the LoRA weight delta of extensions-builtin/Lora/network_lora.py:65-80 (rebuild_conventional,
// lyco_helpers.py:9-15) folded into the weight, fp32 accumulate in k order.  up [rows][rank], down [rank][cols] (cols = Cin*kh*kw
// for conv weights), W any of fp16 / fp32; one thread per output element (a few MFLOP per layer, done once per LoRA change).
template <typename TW, typename TU, typename TD>
__global__ __launch_bounds__(256) void lora_merge_kernel(float* out, const TW* w, const TU* up, const TD* down, int rows,
                                                         int cols, int rank, float scale) {
    const long n = (long)rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
        float acc = 0.f;
        for (int k = 0; k < rank; ++k) acc += (float)up[(long)r * rank + k] * (float)down[(long)k * cols + c];
        out[i] = (float)w[i] + acc * scale;
    }
}
int launch_lora_merge(float* out, const void* w, int w_dtype, const void* up, int up_dtype, const void* down, int down_dtype,
                      int rows, int cols, int rank, float scale, hipStream_t s) {
    const long n = (long)rows * cols;
#define SDMI_LM(TW, TU, TD)                                                                                              \
    hipLaunchKernelGGL((lora_merge_kernel<TW, TU, TD>), dim3(ew_blocks(n)), dim3(256), 0, s, out, (const TW*)w,           \
                       (const TU*)up, (const TD*)down, rows, cols, rank, scale)
    const int sel = (w_dtype == 0 ? 0 : 4) | (up_dtype == 0 ? 0 : 2) | (down_dtype == 0 ? 0 : 1);
    switch (sel) {
        case 0: SDMI_LM(half_t, half_t, half_t); break;
        case 1: SDMI_LM(half_t, half_t, float); break;
        case 2: SDMI_LM(half_t, float, half_t); break;
        case 3: SDMI_LM(half_t, float, float); break;
        case 4: SDMI_LM(float, half_t, half_t); break;
        case 5: SDMI_LM(float, half_t, float); break;
        case 6: SDMI_LM(float, float, half_t); break;
        default: SDMI_LM(float, float, float); break;
    }
#undef SDMI_LM
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

// Variation-seed slerp of modules/rng.py:85-96 for one image: low / high [C][H][W]; the angle is measured per (c, w) column
// along H (the reference's dim 1), the mean cosine over all columns selects the (reversed-weight) lerp fallback.  One workgroup:
// the tensors are a few hundred KB and this runs once per job.
__global__ __launch_bounds__(256) void slerp_kernel(float* out, const float* low, const float* high, float val, int C, int H, int W,
                                                    float* dots) {
    __shared__ float red[256];
    const int cols = C * W;
    float part = 0.f;
    for (int col = threadIdx.x; col < cols; col += 256) {
        const int c = col / W, w = col - c * W;
        const long base = (long)c * H * W + w;
        float nl = 0.f, nh = 0.f;
        for (int h = 0; h < H; ++h) {
            const float l = low[base + (long)h * W], g = high[base + (long)h * W];
            nl += l * l;
            nh += g * g;
        }
        nl = sqrtf(nl);
        nh = sqrtf(nh);
        float dot = 0.f;
        for (int h = 0; h < H; ++h) dot += (low[base + (long)h * W] / nl) * (high[base + (long)h * W] / nh);
        dots[col] = dot;
        part += dot;
    }
    red[threadIdx.x] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const bool lerp = red[0] / (float)cols > 0.9995f;
    for (int col = threadIdx.x; col < cols; col += 256) {
        const int c = col / W, w = col - c * W;
        const long base = (long)c * H * W + w;
        float a, b;
        if (lerp) {
            a = val;
            b = 1.0f - val;
        } else {
            const float omega = acosf(dots[col]), so = sinf(omega);
            a = sinf((1.0f - val) * omega) / so;
            b = sinf(val * omega) / so;
        }
        for (int h = 0; h < H; ++h) out[base + (long)h * W] = a * low[base + (long)h * W] + b * high[base + (long)h * W];
    }
}
int launch_slerp(float* out, const float* low, const float* high, float val, int C, int H, int W, float* scratch, hipStream_t s) {
    hipLaunchKernelGGL(slerp_kernel, dim3(1), dim3(256), 0, s, out, low, high, val, C, H, W, scratch);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- LyCORIS weight deltas other than plain LoRA (extensions-builtin/Lora/network_{hada,lokr,ia3}.py, network.py:175-194).
// All fp32, W = the layer's current weight viewed [rows][cols] (cols = Cin*kh*kw), done once per network change.

// LoHa (network_hada.py:52): out = W + scale * a * b, a and b the two rebuilt low-rank products.
__global__ __launch_bounds__(256) void weight_hadamard_kernel(float* out, const float* w, const float* a, const float* b, float scale, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = w[i] + scale * (a[i] * b[i]);
}
// LoKr (network_lokr.py:19-23 make_kron): out[(i1*r2+i2)][(j1*c2+j2)][k] = W + scale * w1[i1][j1] * w2[i2][j2][k], k = kh*kw taps.
__global__ __launch_bounds__(256) void weight_kron_kernel(float* out, const float* w, const float* w1, const float* w2, int r1, int c1,
                                                          int r2, int c2, int k, float scale) {
    const long n = (long)r1 * r2 * c1 * c2 * k;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int kk = (int)(i % k);
        const long rc = i / k;
        const int col = (int)(rc % ((long)c1 * c2)), row = (int)(rc / ((long)c1 * c2));
        const int i1 = row / r2, i2 = row - i1 * r2, j1 = col / c2, j2 = col - j1 * c2;
        out[i] = w[i] + scale * (w1[(long)i1 * c1 + j1] * w2[((long)i2 * c2 + j2) * k + kk]);
    }
}
// IA3 (network_ia3.py:18-30): out = W + scale * W * v, v indexed by the input column (on_input) or the output row.
__global__ __launch_bounds__(256) void weight_ia3_kernel(float* out, const float* w, const float* v, int rows, int cols, int on_input,
                                                         float scale) {
    const long n = (long)rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
        out[i] = w[i] + scale * (w[i] * v[on_input ? c : r]);
    }
}
// DoRA (network.py:175-194 apply_weight_decompose): merged = W + delta; per INPUT channel j the L2 norm over (out, kh, kw);
// out = W + mult * (merged * dora_scale[j] / norm[j] - W).  One workgroup per input channel: reduce, then rescale.
__global__ __launch_bounds__(256) void weight_dora_kernel(float* out, const float* w, const float* delta, const float* dora_scale, int rows,
                                                          int cin, int k, float mult) {
    __shared__ float red[256];
    const int j = blockIdx.x;
    const long per_row = (long)cin * k;
    const int cnt = rows * k;
    float acc = 0.f;
    for (int e = threadIdx.x; e < cnt; e += 256) {
        const long idx = (long)(e / k) * per_row + (long)j * k + (e % k);
        const float m = w[idx] + delta[idx];
        acc += m * m;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const float f = dora_scale[j] / sqrtf(red[0]);
    for (int e = threadIdx.x; e < cnt; e += 256) {
        const long idx = (long)(e / k) * per_row + (long)j * k + (e % k);
        const float m = w[idx] + delta[idx];
        out[idx] = w[idx] + mult * (m * f - w[idx]);
    }
}
int launch_weight_hadamard(float* out, const float* w, const float* a, const float* b, float scale, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(weight_hadamard_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, out, w, a, b, scale, (long)n);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_weight_kron(float* out, const float* w, const float* w1, const float* w2, int r1, int c1, int r2, int c2, int k, float scale,
                       hipStream_t s) {
    hipLaunchKernelGGL(weight_kron_kernel, dim3(ew_blocks((int64_t)r1 * r2 * c1 * c2 * k)), dim3(256), 0, s, out, w, w1, w2, r1, c1, r2, c2, k, scale);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_weight_ia3(float* out, const float* w, const float* v, int rows, int cols, int on_input, float scale, hipStream_t s) {
    hipLaunchKernelGGL(weight_ia3_kernel, dim3(ew_blocks((int64_t)rows * cols)), dim3(256), 0, s, out, w, v, rows, cols, on_input, scale);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_weight_dora(float* out, const float* w, const float* delta, const float* dora_scale, int rows, int cin, int k, float mult,
                       hipStream_t s) {
    hipLaunchKernelGGL(weight_dora_kernel, dim3(cin), dim3(256), 0, s, out, w, delta, dora_scale, rows, cin, k, mult);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void image_to_u8_kernel(const float* img, uint8_t* out, int C, long HW, long n) {
    // n = B*HW*C output elements, NHWC
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long pix = (i / C) % HW;
        const long b = i / ((long)C * HW);
        float v = img[(b * C + c) * HW + pix];
        v = (v + 1.0f) / 2.0f;
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        out[i] = (uint8_t)(255.f * v);              // numpy astype(uint8) truncates
    }
}
int launch_image_to_u8(const float* img, uint8_t* out, int B, int C, int H, int W, hipStream_t s) {
    const long n = (long)B * C * H * W;
    hipLaunchKernelGGL(image_to_u8_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, img, out, C, (long)H * W, n);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// boundary conversions
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const T* x, half_t* out, int C, long HW, int cpad, float scale,
                                                          const float* mix_w, const float* mix_b, long npix, int lo_ch) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
        const long b = i / HW, pix = i - b * HW;
        float v[16];
        for (int c = 0; c < C; ++c) v[c] = (float)x[(b * C + c) * HW + pix] * scale;
        half_t* dst = out + i * cpad;
        if (mix_w) {
            // 1x1 conv over the C input channels (post_quant_conv) in fp32
            for (int o = 0; o < C; ++o) {
                float acc = mix_b ? mix_b[o] : 0.f;
                for (int c = 0; c < C; ++c) acc = fmaf(mix_w[o * C + c], v[c], acc);
                dst[o] = (half_t)acc;
            }
        } else {
            for (int c = 0; c < C; ++c) dst[c] = (half_t)v[c];
        }
        int cz = C;
        if (lo_ch) {                                         // (hi, lo) input: what the fp16 rounding dropped, in channels [C, 2C)
            for (int c = 0; c < C; ++c) dst[C + c] = (half_t)(v[c] - (float)(half_t)v[c]);
            cz = 2 * C;
        }
        for (int c = cz; c < cpad; ++c) dst[c] = (half_t)0.f;
    }
}
// lo_ch (engine option "residual_fp32", UNet input): channels [C, 2C) of the output take fp16(x - fp16(x)); the packed conv_in weights
// repeat their input channels there (launch_pack_conv_weight, flag bit 1), so the first conv sees the latent with ~22 bits — free: the
// K dimension is padded to 64 either way, and with zeros in those channels (the default) the repeated weights contribute exactly 0.
int launch_nchw_to_nhwc(const void* x, int dtype, half_t* out, int B, int C, int HW, int cpad, float scale, const float* mix_w,
                        const float* mix_b, hipStream_t s, bool lo_ch) {
    SDMI_REQUIRE(C <= 16 && cpad >= C, "nchw_to_nhwc: C <= 16");
    SDMI_REQUIRE(!lo_ch || (2 * C <= cpad && !mix_w), "nchw_to_nhwc: the lo channels need room in the padding");
    const long npix = (long)B * HW;
    if (dtype == 0)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<half_t>, dim3(ew_blocks(npix)), dim3(256), 0, s, (const half_t*)x, out, C, (long)HW,
                           cpad, scale, mix_w, mix_b, npix, lo_ch ? 1 : 0);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(ew_blocks(npix)), dim3(256), 0, s, (const float*)x, out, C, (long)HW,
                           cpad, scale, mix_w, mix_b, npix, lo_ch ? 1 : 0);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

// dst[b][p][c] = src[b][p][c] (+ src_lo) + ctrl[b][c][p]: a ControlNet residual (NCHW, the caller's dtype) added to an NHWC activation of
// the UNet (engine.cpp unet_run: the skip connections and the middle block's output).  A thread owns 8 channels of one pixel; adjacent
// threads own adjacent pixels, so each of the 8 strided reads of ctrl is coalesced across the wave.  (hi, lo) activations in, pair out.
template <typename T>
__global__ __launch_bounds__(256) void add_nchw_residual_kernel(const half_t* src, const half_t* src_lo, const T* ctrl, half_t* dst, half_t* dst_lo,
                                                                 int C, long HW, long total) {
    const int CV = C / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long pix = i % HW;
        const long rest = i / HW;
        const int cv = (int)(rest % CV);
        const long b = rest / CV;
        const long o = (b * HW + pix) * C + cv * 8;
        const h8 v = *reinterpret_cast<const h8*>(src + o);
        h8 l = {0, 0, 0, 0, 0, 0, 0, 0};
        if (src_lo) l = *reinterpret_cast<const h8*>(src_lo + o);
        h8 oh, ol;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = ((float)v[e] + (float)l[e]) + (float)ctrl[(b * C + cv * 8 + e) * HW + pix];
            oh[e] = (half_t)f;
            ol[e] = (half_t)(f - (float)oh[e]);
        }
        *reinterpret_cast<h8*>(dst + o) = oh;
        if (dst_lo) *reinterpret_cast<h8*>(dst_lo + o) = ol;
    }
}
int launch_add_nchw_residual(const half_t* src, const half_t* src_lo, const void* ctrl, int dtype, half_t* dst, half_t* dst_lo, int B, int C,
                             int HW, hipStream_t s) {
    SDMI_REQUIRE(C % 8 == 0 && (src_lo == nullptr) == (dst_lo == nullptr), "add_nchw_residual: C % 8 == 0; (hi, lo) in means (hi, lo) out");
    const long total = (long)B * HW * (C / 8);
    if (dtype == 0)
        hipLaunchKernelGGL(add_nchw_residual_kernel<half_t>, dim3(ew_blocks(total)), dim3(256), 0, s, src, src_lo, (const half_t*)ctrl, dst, dst_lo,
                           C, (long)HW, total);
    else
        hipLaunchKernelGGL(add_nchw_residual_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, src, src_lo, (const float*)ctrl, dst, dst_lo,
                           C, (long)HW, total);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void convert_kernel(const TI* src, TO* dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (TO)src[i];
}
int launch_copy_out(const float* src, void* dst, int dtype, int64_t n, hipStream_t s) {
    if (dtype == 0)
        hipLaunchKernelGGL((convert_kernel<float, half_t>), dim3(ew_blocks(n)), dim3(256), 0, s, src, (half_t*)dst, (long)n);
    else
        hipLaunchKernelGGL((convert_kernel<float, float>), dim3(ew_blocks(n)), dim3(256), 0, s, src, (float*)dst, (long)n);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_convert_to_f16(const void* src, int dtype, half_t* dst, int64_t n, hipStream_t s) {
    if (dtype == 0)
        hipLaunchKernelGGL((convert_kernel<half_t, half_t>), dim3(ew_blocks(n)), dim3(256), 0, s, (const half_t*)src, dst, (long)n);
    else
        hipLaunchKernelGGL((convert_kernel<float, half_t>), dim3(ew_blocks(n)), dim3(256), 0, s, (const float*)src, dst, (long)n);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}
template <typename TI>
__global__ __launch_bounds__(256) void ctx_compare_kernel(const TI* src, const half_t* cached, int L, int Lpad, int C, long n, int* gate) {
    bool diff = false;
    const long per = (long)L * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long b = i / per, r = i - b * per;
        const half_t v = (half_t)src[i], c = cached[b * (long)Lpad * C + r];
        diff |= __builtin_bit_cast(unsigned short, v) != __builtin_bit_cast(unsigned short, c);
    }
    if (__any(diff) && (threadIdx.x & 63) == 0) *gate = 1;      // racing writers all store the same value
}
template <typename TI>
__global__ __launch_bounds__(256) void ctx_update_gated_kernel(const TI* src, half_t* cached, int L, int Lpad, int C, long n, const int* gate) {
    if (*gate == 0) return;
    const long per = (long)L * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long b = i / per, r = i - b * per;
        cached[b * (long)Lpad * C + r] = (half_t)src[i];
    }
}
int launch_ctx_compare(const void* src, int dtype, const half_t* cached, int B, int L, int Lpad, int C, int* gate, hipStream_t s) {
    const long n = (long)B * L * C;
    if (dtype == 0)
        hipLaunchKernelGGL(ctx_compare_kernel<half_t>, dim3(ew_blocks(n)), dim3(256), 0, s, (const half_t*)src, cached, L, Lpad, C, n, gate);
    else
        hipLaunchKernelGGL(ctx_compare_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, s, (const float*)src, cached, L, Lpad, C, n, gate);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_ctx_update_gated(const void* src, int dtype, half_t* cached, int B, int L, int Lpad, int C, const int* gate, hipStream_t s) {
    const long n = (long)B * L * C;
    if (dtype == 0)
        hipLaunchKernelGGL(ctx_update_gated_kernel<half_t>, dim3(ew_blocks(n)), dim3(256), 0, s, (const half_t*)src, cached, L, Lpad, C, n, gate);
    else
        hipLaunchKernelGGL(ctx_update_gated_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, s, (const float*)src, cached, L, Lpad, C, n, gate);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_convert_to_f32(const void* src, int dtype, float* dst, int64_t n, hipStream_t s) {
    if (dtype == 0)
        hipLaunchKernelGGL((convert_kernel<half_t, float>), dim3(ew_blocks(n)), dim3(256), 0, s, (const half_t*)src, dst, (long)n);
    else
        hipLaunchKernelGGL((convert_kernel<float, float>), dim3(ew_blocks(n)), dim3(256), 0, s, (const float*)src, dst, (long)n);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void timestep_embedding_kernel(const T* t, float* out, int B, int dim) {
    const int half = dim / 2;
    const int n = B * half;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int b = i / half, j = i - b * half;
        // freqs = exp(-ln(10000) * j / half) in fp32, args = t.float() * freqs  (sd_hijack_unet.py:69-73)
        const float freq = expf(-9.210340371976184f * (float)j / (float)half);
        const float a = (float)t[b] * freq;
        out[(long)b * dim + j] = cosf(a);
        out[(long)b * dim + half + j] = sinf(a);
        if ((dim & 1) && j == 0) out[(long)b * dim + dim - 1] = 0.f;
    }
}
int launch_timestep_embedding(const void* t, int dtype, float* out, int B, int dim, hipStream_t s) {
    if (dtype == 0)
        hipLaunchKernelGGL(timestep_embedding_kernel<half_t>, dim3(ew_blocks((int64_t)B * dim / 2)), dim3(256), 0, s, (const half_t*)t, out, B, dim);
    else
        hipLaunchKernelGGL(timestep_embedding_kernel<float>, dim3(ew_blocks((int64_t)B * dim / 2)), dim3(256), 0, s, (const float*)t, out, B, dim);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

// one wave per output column n, ALL batch rows: the weight row is read once (51 MB for the fused ResBlock emb projection of
// SD1.5 — the earlier one-wave-per-(b, n) version re-read it B times: 265 MB / launch in the PMC profile).
template <int BMAX>
__global__ __launch_bounds__(256) void small_linear_kernel(const float* a, const half_t* w, const float* bias, const float* add,
                                                          float* out, int B, int N, int K, int lda, int ldo, int silu_in,
                                                          int silu_out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const half_t* wr = w + (long)n * K;
    for (int b0 = 0; b0 < B; b0 += BMAX) {
        float acc[BMAX];
#pragma unroll
        for (int i = 0; i < BMAX; ++i) acc[i] = 0.f;
        for (int k = lane * 8; k < K; k += 64 * 8) {
            const h8 wv = *reinterpret_cast<const h8*>(wr + k);
            float wf[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) wf[e] = (float)wv[e];
#pragma unroll
            for (int i = 0; i < BMAX; ++i) {
                if (b0 + i < B) {
                    const float* ar = a + (long)(b0 + i) * lda + k;
                    const f4 a0 = *reinterpret_cast<const f4*>(ar), a1 = *reinterpret_cast<const f4*>(ar + 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float av = e < 4 ? a0[e] : a1[e - 4];
                        if (silu_in) av = av / (1.0f + expf(-av));
                        acc[i] = fmaf(av, wf[e], acc[i]);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < BMAX; ++i) {
            float v = acc[i];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && b0 + i < B) {
                v += bias ? bias[n] : 0.f;
                if (silu_out) v = v / (1.0f + expf(-v));
                if (add) v += add[(long)(b0 + i) * ldo + n];
                out[(long)(b0 + i) * ldo + n] = v;
            }
        }
    }
}
// Round 4: the same product with the activations staged ONCE per workgroup in LDS (fp32 [B][K], silu applied while staging) and four
// output columns per wave and pass, so an activation value read from LDS feeds four fmas.  The wave-per-column kernel above re-reads
// the B x K activation block from L1 / L2 for every column — 80 KB per 2.5 KB weight row on the fused ResBlock embedding projection
// (16 x 1280 -> 17920: 1.4 GB of cache reads per launch, ~150 us for 46 MB of weights).  Per output the lane -> k assignment, the fma
// order and the shuffle tree are the kernel above's: identical bits (tests/test_gpu_ops.py).
template <int BMAX, int CW>
__global__ __launch_bounds__(256) void small_linear_lds_kernel(const float* a, const half_t* w, const float* bias, const float* add,
                                                              float* out, int B, int N, int K, int lda, int ldo, int silu_in,
                                                              int silu_out) {
    extern __shared__ __attribute__((aligned(16))) float sa[];            // [BMAX][K]; rows >= B are zero (computed, never stored)
    for (int idx = threadIdx.x * 4; idx < BMAX * K; idx += 256 * 4) {       // K % 8 == 0: a quad never straddles rows
        const int b = idx / K, k = idx - b * K;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (b < B) {
            v = *reinterpret_cast<const f4*>(a + (long)b * lda + k);
            if (silu_in) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f + expf(-v[e]));
            }
        }
        *reinterpret_cast<f4*>(sa + idx) = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (int n0 = wave * CW; n0 < N; n0 += nwaves * CW) {
        float acc[BMAX][CW];
#pragma unroll
        for (int i = 0; i < BMAX; ++i)
#pragma unroll
            for (int c = 0; c < CW; ++c) acc[i][c] = 0.f;
        for (int k = lane * 8; k < K; k += 64 * 8) {
            float wf[CW][8];
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const h8 wv = *reinterpret_cast<const h8*>(w + (long)min(n0 + c, N - 1) * K + k);
#pragma unroll
                for (int e = 0; e < 8; ++e) wf[c][e] = (float)wv[e];
            }
#pragma unroll
            for (int i = 0; i < BMAX; ++i) {
                const f4 a0 = *reinterpret_cast<const f4*>(sa + i * K + k), a1 = *reinterpret_cast<const f4*>(sa + i * K + k + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float av = e < 4 ? a0[e] : a1[e - 4];
#pragma unroll
                    for (int c = 0; c < CW; ++c) acc[i][c] = fmaf(av, wf[c][e], acc[i][c]);
                }
            }
        }
        // the xor butterfly leaves the full sum in EVERY lane (both partners add the same two values at each level), so lane
        // i * CW + c keeps output (i, c) and the BMAX * CW results leave in one guarded store per lane
        float mine = 0.f;
        int ln = lane;
        asm volatile("" : "+v"(ln));                         // (the 64 lane masks below are loop-invariant: hoisted they spill 72 SGPRs)
#pragma unroll
        for (int i = 0; i < BMAX; ++i) {
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                float v = acc[i][c];
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                mine = ln == i * CW + c ? v : mine;
            }
        }
        const int oi = lane / CW, n = n0 + lane % CW;
        if (lane < BMAX * CW && oi < B && n < N) {
            float v = mine + (bias ? bias[n] : 0.f);
            if (silu_out) v = v / (1.0f + expf(-v));
            if (add) v += add[(long)oi * ldo + n];
            out[(long)oi * ldo + n] = v;
        }
    }
}
int g_small_linear_lds = [] { const char* e = getenv("SDMI_SMALL_LINEAR_LDS"); return e ? atoi(e) : 1; }();

// y = silu(x) elementwise fp32 (same expression as small_linear's silu_in, so hoisting it out changes no bits)
__global__ __launch_bounds__(256) void silu_f32_kernel(const float* x, float* y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float av = x[i];
        y[i] = av / (1.0f + expf(-av));
    }
}
int launch_silu_f32(const float* x, float* y, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(silu_f32_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, y, (long)n);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- hypernetwork MLP pieces (modules/hypernetworks/hypernetwork.py:25-113): activation in place on fp16 rows, y = x + a * h ----
__device__ __forceinline__ float hn_act(float v, int kind) {
    switch (kind) {
        case 1: return fmaxf(v, 0.f);                                            // relu
        case 2: return v > 0.f ? v : 0.01f * v;                                  // leakyrelu (default slope)
        case 3: return v > 0.f ? v : expm1f(v);                                  // elu (alpha 1)
        case 4: return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);        // hardswish ("swish" in the webui's table)
        case 5: return tanhf(v);
        case 6: return 1.f / (1.f + expf(-v));                                   // sigmoid
        case 7: return v / (1.f + expf(-v));                                     // silu
        case 8: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));        // gelu (erf)
        case 9: return v * tanhf(log1pf(expf(fminf(v, 20.f))));                  // mish (softplus threshold 20)
        case 10: return fminf(fmaxf(v, 0.f), 6.f);                               // relu6
        case 11: return 1.0507009873554805f * (v > 0.f ? v : 1.6732632423543772f * expm1f(v));   // selu
        case 12: return v > 20.f ? v : log1pf(expf(v));                          // softplus (beta 1, threshold 20)
        case 13: return v / (1.f + fabsf(v));                                    // softsign
        case 14: return fminf(fmaxf(v, -1.f), 1.f);                              // hardtanh
        case 15: return fminf(fmaxf(v * (1.f / 6.f) + 0.5f, 0.f), 1.f);          // hardsigmoid
        default: return v;
    }
}
__global__ __launch_bounds__(256) void act_f16_kernel(half_t* x, long n, int kind) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] = (half_t)hn_act((float)x[i], kind);
}
int launch_act_f16(half_t* x, int64_t n, int kind, hipStream_t s) {
    hipLaunchKernelGGL(act_f16_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, (long)n, kind);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}
__global__ __launch_bounds__(256) void axpy_f16_kernel(half_t* y, const half_t* x, const half_t* h, float a, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = (half_t)((float)x[i] + a * (float)h[i]);
}
int launch_axpy_f16(half_t* y, const half_t* x, const half_t* h, float a, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(axpy_f16_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, y, x, h, a, (long)n);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_small_linear(const float* a, const half_t* w, const float* bias, const float* add, float* out, int B, int N, int K,
                        int lda, int ldo, bool silu_in, bool silu_out, hipStream_t s) {
    SDMI_REQUIRE(K % 8 == 0 && lda % 4 == 0, "small_linear: K % 8 == 0, lda % 4 == 0");
    char pname[64];
    if (prof_enabled()) snprintf(pname, sizeof pname, "small_linear B%d N%d K%d", B, N, K);
    ProfScope ps(pname, 2.0 * B * (double)N * K, (double)N * K * 2.0 + ((double)B * K + (double)B * N) * 4.0, s);     // weights once + in / out
    const int rows = B <= 4 ? 4 : 16;
    const size_t lds = (size_t)rows * K * sizeof(float);
    if (g_small_linear_lds && B <= 16 && lds <= 96 * 1024 && N >= 256) {         // activations fit LDS: staged once per workgroup
        auto kern = rows == 4 ? small_linear_lds_kernel<4, 4> : small_linear_lds_kernel<16, 4>;
        static PerDeviceOnce attr[2];
        if (attr[rows == 4].need()) SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        const int blocks = std::min(cdiv(N, 16), 256);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, a, w, bias, add, out, B, N, K, lda, ldo, silu_in ? 1 : 0, silu_out ? 1 : 0);
        SDMI_CHECK_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(small_linear_kernel<16>, dim3(cdiv(N, 4)), dim3(256), 0, s, a, w, bias, add, out, B, N, K, lda, ldo,
                       silu_in ? 1 : 0, silu_out ? 1 : 0);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

// row softmax over fp32 scores -> fp16 probabilities (one block per row, row re-read from L2)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* in, half_t* out, int cols, int ldi, int ldo) {
    __shared__ float red[8];
    const long row = blockIdx.x;
    const float* src = in + row * ldi;
    half_t* dst = out + row * ldo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int c = tid; c < cols; c += 256) mx = fmaxf(mx, src[c]);
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int c = tid; c < cols; c += 256) sum += expf(src[c] - mx);
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    const float inv = 1.0f / sum;
    for (int c = tid; c < ldo; c += 256) dst[c] = c < cols ? (half_t)(expf(src[c] - mx) * inv) : (half_t)0.f;
}
int launch_softmax_rows(const float* in, half_t* out, int64_t rows, int cols, int ldi, int ldo, hipStream_t s) {
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, in, out, cols, ldi, ldo);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// weight repacking (run once at load)
// ------------------------------------------------------------------------------------------------------------
// packed row index of output channel o under the GEGLU interleave: groups of 32 values followed by their 32 gates
__device__ __forceinline__ int geglu_row(int o, int O) {
    const int half = O / 2;
    return o < half ? (o >> 5) * 64 + (o & 31) : ((o - half) >> 5) * 64 + 32 + ((o - half) & 31);
}

template <typename T>
__global__ __launch_bounds__(256) void pack_conv_weight_kernel(const T* w, half_t* out, int O, int I, int KK, int O_pad, int I_pad,
                                                              int geglu) {
    // out[row][tap][i] over (O_pad, KK, I_pad); source OIHW = w[o][i][tap]
    const long n = (long)O_pad * KK * I_pad;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int i = (int)(idx % I_pad);
        const int tap = (int)((idx / I_pad) % KK);
        const int row = (int)(idx / ((long)I_pad * KK));
        // invert the row permutation: which source channel lands in `row`?
        int o = row;
        if ((geglu & 1) && row < O) {
            const int g = row >> 6, r = row & 63;
            o = r < 32 ? g * 32 + r : O / 2 + g * 32 + (r - 32);
        }
        float v = 0.f;
        const int is = ((geglu & 2) && i >= I && i < 2 * I) ? i - I : i;     // flag bit 1: input channels [I, 2I) repeat [0, I)
        if (o < O && is < I && row < O) v = (float)w[((long)o * I + is) * KK + tap];
        out[idx] = (half_t)v;
    }
}
int launch_pack_conv_weight(const void* w, int dtype, half_t* out, int O, int I, int kh, int kw, int O_pad, int I_pad, int geglu,
                            hipStream_t s) {
    const long n = (long)O_pad * kh * kw * I_pad;
    SDMI_REQUIRE(!(geglu & 1) || (O % 64 == 0 && O_pad == O), "GEGLU packing needs O % 64 == 0");
    SDMI_REQUIRE(!(geglu & 2) || 2 * I <= I_pad, "repeated input channels need room in the padding");
    if (dtype == 0)
        hipLaunchKernelGGL(pack_conv_weight_kernel<half_t>, dim3(ew_blocks(n)), dim3(256), 0, s, (const half_t*)w, out, O, I, kh * kw,
                           O_pad, I_pad, geglu);
    else
        hipLaunchKernelGGL(pack_conv_weight_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, s, (const float*)w, out, O, I, kh * kw,
                           O_pad, I_pad, geglu);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void pack_bias_kernel(const T* b, float* out, int O, int O_pad, int geglu) {
    for (int row = blockIdx.x * 256 + threadIdx.x; row < O_pad; row += gridDim.x * 256) {
        int o = row;
        if (geglu && row < O) {
            const int g = row >> 6, r = row & 63;
            o = r < 32 ? g * 32 + r : O / 2 + g * 32 + (r - 32);
        }
        out[row] = (row < O) ? (float)b[o] : 0.f;
    }
}
int launch_pack_bias(const void* b, int dtype, float* out, int O, int O_pad, int geglu, hipStream_t s) {
    if (dtype == 0)
        hipLaunchKernelGGL(pack_bias_kernel<half_t>, dim3(ew_blocks(O_pad)), dim3(256), 0, s, (const half_t*)b, out, O, O_pad, geglu);
    else
        hipLaunchKernelGGL(pack_bias_kernel<float>, dim3(ew_blocks(O_pad)), dim3(256), 0, s, (const float*)b, out, O, O_pad, geglu);
    SDMI_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace sdmi
