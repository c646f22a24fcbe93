// Step glue of the denoising loop as three tiny kernels, so that one step is [hipGraph: embed + pack + UNet] + [CFG/DPM update]
// with no framework launches in between (reference: models/unet_3d_condition_mask.py:376,408-428 and models/pipeline.py:165-192):
//  * timestep_embed_kernel   diffusers Timesteps(flip_sin_to_cos=True, downscale_freq_shift=0): [n] -> [n, dim] (cos | sin)
//  * pack_latents_kernel     cat([condition_latent, sample], dim=2), cat([mask, .], dim=1), NCTHW -> channels-last tokens
//                            zero-padded to 8 channels, classifier-free-guidance duplication of the batch
//  * cfg_dpm_step_tok_kernel aa_cfg_dpm_step reading the UNet's token-layout output (frame 0 dropped, :522) and writing
//                            the next UNet input
#pragma once
#include "dev.h"
#include "aa_mi355.h"
#include "conv_gemm.h"      // silu_f, gelu_erf_f

namespace aa {

template <typename T>
__global__ void __launch_bounds__(256) timestep_embed_kernel(const float* t, T* out, int n, int dim) {
    const int half = dim >> 1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n * half; i += gridDim.x * 256) {
        const int r = i / half, k = i - r * half;
        const float freq = expf(-9.210340371976184f * (float)k / (float)half);        // exp(-ln(10000) * k / half)
        const float arg = t[r] * freq;
        out[(int64_t)r * dim + k] = (T)cosf(arg);                                       // flip_sin_to_cos: [cos | sin]
        out[(int64_t)r * dim + half + k] = (T)sinf(arg);
    }
}

// one thread per output token: 8 channels = 16 bytes
template <typename T>
__global__ void __launch_bounds__(256) pack_latents_kernel(const AaPackLatents p) {
    const int T1 = p.frames + 1;
    const int64_t tokens = (int64_t)p.batch * T1 * p.hw;
    const T* cond = reinterpret_cast<const T*>(p.cond);
    const T* mask = reinterpret_cast<const T*>(p.mask);
    for (int64_t tok = (int64_t)blockIdx.x * 256 + threadIdx.x; tok < tokens; tok += (int64_t)gridDim.x * 256) {
        const int pix = (int)(tok % p.hw);
        const int64_t img = tok / p.hw;
        const int f = (int)(img % T1), b = (int)(img / T1);
        Pack8<T> o;
        o.raw = u32x4{0u, 0u, 0u, 0u};
        int ch = 0;
        if (mask) o.e[ch++] = mask[(int64_t)(b % p.mask_batch) * p.hw + pix];           // same mask for every frame (:425)
        for (int c = 0; c < p.channels; ++c) {
            float v;
            if (f == 0) v = (float)cond[((int64_t)(b % p.cond_batch) * p.channels + c) * p.hw + pix];   // frame 0 = condition latent (:376)
            else {
                const int64_t idx = (((int64_t)(b % p.sample_batch) * p.channels + c) * p.frames + (f - 1)) * p.hw + pix;
                v = p.sample_dtype == AA_F32 ? reinterpret_cast<const float*>(p.sample)[idx] : (float)reinterpret_cast<const T*>(p.sample)[idx];
            }
            o.e[ch++] = (T)v;
        }
        *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.out) + tok * 8) = o.raw;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) cfg_dpm_step_tok_kernel(const AaDpmStepTok p) {
    const T* eps = reinterpret_cast<const T*>(p.eps_tokens);
    float* x = reinterpret_cast<float*>(p.latents);
    float* x0p = reinterpret_cast<float*>(p.x0_prev);
    const int T1 = p.frames + 1;
    const int64_t n = (int64_t)p.clips * p.channels * p.frames * p.hw;
    if (p.next_t && blockIdx.x == 0 && threadIdx.x < p.next_t_count) p.next_t[threadIdx.x] = p.next_t_value;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int pix = (int)(i % p.hw);
        int64_t r = i / p.hw;
        const int f = (int)(r % p.frames); r /= p.frames;
        const int c = (int)(r % p.channels);
        const int b = (int)(r / p.channels);
        const int64_t tok_u = ((int64_t)b * T1 + f + 1) * p.hw + pix;                   // frame 0 of the UNet output is dropped (:522)
        const float u = (float)eps[tok_u * p.eps_ld + c];
        float e = u;
        if (p.guidance_on) {
            const int64_t tok_t = ((int64_t)(p.clips + b) * T1 + f + 1) * p.hw + pix;   // [uncond clips | text clips] (pipeline.py:165)
            e = u + p.guidance * ((float)eps[tok_t * p.eps_ld + c] - u);
        }
        const float xi = x[i];
        const float x0 = (xi - p.sigma_s * e) / p.alpha_s;
        const float nx = p.c_x * xi - p.c_d0 * x0 - p.c_d1 * (x0 - x0p[i]);
        x[i] = nx;
        x0p[i] = x0;
        if (p.latents_lp) reinterpret_cast<T*>(p.latents_lp)[i] = (T)nx;
    }
}

// ---- Stable-Video-Diffusion glue (reference models/pipeline.py:223-731) ----
// out = act(a*x + b*y + rowvec[(row / div) % mod]): diffusers AlphaBlender, the frame-position embedding add of
// TransformerSpatioTemporalModel, silu(emb + aug_emb).  One 16-byte chunk per thread.
template <typename T>
__global__ void __launch_bounds__(256) blend_kernel(const AaBlend p) {
    const int cpr = p.channels >> 3;
    const int64_t total = p.rows * cpr;
    const T* x = reinterpret_cast<const T*>(p.x);
    const T* y = reinterpret_cast<const T*>(p.y);
    const T* rv = reinterpret_cast<const T*>(p.rowvec);
    T* out = reinterpret_cast<T*>(p.out);
    const int rv_ld = p.rowvec_ld ? p.rowvec_ld : p.channels;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < total; c += (int64_t)gridDim.x * 256) {
        const int64_t r = c / cpr;
        const int n = (int)(c - r * cpr) * 8;
        Pack8<T> xv, yv, vv, o;
        xv.raw = *reinterpret_cast<const u32x4*>(x + r * p.channels + n);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p.a * (float)xv.e[e];
        if (y) {
            yv.raw = *reinterpret_cast<const u32x4*>(y + r * p.channels + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += p.b * (float)yv.e[e];
        }
        if (rv) {
            int64_t g = r / p.rowvec_div;
            if (p.rowvec_mod) g %= p.rowvec_mod;
            vv.raw = *reinterpret_cast<const u32x4*>(rv + g * rv_ld + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)vv.e[e];
        }
        if (p.act == AA_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
        } else if (p.act == AA_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_erf_f(v[e]);
        } else if (p.act == AA_ACT_QUICK_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] / (1.0f + __expf(-1.702f * v[e]));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o.e[e] = (T)v[e];
        *reinterpret_cast<u32x4*>(out + r * p.channels + n) = o.raw;
    }
}

// cat([mask, latents * scale, image_latents], dim=2) of [B, F, C, hw] sources -> channels-last tokens (pipeline.py:417-422)
template <typename T, int OC>
__global__ void __launch_bounds__(256) pack_frames_kernel(const AaPackFrames p) {
    const int64_t tokens = (int64_t)p.batch * p.frames * p.hw;
    const float scale = p.scale ? *p.scale : 1.0f;
    for (int64_t tok = (int64_t)blockIdx.x * 256 + threadIdx.x; tok < tokens; tok += (int64_t)gridDim.x * 256) {
        const int pix = (int)(tok % p.hw);
        const int64_t img = tok / p.hw;
        const int f = (int)(img % p.frames), b = (int)(img / p.frames);
        Pack8<T> o[OC / 8];
#pragma unroll
        for (int q = 0; q < OC / 8; ++q) o[q].raw = u32x4{0u, 0u, 0u, 0u};
        int ch = 0;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (!p.src[s]) continue;
            const int cs = p.src_channels[s];
            const int64_t base = (((int64_t)(b % p.src_batch[s]) * p.frames + f) * cs) * p.hw + pix;
            for (int c = 0; c < cs; ++c, ++ch) {
                float v = p.src_f32[s] ? reinterpret_cast<const float*>(p.src[s])[base + (int64_t)c * p.hw]
                                       : (float)reinterpret_cast<const T*>(p.src[s])[base + (int64_t)c * p.hw];
                if (s == p.scaled_src) v *= scale;
                const T tv = (T)v;
#pragma unroll
                for (int q = 0; q < OC / 8; ++q)            // (static register indices: a dynamic o[ch >> 3].e[ch & 7] would live in scratch)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (ch == q * 8 + e) o[q].e[e] = tv;
            }
        }
#pragma unroll
        for (int q = 0; q < OC / 8; ++q) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.out) + tok * OC + q * 8) = o[q].raw;
    }
}

// per-frame guidance + Euler (v-prediction) update, x' = c_x * x + c_v * v (pipeline.py:435-440; EulerDiscreteScheduler.step)
template <typename T>
__global__ void __launch_bounds__(256) cfg_euler_step_tok_kernel(const AaEulerStepTok p) {
    const T* vt = reinterpret_cast<const T*>(p.v_tokens);
    float* x = reinterpret_cast<float*>(p.latents);
    const int64_t n = (int64_t)p.clips * p.frames * p.channels * p.hw;
    if (p.next_t && blockIdx.x == 0 && threadIdx.x < p.next_t_count) p.next_t[threadIdx.x] = p.next_t_value;
    if (p.next_scale && blockIdx.x == 0 && threadIdx.x == 0) *p.next_scale = p.next_scale_value;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int pix = (int)(i % p.hw);
        int64_t r = i / p.hw;
        const int c = (int)(r % p.channels); r /= p.channels;
        const int f = (int)(r % p.frames);
        const int b = (int)(r / p.frames);
        const int64_t tok_u = ((int64_t)b * p.frames + f) * p.hw + pix;
        float v = (float)vt[tok_u * p.ld + c];
        if (p.guidance) {
            const int64_t tok_c = ((int64_t)(p.clips + b) * p.frames + f) * p.hw + pix;     // [uncond clips | cond clips]
            v = v + p.guidance[f] * ((float)vt[tok_c * p.ld + c] - v);
        }
        x[i] = p.c_x * x[i] + p.c_v * v;
    }
}

}  // namespace aa
