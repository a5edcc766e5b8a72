// HBM-bound elementwise / reduction kernels of the U-Net train step (gfx950).
// Every kernel is a single streaming pass; roofline = HBM bandwidth.
#include "psg_common.h"

namespace psg {

// psg_set_seed_source: one word per process (one process per GPU), read by every launch that draws a dropout mask
const uint64_t* g_seed_source = nullptr;
const uint64_t* seed_source() { return g_seed_source; }


static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    snprintf(g_err, sizeof(g_err), "HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
    (void)hipGetLastError();
    return PSG_ERR_HIP;
}

static inline int grid_for(int64_t n, int block, int max_blocks = 2048) {
    int64_t g = (n + block - 1) / block;
    if (g > max_blocks) g = max_blocks;
    if (g < 1) g = 1;
    return (int)g;
}

// torch.clamp(x, -3, 3) (improved_diffusion_trainer.py:363): NaN stays NaN (fminf / fmaxf alone would turn it into a bound
// and hide a bad latent from the NaN checks that follow)
__device__ __forceinline__ float clamp3(float x) { return x != x ? x : fminf(fmaxf(x, -3.0f), 3.0f); }

// ---------------------------------------------------------------------------
// add_noise: bit-exact (two rounded multiplies, one rounded add; no FMA contraction)
// ---------------------------------------------------------------------------
__global__ void noise_add_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                 const int64_t* __restrict__ t, const float* __restrict__ tabA,
                                 const float* __restrict__ tabB, float* __restrict__ out,
                                 int32_t* flag, int64_t B, int64_t chw, int num_t, int do_clamp) {
    const int64_t n = B * chw;
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / chw;
        int64_t tb = t[b];
        if (tb < 0 || tb >= num_t) { bad |= 2; tb = tb < 0 ? 0 : num_t - 1; }
        const float a = tabA[tb], c = tabB[tb];
        float x = x0[i];
        if (do_clamp) x = clamp3(x);
        const float r = __fadd_rn(__fmul_rn(a, x), __fmul_rn(c, noise[i]));
        out[i] = r;
        if (isnan(r) || isinf(r)) bad |= PSG_FLAG_FALLBACK;      // the reference then returns x0 + 0.1*noise (:61-63)
    }
    if (bad) atomicOr(flag, bad);
}

__global__ void noise_fallback_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                      float* __restrict__ out, int32_t* flag, int64_t n, int do_clamp) {
    // (workgroups that find the fallback bit set may OR bit 0 into the word below; the fallback bit itself is never cleared)
    if (((*flag) & PSG_FLAG_FALLBACK) == 0) return;
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float x = x0[i];
        if (do_clamp) x = clamp3(x);
        const float r = __fadd_rn(x, __fmul_rn(0.1f, noise[i]));
        out[i] = r;
        if (isnan(r) || isinf(r)) bad = PSG_FLAG_NOISY_BAD;     // the rescued batch is still non-finite: :376 skips it
    }
    if (bad) atomicOr(flag, bad);
}

__global__ void ddpm_update_kernel(float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ z,
                                   const float* __restrict__ c1t, const float* __restrict__ c2t,
                                   const float* __restrict__ sgt, const int32_t* t_dev, int64_t n) {
    const int t = *t_dev;
    const float c1 = c1t[t], c2 = c2t[t], sg = sgt[t];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float r = __fmul_rn(c1, __fsub_rn(x[i], __fmul_rn(c2, eps[i])));
        if (t > 0) r = __fadd_rn(r, __fmul_rn(sg, z[i]));
        x[i] = r;
    }
}

// The other samplers that consume the trained U-Net, each with the reference's own operation order so
// that fp32 results are bit-identical to the PyTorch CPU path (separately rounded multiply / divide / add, no FMA):
//   mode 1  final_trainer.py:58-69   mean = c0 * (x - (c1 * eps) / c2)  [+ c3 * z]      sample_previous_timestep
//   mode 2  final_trainer.py:203     x - eps                                            last step of FinalPokemonGenerator
//   mode 3  gradio_app.py:349-358    y = (x - c0 * eps) / c1  [; y = c2 * y + c3 * z]   the demo's ddpm_sample
__global__ void sampler_update_kernel(float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ z,
                                      int mode, float c0, float c1, float c2, float c3, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float xv = x[i], ev = eps[i];
        float r;
        if (mode == 1) {
            r = __fmul_rn(c0, __fsub_rn(xv, __fdiv_rn(__fmul_rn(c1, ev), c2)));
            if (z) r = __fadd_rn(r, __fmul_rn(c3, z[i]));
        } else if (mode == 2) {
            r = __fsub_rn(xv, ev);
        } else {
            r = __fdiv_rn(__fsub_rn(xv, __fmul_rn(c0, ev)), c1);
            if (z) r = __fadd_rn(__fmul_rn(c2, r), __fmul_rn(c3, z[i]));
        }
        x[i] = r;
    }
}

// ---------------------------------------------------------------------------
// deterministic two-stage reductions (fixed grid, fixed order)
// ---------------------------------------------------------------------------
constexpr int RED_BLOCKS = 1024;
constexpr int RED_THREADS = 256;

__global__ void smooth_l1_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                 float* __restrict__ grad, float* __restrict__ partial, int32_t* nan_flag,
                                 float beta, float grad_scale, int64_t n) {
    __shared__ float red[16];
    float acc = 0.f;
    int bad = 0;
    const float inv_n = 1.0f / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float p = pred[i];
        const float d = p - target[i];
        const float ad = fabsf(d);
        float g;
        if (ad < beta) { acc += 0.5f * d * d / beta; g = d / beta; }
        else { acc += ad - 0.5f * beta; g = d > 0.f ? 1.f : -1.f; }
        if (isnan(p) || isinf(p)) bad = 1;
        if (grad) grad[i] = g * inv_n * grad_scale;
    }
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
    if (bad && nan_flag) atomicOr(nan_flag, PSG_FLAG_PRED_BAD);
}

__global__ void sumsq_kernel(const float* __restrict__ g, float* __restrict__ partial, int64_t n) {
    __shared__ float red[16];
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 v = g4[i];
        acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float v = g[(n4 << 2) + threadIdx.x]; acc += v * v; }
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// final stage: one block sums `count` partials in a fixed order; out = scale*sum (+ out if accumulate)
__global__ void finish_sum_kernel(const float* __restrict__ partial, int count, float* out, float scale,
                                  int accumulate, int32_t* nan_flag) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < count; i += blockDim.x) acc += partial[i];
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) {
        float r = s * scale;
        if (accumulate) r += *out;
        *out = r;
        if (nan_flag && (isnan(r) || isinf(r))) atomicOr(nan_flag, PSG_FLAG_LOSS_BAD);
    }
}

// ---------------------------------------------------------------------------
// layout conversions at the NCHW fp32 boundary (tiny: 8 channels)
// ---------------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t ld, int B, int C, int HW) {
    const int64_t n = (int64_t)B * C * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t r = i / C;           // b*HW + p
        const int p = (int)(r % HW);
        const int64_t b = r / HW;
        Elem<T>::st(dst + r * ld + c, src[(b * C + c) * HW + p]);
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, int64_t ld, float* __restrict__ dst, int B, int C, int HW) {
    const int64_t n = (int64_t)B * C * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const int64_t r = i / HW;          // b*C + c
        const int c = (int)(r % C);
        const int64_t b = r / C;
        dst[i] = Elem<T>::ld(src + (b * HW + p) * ld + c);
    }
}

template <typename T>
__global__ void text_pool_kernel(const float* __restrict__ text, T* __restrict__ pooled, int64_t ldp,
                                 T* __restrict__ cast, int B, int S, int D) {
    const int b = blockIdx.x;
    for (int dcol = threadIdx.x; dcol < D; dcol += blockDim.x) {
        float acc = 0.f;
        for (int s = 0; s < S; ++s) {
            const float v = text[((int64_t)b * S + s) * D + dcol];
            acc += v;
            if (cast) Elem<T>::st(cast + ((int64_t)b * S + s) * D + dcol, v);
        }
        Elem<T>::st(pooled + (int64_t)b * ldp + dcol, acc / (float)S);
    }
}

template <typename T>
__global__ void sinusoid_kernel(const int64_t* __restrict__ t, const float* __restrict__ coeff, T* __restrict__ out,
                                int64_t ld, int B, int half) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, j = i % half;
    const float e = __fmul_rn((float)t[b], coeff[j]);   // t.float() * coeff, unet.py:47-49
    Elem<T>::st(out + (int64_t)b * ld + j, sinf(e));
    Elem<T>::st(out + (int64_t)b * ld + half + j, cosf(e));
}

// bilinear, align_corners=False: src = (dst + 0.5) * (in/out) - 0.5, clamped at 0 (PyTorch area_pixel_compute_source_index)
__device__ __forceinline__ void bilin_coord(int o, int in, int out, int& i0, int& i1, float& w1) {
    const float scale = (float)in / (float)out;
    float s = ((float)o + 0.5f) * scale - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    w1 = s - (float)i0;
}

template <typename T>
__global__ void upsample_fwd_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy,
                                    int B, int Hi, int Wi, int Ho, int Wo, int C) {
    const int c4n = C >> 2;
    const int64_t n = (int64_t)B * Ho * Wo * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        int64_t r = i / c4n;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int64_t b = r / Ho;
        int h0, h1, w0, w1; float lh, lw;
        bilin_coord(ho, Hi, Ho, h0, h1, lh);
        bilin_coord(wo, Wi, Wo, w0, w1, lw);
        const T* base = x + (b * Hi * Wi) * ldx + c;
        f32x4 v00 = load4<T>(base + ((int64_t)h0 * Wi + w0) * ldx);
        f32x4 v01 = load4<T>(base + ((int64_t)h0 * Wi + w1) * ldx);
        f32x4 v10 = load4<T>(base + ((int64_t)h1 * Wi + w0) * ldx);
        f32x4 v11 = load4<T>(base + ((int64_t)h1 * Wi + w1) * ldx);
        const float a00 = (1.f - lh) * (1.f - lw), a01 = (1.f - lh) * lw, a10 = lh * (1.f - lw), a11 = lh * lw;
        f32x4 o = v00 * a00 + v01 * a01 + v10 * a10 + v11 * a11;
        store4<T>(y + ((b * Ho + ho) * Wo + wo) * ldy + c, o);
    }
}

// gather-form backward: dx[hi,wi] = sum over the (few) output pixels whose stencil touches it; deterministic
template <typename T>
__global__ void upsample_bwd_kernel(const T* __restrict__ dy, int64_t lddy, T* __restrict__ dx, int64_t lddx,
                                    int B, int Hi, int Wi, int Ho, int Wo, int C) {
    const int c4n = C >> 2;
    const int64_t n = (int64_t)B * Hi * Wi * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        int64_t r = i / c4n;
        const int wi = (int)(r % Wi); r /= Wi;
        const int hi = (int)(r % Hi);
        const int64_t b = r / Hi;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // output rows that can touch hi: scale = Hi/Ho <= 1 -> ho in a small window
        const int ho_lo = max(0, (int)floorf(((float)hi - 1.0f + 0.5f) * (float)Ho / (float)Hi - 0.5f) - 1);
        const int ho_hi = min(Ho - 1, (int)ceilf(((float)hi + 1.0f + 0.5f) * (float)Ho / (float)Hi - 0.5f) + 1);
        const int wo_lo = max(0, (int)floorf(((float)wi - 1.0f + 0.5f) * (float)Wo / (float)Wi - 0.5f) - 1);
        const int wo_hi = min(Wo - 1, (int)ceilf(((float)wi + 1.0f + 0.5f) * (float)Wo / (float)Wi - 0.5f) + 1);
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            int h0, h1; float lh;
            bilin_coord(ho, Hi, Ho, h0, h1, lh);
            float wh = 0.f;
            if (h0 == hi) wh += 1.f - lh;
            if (h1 == hi) wh += lh;
            if (wh == 0.f) continue;
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                int w0, w1; float lw;
                bilin_coord(wo, Wi, Wo, w0, w1, lw);
                float ww = 0.f;
                if (w0 == wi) ww += 1.f - lw;
                if (w1 == wi) ww += lw;
                if (ww == 0.f) continue;
                f32x4 g = load4<T>(dy + ((b * Ho + ho) * Wo + wo) * lddy + c);
                acc += g * (wh * ww);
            }
        }
        store4<T>(dx + ((b * Hi + hi) * Wi + wi) * lddx + c, acc);
    }
}

// 8 bf16 channels (16 bytes) per lane, 32-bit index arithmetic: the 4-channel grid-stride forms above spend their time in
// 64-bit divisions and 8-byte accesses (forward 2.0, backward 1.2 TB/s in round 3's profile); same arithmetic per element.
__global__ __launch_bounds__(256) void upsample_fwd8_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy,
                                                            int B, int Hi, int Wi, int Ho, int Wo, int C) {
    const int c8n = C >> 3;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n = (uint32_t)B * Ho * Wo * c8n;
    if (i >= n) return;
    const uint32_t pix = i / (uint32_t)c8n;
    const int c = (int)(i - pix * c8n) * 8;
    const uint32_t bh = pix / (uint32_t)Wo;
    const int wo = (int)(pix - bh * Wo);
    const uint32_t b = bh / (uint32_t)Ho;
    const int ho = (int)(bh - b * Ho);
    int h0, h1, w0, w1; float lh, lw;
    bilin_coord(ho, Hi, Ho, h0, h1, lh);
    bilin_coord(wo, Wi, Wo, w0, w1, lw);
    const bf16_t* base = x + (int64_t)(b * Hi * Wi) * ldx + c;
    const bf16x8 v00 = *reinterpret_cast<const bf16x8*>(base + (h0 * Wi + w0) * ldx);
    const bf16x8 v01 = *reinterpret_cast<const bf16x8*>(base + (h0 * Wi + w1) * ldx);
    const bf16x8 v10 = *reinterpret_cast<const bf16x8*>(base + (h1 * Wi + w0) * ldx);
    const bf16x8 v11 = *reinterpret_cast<const bf16x8*>(base + (h1 * Wi + w1) * ldx);
    const float a00 = (1.f - lh) * (1.f - lw), a01 = (1.f - lh) * lw, a10 = lh * (1.f - lw), a11 = lh * lw;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)v00[e] * a00 + (float)v01[e] * a01 + (float)v10[e] * a10 + (float)v11[e] * a11);
    *reinterpret_cast<bf16x8*>(y + (int64_t)pix * ldy + c) = o;
}

__global__ __launch_bounds__(256) void upsample_bwd8_kernel(const bf16_t* __restrict__ dy, int lddy, bf16_t* __restrict__ dx, int lddx,
                                                            int B, int Hi, int Wi, int Ho, int Wo, int C) {
    const int c8n = C >> 3;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n = (uint32_t)B * Hi * Wi * c8n;
    if (i >= n) return;
    const uint32_t pix = i / (uint32_t)c8n;
    const int c = (int)(i - pix * c8n) * 8;
    const uint32_t bh = pix / (uint32_t)Wi;
    const int wi = (int)(pix - bh * Wi);
    const uint32_t b = bh / (uint32_t)Hi;
    const int hi = (int)(bh - b * Hi);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const int ho_lo = max(0, (int)floorf(((float)hi - 1.0f + 0.5f) * (float)Ho / (float)Hi - 0.5f) - 1);
    const int ho_hi = min(Ho - 1, (int)ceilf(((float)hi + 1.0f + 0.5f) * (float)Ho / (float)Hi - 0.5f) + 1);
    const int wo_lo = max(0, (int)floorf(((float)wi - 1.0f + 0.5f) * (float)Wo / (float)Wi - 0.5f) - 1);
    const int wo_hi = min(Wo - 1, (int)ceilf(((float)wi + 1.0f + 0.5f) * (float)Wo / (float)Wi - 0.5f) + 1);
    const bf16_t* base = dy + (int64_t)(b * Ho * Wo) * lddy + c;
    // the output columns whose stencil touches wi, with their weights, once (not once per output row); same order, same sums
    int wos[6]; float wws[6]; int nw = 0;
    for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        int w0, w1; float lw;
        bilin_coord(wo, Wi, Wo, w0, w1, lw);
        float ww = 0.f;
        if (w0 == wi) ww += 1.f - lw;
        if (w1 == wi) ww += lw;
        if (ww != 0.f && nw < 6) { wos[nw] = wo; wws[nw] = ww; ++nw; }
    }
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
        int h0, h1; float lh;
        bilin_coord(ho, Hi, Ho, h0, h1, lh);
        float wh = 0.f;
        if (h0 == hi) wh += 1.f - lh;
        if (h1 == hi) wh += lh;
        if (wh == 0.f) continue;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (k < nw) {
                const bf16x8 g = *reinterpret_cast<const bf16x8*>(base + (ho * Wo + wos[k]) * lddy);
                const float wgt = wh * wws[k];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)g[e] * wgt;
            }
        }
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)acc[e];
    *reinterpret_cast<bf16x8*>(dx + (int64_t)pix * lddx + c) = o;
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, int64_t lda, const T* __restrict__ b, int64_t ldb,
                           T* __restrict__ y, int64_t ldy, int64_t rows, int cols) {
    const int c4n = cols >> 2;
    const int64_t n = rows * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const int64_t r = i / c4n;
        store4<T>(y + r * ldy + c, load4<T>(a + r * lda + c) + load4<T>(b + r * ldb + c));
    }
}

// y = a (+ b (+ c)): row-strided operands, 16-byte chunks, ONE rounding of the fp32 sum.  One source = a strided copy
// (a skip tensor into its half of a concat buffer), three = the gradient fan-in of a skip tensor (next encoder stage +
// the two decoder blocks that read it) in one pass instead of autograd's clone + add + add.
template <typename T>
__global__ void sum_rows_kernel(const T* __restrict__ a, int64_t lda, const T* __restrict__ b, int64_t ldb, const T* __restrict__ c,
                                int64_t ldc, T* __restrict__ y, int64_t ldy, int64_t rows, int cols) {
    constexpr int CH = 16 / (int)sizeof(T);
    typedef float VecT __attribute__((ext_vector_type(CH)));
    const int cpr = cols / CH;
    const int64_t n = rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cpr;
        const int col = (int)(i - r * cpr) * CH;
        uint4 ra = *reinterpret_cast<const uint4*>(a + r * lda + col), rb = ra, rc = ra;
        if (b) rb = *reinterpret_cast<const uint4*>(b + r * ldb + col);
        if (c) rc = *reinterpret_cast<const uint4*>(c + r * ldc + col);
        if (b) {
            const T* pa = reinterpret_cast<const T*>(&ra);
            const T* pb = reinterpret_cast<const T*>(&rb);
            const T* pc = reinterpret_cast<const T*>(&rc);
            T o[CH];
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                float v = (float)pa[e] + (float)pb[e];
                if (c) v += (float)pc[e];
                o[e] = (T)v;
            }
            ra = *reinterpret_cast<const uint4*>(o);
        }
        *reinterpret_cast<uint4*>(y + r * ldy + col) = ra;
    }
}

template <typename T>
__global__ void dropout_apply_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy,
                                     int64_t rows, int cols, uint32_t thresh, uint64_t seed0, float scale, const uint64_t* seed_dev) {
    const uint64_t seed = eff_seed(seed0, seed_dev);
    const int c4n = cols >> 2;
    const int64_t n = rows * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const int64_t r = i / c4n;
        f32x4 v = load4<T>(x + r * ldx + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = drop_keep(seed, (uint64_t)r * cols + c + e, thresh) ? v[e] * scale : 0.f;
        store4<T>(y + r * ldy + c, v);
    }
}

template <typename T>
__global__ void epilogue_bwd_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ u, int64_t ldu,
                                    T* __restrict__ g, int64_t ldg, int64_t rows, int cols, int act, float alpha,
                                    uint32_t thresh, float drop_scale, uint64_t seed0, const uint64_t* seed_dev) {
    const uint64_t seed = eff_seed(seed0, thresh ? seed_dev : nullptr);
    const int c4n = cols >> 2;
    const int64_t n = rows * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const int64_t r = i / c4n;
        f32x4 v = load4<T>(dy + r * lddy + c) * alpha;
        if (thresh) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = drop_keep(seed, (uint64_t)r * cols + c + e, thresh) ? v[e] * drop_scale : 0.f;
        }
        if (act != PSG_ACT_NONE) {
            const f32x4 uu = load4<T>(u + r * ldu + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= act_grad(uu[e], act);
        }
        store4<T>(g + r * ldg + c, v);
    }
}

// ---------------------------------------------------------------------------
// optimizer
// ---------------------------------------------------------------------------
__device__ __forceinline__ float clip_coef_dev(const float* normsq, float max_norm) {
    if (!normsq) return 1.0f;
    const float nrm = sqrtf(*normsq);
    const float c = max_norm / (nrm + 1e-6f);
    return c < 1.0f ? c : 1.0f;
}

// Step count and learning rate either come from the host (step_dev == NULL: lr, bc1, bc2_sqrt as passed) or live on
// the device: step = *step_dev + 1 counts the updates that really happened (a skipped NaN batch advances nothing,
// improved_diffusion_trainer.py:353-393 `continue`), lr = lr_table[min(step - 1, lr_len - 1)] is the schedule value
// the reference's scheduler.step() would have set after that many optimizer steps.
struct AdamSched { const int32_t* step_dev; const float* lr_table; const float* beta1_table; int len; };
__device__ __forceinline__ void adam_sched(const AdamSched& sc, float& beta1, float beta2, float& lr, float& bc1, float& bc2_sqrt) {
    if (!sc.step_dev) return;
    const int step = *sc.step_dev + 1;
    const int k = min(step - 1, sc.len - 1);
    if (sc.lr_table) lr = sc.lr_table[k];
    if (sc.beta1_table) beta1 = sc.beta1_table[k];         // (torch computes the bias correction with the CURRENT beta1 too)
    bc1 = 1.0f - powf(beta1, (float)step);
    bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
}
__global__ void adam_step_advance_kernel(int32_t* step_dev, const int32_t* skip_flag) {
    if (!(skip_flag && (*skip_flag & PSG_FLAG_SKIP_MASK))) *step_dev += 1;
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, int64_t n, float lr, float beta1, float beta2, float eps,
                             float wd, float bc1, float bc2_sqrt, const float* normsq, float max_norm,
                             const int32_t* skip_flag, const AdamSched sc) {
    if (skip_flag && (*skip_flag & PSG_FLAG_SKIP_MASK)) return;
    adam_sched(sc, beta1, beta2, lr, bc1, bc2_sqrt);
    const float coef = clip_coef_dev(normsq, max_norm);
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= step_size * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

// float4 form for the flat arenas (n % 4 == 0, 16-byte aligned): same arithmetic per element, optional bf16 shadow
__global__ void adamw_kernel4(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                              float* __restrict__ v, int64_t n4, float lr, float beta1, float beta2, float eps,
                              float wd, float bc1, float bc2_sqrt, const float* normsq, float max_norm,
                              const int32_t* skip_flag, bf16_t* __restrict__ shadow, const AdamSched sc) {
    if (skip_flag && (*skip_flag & PSG_FLAG_SKIP_MASK)) return;
    adam_sched(sc, beta1, beta2, lr, bc1, bc2_sqrt);
    const float coef = clip_coef_dev(normsq, max_norm);
    const float step_size = lr / bc1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += 2 * stride) {
        // two independent float4 groups per trip: 8 x 16-byte loads in flight per lane before the first use
        const int64_t i1 = i0 + stride;
        const bool two = i1 < n4;
        const int64_t j1 = two ? i1 : i0;
        // (nontemporal loads / stores measured: -1 % at best, +17 % on large grids - plain accesses)
        auto ld = [](const float* a, int64_t i) { return reinterpret_cast<const f32x4*>(a)[i]; };
        auto stv = [](float* a, int64_t i, f32x4 x) { reinterpret_cast<f32x4*>(a)[i] = x; };
        f32x4 G[2] = {ld(g, i0), ld(g, j1)};
        f32x4 P[2] = {ld(p, i0), ld(p, j1)};
        f32x4 M[2] = {ld(m, i0), ld(m, j1)};
        f32x4 V[2] = {ld(v, i0), ld(v, j1)};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gi = G[u][e] * coef;
                float pi = P[u][e] * (1.0f - lr * wd);
                const float mi = beta1 * M[u][e] + (1.0f - beta1) * gi;
                const float vi = beta2 * V[u][e] + (1.0f - beta2) * gi * gi;
                const float denom = sqrtf(vi) / bc2_sqrt + eps;
                pi -= step_size * (mi / denom);
                P[u][e] = pi; M[u][e] = mi; V[u][e] = vi;
            }
        }
        stv(p, i0, P[0]); stv(m, i0, M[0]); stv(v, i0, V[0]);
        if (shadow) store4<bf16_t>(shadow + 4 * i0, P[0]);
        if (two) {
            stv(p, i1, P[1]); stv(m, i1, M[1]); stv(v, i1, V[1]);
            if (shadow) store4<bf16_t>(shadow + 4 * i1, P[1]);
        }
    }
}

__global__ void clip_scale_kernel(float* __restrict__ g, int64_t n, const float* normsq, float max_norm) {
    const float coef = clip_coef_dev(normsq, max_norm);
    if (coef >= 1.0f) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) g[i] *= coef;
}

// ---------------------------------------------------------------------------
// weight preparation: fp32 OIHW -> [O][Kpad] (kh,kw,ci) and [I][Kpad'] (kh,kw,co)
// ---------------------------------------------------------------------------
template <typename T>
__global__ void prep_weight_kernel(const float* __restrict__ w, int ohwi, T* __restrict__ wf, T* __restrict__ wd,
                                   int O, int I, int taps, int64_t kpf, int64_t kpd) {
    // One workgroup per (32 co) x (32 ci) x taps tile: W is read ONCE, coalesced (32*taps contiguous floats per
    // co row), transposed through LDS, and both prepared layouts are written in 64-byte runs.
    __shared__ float t[32][32 * 9 + 1];
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int nci = min(32, I - ci0), nco = min(32, O - co0);
    const int run = nci * taps;
    if (ohwi) {                                        // w[co][tap][ci]: nci-float runs
        for (int e = threadIdx.x; e < nco * taps * 32; e += blockDim.x) {
            const int ci = e & 31, rt = e >> 5;
            const int tap = rt % taps, r = rt / taps;
            if (ci < nci) t[r][ci * taps + tap] = w[((int64_t)(co0 + r) * taps + tap) * I + ci0 + ci];
        }
    } else {                                           // w[co][ci][tap]: nci*taps-float runs
        for (int e = threadIdx.x; e < nco * run; e += blockDim.x) {
            const int r = e / run, c = e - r * run;
            t[r][c] = w[((int64_t)(co0 + r) * I + ci0) * taps + c];
        }
    }
    __syncthreads();
    if (wf) {
        for (int e = threadIdx.x; e < nco * taps * 32; e += blockDim.x) {
            const int ci = e & 31, rt = e >> 5;
            const int tap = rt % taps, r = rt / taps;
            if (ci < nci) Elem<T>::st(wf + (int64_t)(co0 + r) * kpf + (int64_t)tap * I + ci0 + ci, t[r][ci * taps + tap]);
        }
        if (blockIdx.x == 0) {                         // zero the K padding of these rows
            const int64_t k0 = (int64_t)taps * I, npad = kpf - k0;
            for (int64_t e = threadIdx.x; e < nco * npad; e += blockDim.x)
                Elem<T>::st(wf + (int64_t)(co0 + e / npad) * kpf + k0 + e % npad, 0.f);
        }
    }
    if (wd) {
        for (int e = threadIdx.x; e < nci * taps * 32; e += blockDim.x) {
            const int r = e & 31, ct = e >> 5;
            const int tap = ct % taps, ci = ct / taps;
            if (r < nco) Elem<T>::st(wd + (int64_t)(ci0 + ci) * kpd + (int64_t)tap * O + co0 + r, t[r][ci * taps + tap]);
        }
        if (blockIdx.y == 0) {
            const int64_t k0 = (int64_t)taps * O, npad = kpd - k0;
            for (int64_t e = threadIdx.x; e < nci * npad; e += blockDim.x)
                Elem<T>::st(wd + (int64_t)(ci0 + e / npad) * kpd + k0 + e % npad, 0.f);
        }
    }
}

// The per-step case: bf16 OHWI shadow -> wd only (the data-gradient operand wd[ci][tap][co], a transposition of the
// shadow itself).  16-byte loads and stores (8 values), transposed through a bf16 LDS tile; the generic kernel below moves
// 2 bytes per lane and load (1.3 TB/s over the 146 launches of a step).  Needs O % 8 == 0 and I % 8 == 0.
__global__ __launch_bounds__(256) void prep_wd_bf16_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wd, int O, int I, int taps, int64_t kpd) {
    __shared__ unsigned short t[64][66];                  // [co][ci], 132-byte rows: the 8 strided 2-byte reads of a store chunk hit distinct banks
    const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64, tap = blockIdx.z;
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int co = k * 32 + (tid >> 3), ch = tid & 7;
        uint4 v = {0u, 0u, 0u, 0u};
        if (co0 + co < O && ci0 + ch * 8 < I) v = *reinterpret_cast<const uint4*>(w + ((int64_t)(co0 + co) * taps + tap) * I + ci0 + ch * 8);
        uint32_t* d = reinterpret_cast<uint32_t*>(&t[co][ch * 8]);      // (rows are 4-byte aligned: 132 = 4 * 33)
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int id = tid + 256 * k;
        const int ci = id >> 3, q = (id & 7) * 8;
        if (ci0 + ci < I && co0 + q < O) {
            uint32_t o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (uint32_t)t[q + 2 * e][ci] | ((uint32_t)t[q + 2 * e + 1][ci] << 16);
            const uint4 v = {o[0], o[1], o[2], o[3]};
            *reinterpret_cast<uint4*>(wd + (int64_t)(ci0 + ci) * kpd + (int64_t)tap * O + co0 + q) = v;
        }
    }
    if (blockIdx.y == 0 && tap == 0) {                    // zero the K padding of these rows
        const int64_t k0 = (int64_t)taps * O;
        const int npad = (int)(kpd - k0);
        for (int e = tid; e < 64 * npad; e += 256) {
            const int r = e / npad, c = e - r * npad;
            if (ci0 + r < I) wd[(int64_t)(ci0 + r) * kpd + k0 + c] = (bf16_t)0.f;
        }
    }
}

// OHWI master (w[co][tap][ci], the training layout) or any 1x1/linear weight: one 64co x 64ci tile of one tap per
// workgroup.  Shift-only index math, 256-byte coalesced reads, 8/16-byte stores of 4 elements along ci (wf) and
// along co (wd, transposed through LDS).  Needs O % 4 == 0 and I % 4 == 0.
template <typename S, typename T>
__global__ __launch_bounds__(256) void prep_weight_ohwi_kernel(const S* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd,
                                                               int O, int I, int taps, int64_t kpf, int64_t kpd) {
    __shared__ float t[64][65];
    const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64, tap = blockIdx.z;
    const int tid = threadIdx.x;
    {
        const int c = tid & 63, rr = tid >> 6;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = i * 4 + rr;
            float v = 0.f;
            if (co0 + co < O && ci0 + c < I) v = (float)w[((int64_t)(co0 + co) * taps + tap) * I + ci0 + c];
            t[co][c] = v;
        }
    }
    __syncthreads();
    const int q4 = (tid & 15) * 4, rq = tid >> 4;
    if (wf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = rq + 16 * i;
            if (co0 + co < O && ci0 + q4 < I) {
                f32x4 v = {t[co][q4], t[co][q4 + 1], t[co][q4 + 2], t[co][q4 + 3]};
                store4<T>(wf + (int64_t)(co0 + co) * kpf + (int64_t)tap * I + ci0 + q4, v);
            }
        }
        if (blockIdx.x == 0 && tap == 0) {             // zero the K padding of these rows
            const int64_t k0 = (int64_t)taps * I;
            const int npad = (int)(kpf - k0);
            for (int e = tid; e < 64 * npad; e += 256) {
                const int r = e / npad, c = e - r * npad;
                if (co0 + r < O) Elem<T>::st(wf + (int64_t)(co0 + r) * kpf + k0 + c, 0.f);
            }
        }
    }
    if (wd) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ci = rq + 16 * i;
            if (ci0 + ci < I && co0 + q4 < O) {
                f32x4 v = {t[q4][ci], t[q4 + 1][ci], t[q4 + 2][ci], t[q4 + 3][ci]};
                store4<T>(wd + (int64_t)(ci0 + ci) * kpd + (int64_t)tap * O + co0 + q4, v);
            }
        }
        if (blockIdx.y == 0 && tap == 0) {
            const int64_t k0 = (int64_t)taps * O;
            const int npad = (int)(kpd - k0);
            for (int e = tid; e < 64 * npad; e += 256) {
                const int r = e / npad, c = e - r * npad;
                if (ci0 + r < I) Elem<T>::st(wd + (int64_t)(ci0 + r) * kpd + k0 + c, 0.f);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// column sums (bias gradient / per-sample rowadd gradient), deterministic 2-stage
// grid: (col blocks of 64*4 columns... see launcher)
// ---------------------------------------------------------------------------
// out[(g*splits + s), c] = sum of rows [s*rps, (s+1)*rps) of group g.  256 threads = 64 columns x 4 row lanes,
// 4 independent loads in flight per lane, LDS combine: coalesced 128/256-byte row segments, fixed order.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void colsum_tile(const TI* __restrict__ a, int64_t lda, TO* __restrict__ out, int64_t ld_out,
                                                   int64_t R, int cols, int splits, int64_t rps, int accumulate) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int gs = blockIdx.y;
    const int g = gs / splits, s = gs - g * splits;
    const int64_t r0 = (int64_t)s * rps;
    int64_t r1 = r0 + rps;
    if (r1 > R) r1 = R;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < cols) {
        const TI* base = a + ((int64_t)g * R) * lda + c;
        int64_t r = r0 + rl;
        for (; r + 12 < r1; r += 16) {
            a0 += Elem<TI>::ld(base + r * lda);
            a1 += Elem<TI>::ld(base + (r + 4) * lda);
            a2 += Elem<TI>::ld(base + (r + 8) * lda);
            a3 += Elem<TI>::ld(base + (r + 12) * lda);
        }
        for (; r < r1; r += 4) a0 += Elem<TI>::ld(base + r * lda);
    }
    red[rl][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rl == 0 && c < cols) {
        float v = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        TO* o = out + (int64_t)gs * ld_out + c;
        if (accumulate) v += Elem<TO>::ld(o);
        Elem<TO>::st(o, v);
    }
}

}  // namespace psg

using namespace psg;

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

const char* psg_last_error(void) { return g_err; }
int psg_version(void) { return 100; }

int psg_conv_init_attrs(void);   // conv_gemm.hip
int psg_wgrad_init_attrs(void);  // wgrad.hip
int psg_attn_init_attrs(void);   // attention.hip
int psg_gn_init_attrs(void);     // groupnorm.hip

int psg_init(int device) {
    PSG_HIP_CHECK(hipSetDevice(device));
    int rc;
    if ((rc = psg_conv_init_attrs()) != PSG_OK) return rc;
    if ((rc = psg_wgrad_init_attrs()) != PSG_OK) return rc;
    if ((rc = psg_attn_init_attrs()) != PSG_OK) return rc;
    if ((rc = psg_gn_init_attrs()) != PSG_OK) return rc;
    return PSG_OK;
}

int psg_noise_add_f32(const float* x0, const float* noise, const int64_t* t, const float* tabA,
                      const float* tabB, float* out, int32_t* flag, int64_t B, int64_t chw, int num_t,
                      int do_clamp, psg_stream_t stream) {
    PSG_REQUIRE(x0 && noise && t && tabA && tabB && out && flag, PSG_ERR_ARG, "noise_add: null pointer");
    PSG_REQUIRE(B >= 0 && chw > 0 && num_t > 0, PSG_ERR_SHAPE, "noise_add: bad shape B=%ld chw=%ld", (long)B, (long)chw);
    if (B == 0) return PSG_OK;
    hipLaunchKernelGGL(noise_add_kernel, dim3(grid_for(B * chw, 256)), dim3(256), 0, (hipStream_t)stream, x0, noise, t,
                       tabA, tabB, out, flag, B, chw, num_t, do_clamp);
    PSG_LAUNCH_CHECK("noise_add");
    return PSG_OK;
}

int psg_noise_fallback_f32(const float* x0, const float* noise, float* out, int32_t* flag, int64_t n,
                           int do_clamp, psg_stream_t stream) {
    PSG_REQUIRE(x0 && noise && out && flag, PSG_ERR_ARG, "noise_fallback: null pointer");
    if (n <= 0) return PSG_OK;
    hipLaunchKernelGGL(noise_fallback_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x0, noise, out,
                       flag, n, do_clamp);
    PSG_LAUNCH_CHECK("noise_fallback");
    return PSG_OK;
}

int psg_ddpm_update_f32(float* x, const float* eps, const float* z, const float* c1, const float* c2,
                        const float* sigma, const int32_t* t_dev, int64_t n, psg_stream_t stream) {
    PSG_REQUIRE(x && eps && z && c1 && c2 && sigma && t_dev, PSG_ERR_ARG, "ddpm_update: null pointer");
    if (n <= 0) return PSG_OK;
    hipLaunchKernelGGL(ddpm_update_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, eps, z, c1, c2,
                       sigma, t_dev, n);
    PSG_LAUNCH_CHECK("ddpm_update");
    return PSG_OK;
}

int psg_sampler_update_f32(float* x, const float* eps, const float* z, int mode, float c0, float c1, float c2, float c3,
                           int64_t n, psg_stream_t stream) {
    PSG_REQUIRE(x && eps, PSG_ERR_ARG, "sampler_update: null pointer");
    PSG_REQUIRE(mode >= 1 && mode <= 3, PSG_ERR_ARG, "sampler_update: mode %d", mode);
    if (n <= 0) return PSG_OK;
    hipLaunchKernelGGL(sampler_update_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, eps, z, mode, c0, c1, c2, c3, n);
    PSG_LAUNCH_CHECK("sampler_update");
    return PSG_OK;
}

__global__ void reparam_kernel(const float* __restrict__ mu, const float* __restrict__ logvar, const float* __restrict__ eps,
                               float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = __fadd_rn(mu[i], __fmul_rn(eps[i], expf(__fmul_rn(0.5f, logvar[i]))));
}

int psg_reparam_f32(const float* mu, const float* logvar, const float* eps, float* out, int64_t n, psg_stream_t stream) {
    PSG_REQUIRE(mu && logvar && eps && out, PSG_ERR_ARG, "reparam: null pointer");
    if (n <= 0) return PSG_OK;
    hipLaunchKernelGGL(reparam_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, mu, logvar, eps, out, n);
    PSG_LAUNCH_CHECK("reparam");
    return PSG_OK;
}

int64_t psg_reduce_workspace_bytes(void) { return (int64_t)RED_BLOCKS * sizeof(float); }

int psg_smooth_l1_f32(const float* pred, const float* target, float* grad, float* loss_out, int32_t* nan_flag,
                      float beta, float grad_scale, int64_t n, void* ws, psg_stream_t stream) {
    PSG_REQUIRE(pred && target && loss_out && ws, PSG_ERR_ARG, "smooth_l1: null pointer");
    PSG_REQUIRE(n > 0 && beta > 0.f, PSG_ERR_SHAPE, "smooth_l1: n=%ld beta=%f", (long)n, beta);
    const int g = grid_for(n, RED_THREADS, RED_BLOCKS);
    hipLaunchKernelGGL(smooth_l1_kernel, dim3(g), dim3(RED_THREADS), 0, (hipStream_t)stream, pred, target, grad,
                       (float*)ws, nan_flag, beta, grad_scale, n);
    PSG_LAUNCH_CHECK("smooth_l1");
    hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, g, loss_out,
                       1.0f / (float)n, 0, nan_flag);
    PSG_LAUNCH_CHECK("smooth_l1_finish");
    return PSG_OK;
}

int psg_sumsq_f32(const float* g, int64_t n, float* out, int accumulate, void* ws, psg_stream_t stream) {
    PSG_REQUIRE(g && out && ws, PSG_ERR_ARG, "sumsq: null pointer");
    PSG_REQUIRE(n > 0, PSG_ERR_SHAPE, "sumsq: n=%ld", (long)n);
    PSG_REQUIRE(aligned16(g), PSG_ERR_ALIGN, "sumsq: g must be 16-byte aligned");
    const int gr = grid_for((n + 3) / 4, RED_THREADS, RED_BLOCKS);
    hipLaunchKernelGGL(sumsq_kernel, dim3(gr), dim3(RED_THREADS), 0, (hipStream_t)stream, g, (float*)ws, n);
    PSG_LAUNCH_CHECK("sumsq");
    hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, gr, out, 1.0f,
                       accumulate, (int32_t*)nullptr);
    PSG_LAUNCH_CHECK("sumsq_finish");
    return PSG_OK;
}

static int adamw_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, float bc1, float bc2_sqrt, const float* normsq, float max_norm,
                        const int32_t* skip_flag, void* shadow_bf16, const AdamSched& sc, psg_stream_t stream) {
    const bool vec = n % 4 == 0 && aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v) && (!shadow_bf16 || aligned8(shadow_bf16));
    PSG_REQUIRE(!shadow_bf16 || vec, PSG_ERR_ALIGN, "adamw: the bf16 shadow needs n %% 4 == 0 and 16-byte aligned buffers");
    if (vec) {
        // grid: 262144 workgroups at most (2-3 trips of the 2 x float4 loop on the 640 M-parameter arena): 3.15 ms against
        // 3.35 ms with 8192 long-running ones (6.1 vs 5.7 TB/s)
        hipLaunchKernelGGL(adamw_kernel4, dim3(grid_for(n / 4, 256, 262144)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n / 4, lr,
                           beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, normsq, max_norm, skip_flag, (bf16_t*)shadow_bf16, sc);
    } else {
        hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr,
                           beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, normsq, max_norm, skip_flag, sc);
    }
    PSG_LAUNCH_CHECK("adamw");
    return PSG_OK;
}

int psg_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step, const float* normsq, float max_norm,
                  const int32_t* skip_flag, void* shadow_bf16, psg_stream_t stream) {
    PSG_REQUIRE(p && g && m && v, PSG_ERR_ARG, "adamw: null pointer");
    PSG_REQUIRE(n > 0 && step >= 1, PSG_ERR_SHAPE, "adamw: n=%ld step=%d", (long)n, step);
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2 = 1.0f - powf(beta2, (float)step);
    const AdamSched sc = {nullptr, nullptr, nullptr, 0};
    return adamw_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), normsq, max_norm, skip_flag, shadow_bf16, sc, stream);
}

int psg_adamw_dev_f32(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_table, const float* beta1_table,
                      int sched_len, float beta1, float beta2, float eps, float weight_decay, int32_t* step_dev, const float* normsq,
                      float max_norm, const int32_t* skip_flag, void* shadow_bf16, psg_stream_t stream) {
    PSG_REQUIRE(p && g && m && v && lr_table && step_dev, PSG_ERR_ARG, "adamw_dev: null pointer");
    PSG_REQUIRE(n > 0 && sched_len >= 1, PSG_ERR_SHAPE, "adamw_dev: n=%ld sched_len=%d", (long)n, sched_len);
    const AdamSched sc = {step_dev, lr_table, beta1_table, sched_len};
    const int rc = adamw_launch(p, g, m, v, n, 0.f, beta1, beta2, eps, weight_decay, 1.f, 1.f, normsq, max_norm, skip_flag, shadow_bf16, sc, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(adam_step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, skip_flag);
    PSG_LAUNCH_CHECK("adam_step_advance");
    return PSG_OK;
}

int psg_clip_scale_f32(float* g, int64_t n, const float* normsq, float max_norm, psg_stream_t stream) {
    PSG_REQUIRE(g && normsq, PSG_ERR_ARG, "clip_scale: null pointer");
    if (n <= 0) return PSG_OK;
    hipLaunchKernelGGL(clip_scale_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, g, n, normsq,
                       max_norm);
    PSG_LAUNCH_CHECK("clip_scale");
    return PSG_OK;
}

#define DISPATCH_DTYPE(dtype, CALL_F32, CALL_BF16)                                        \
    if ((dtype) == PSG_F32) { CALL_F32; }                                                 \
    else if ((dtype) == PSG_BF16) { CALL_BF16; }                                          \
    else return set_error(PSG_ERR_DTYPE, "unsupported dtype %d", (int)(dtype));

int psg_nchw_to_nhwc(const float* src, void* dst, int64_t ld_dst, int B, int C, int HW, int dtype, psg_stream_t stream) {
    PSG_REQUIRE(src && dst, PSG_ERR_ARG, "nchw_to_nhwc: null pointer");
    PSG_REQUIRE(B > 0 && C > 0 && HW > 0 && ld_dst >= C, PSG_ERR_SHAPE, "nchw_to_nhwc: bad shape");
    const int g = grid_for((int64_t)B * C * HW, 256);
    DISPATCH_DTYPE(dtype,
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, src, (float*)dst, ld_dst, B, C, HW),
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, ld_dst, B, C, HW));
    PSG_LAUNCH_CHECK("nchw_to_nhwc");
    return PSG_OK;
}

int psg_nhwc_to_nchw(const void* src, int64_t ld_src, float* dst, int B, int C, int HW, int dtype, psg_stream_t stream) {
    PSG_REQUIRE(src && dst, PSG_ERR_ARG, "nhwc_to_nchw: null pointer");
    PSG_REQUIRE(B > 0 && C > 0 && HW > 0 && ld_src >= C, PSG_ERR_SHAPE, "nhwc_to_nchw: bad shape");
    const int g = grid_for((int64_t)B * C * HW, 256);
    DISPATCH_DTYPE(dtype,
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)src, ld_src, dst, B, C, HW),
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ld_src, dst, B, C, HW));
    PSG_LAUNCH_CHECK("nhwc_to_nchw");
    return PSG_OK;
}

int psg_text_pool(const float* text, void* pooled, int64_t ld_pooled, void* text_cast, int B, int S, int D, int dtype,
                  psg_stream_t stream) {
    PSG_REQUIRE(text && pooled, PSG_ERR_ARG, "text_pool: null pointer");
    PSG_REQUIRE(B > 0 && S > 0 && D > 0 && ld_pooled >= D, PSG_ERR_SHAPE, "text_pool: bad shape");
    DISPATCH_DTYPE(dtype,
        hipLaunchKernelGGL(text_pool_kernel<float>, dim3(B), dim3(256), 0, (hipStream_t)stream, text, (float*)pooled, ld_pooled, (float*)text_cast, B, S, D),
        hipLaunchKernelGGL(text_pool_kernel<bf16_t>, dim3(B), dim3(256), 0, (hipStream_t)stream, text, (bf16_t*)pooled, ld_pooled, (bf16_t*)text_cast, B, S, D));
    PSG_LAUNCH_CHECK("text_pool");
    return PSG_OK;
}

int psg_timestep_sinusoid(const int64_t* t, const float* coeff, void* out, int64_t ld_out, int B, int half, int dtype,
                          psg_stream_t stream) {
    PSG_REQUIRE(t && coeff && out, PSG_ERR_ARG, "sinusoid: null pointer");
    PSG_REQUIRE(B > 0 && half > 0 && ld_out >= 2 * half, PSG_ERR_SHAPE, "sinusoid: bad shape");
    const int g = (B * half + 255) / 256;
    DISPATCH_DTYPE(dtype,
        hipLaunchKernelGGL(sinusoid_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, t, coeff, (float*)out, ld_out, B, half),
        hipLaunchKernelGGL(sinusoid_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, t, coeff, (bf16_t*)out, ld_out, B, half));
    PSG_LAUNCH_CHECK("sinusoid");
    return PSG_OK;
}

int psg_upsample_bilinear_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, int B, int Hi, int Wi, int Ho, int Wo,
                              int C, int dtype, psg_stream_t stream) {
    PSG_REQUIRE(x && y, PSG_ERR_ARG, "upsample_fwd: null pointer");
    PSG_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho >= Hi && Wo >= Wi && C > 0 && (C & 3) == 0 && ldx >= C && ldy >= C &&
                (ldx & 3) == 0 && (ldy & 3) == 0, PSG_ERR_SHAPE, "upsample_fwd: bad shape (C, ld multiples of 4)");
    // bf16 with 16-byte channel chunks and 32-bit offsets: the 8-channel form
    if (dtype == PSG_BF16 && (C & 7) == 0 && (ldx & 7) == 0 && (ldy & 7) == 0 && aligned16(x) && aligned16(y) &&
        (int64_t)B * Ho * Wo * (C / 8) < (1ll << 31) && (int64_t)Hi * Wi * ldx < (1ll << 31) && ldy < (1ll << 24)) {
        const int64_t n8 = (int64_t)B * Ho * Wo * (C / 8);
        hipLaunchKernelGGL(upsample_fwd8_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (int)ldx,
                           (bf16_t*)y, (int)ldy, B, Hi, Wi, Ho, Wo, C);
        PSG_LAUNCH_CHECK("upsample_fwd8");
        return PSG_OK;
    }
    const int g = grid_for((int64_t)B * Ho * Wo * (C / 4), 256, 8192);
    DISPATCH_DTYPE(dtype,
        hipLaunchKernelGGL(upsample_fwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (float*)y, ldy, B, Hi, Wi, Ho, Wo, C),
        hipLaunchKernelGGL(upsample_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, B, Hi, Wi, Ho, Wo, C));
    PSG_LAUNCH_CHECK("upsample_fwd");
    return PSG_OK;
}

int psg_upsample_bilinear_bwd(const void* dy, int64_t lddy, void* dx, int64_t lddx, int B, int Hi, int Wi, int Ho,
                              int Wo, int C, int dtype, psg_stream_t stream) {
    PSG_REQUIRE(dy && dx, PSG_ERR_ARG, "upsample_bwd: null pointer");
    PSG_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho >= Hi && Wo >= Wi && C > 0 && (C & 3) == 0 && lddy >= C && lddx >= C &&
                (lddx & 3) == 0 && (lddy & 3) == 0, PSG_ERR_SHAPE, "upsample_bwd: bad shape (C, ld multiples of 4)");
    if (dtype == PSG_BF16 && (C & 7) == 0 && (lddy & 7) == 0 && (lddx & 7) == 0 && aligned16(dy) && aligned16(dx) &&
        (int64_t)B * Hi * Wi * (C / 8) < (1ll << 31) && (int64_t)Ho * Wo * lddy < (1ll << 31) && lddx < (1ll << 24)) {
        const int64_t n8 = (int64_t)B * Hi * Wi * (C / 8);
        hipLaunchKernelGGL(upsample_bwd8_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (int)lddy,
                           (bf16_t*)dx, (int)lddx, B, Hi, Wi, Ho, Wo, C);
        PSG_LAUNCH_CHECK("upsample_bwd8");
        return PSG_OK;
    }
    const int g = grid_for((int64_t)B * Hi * Wi * (C / 4), 256, 8192);
    DISPATCH_DTYPE(dtype,
        hipLaunchKernelGGL(upsample_bwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)dy, lddy, (float*)dx, lddx, B, Hi, Wi, Ho, Wo, C),
        hipLaunchKernelGGL(upsample_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, B, Hi, Wi, Ho, Wo, C));
    PSG_LAUNCH_CHECK("upsample_bwd");
    return PSG_OK;
}

int psg_add(const void* a, int64_t lda, const void* b, int64_t ldb, void* y, int64_t ldy, int64_t rows, int cols,
            int dtype, psg_stream_t stream) {
    PSG_REQUIRE(a && b && y, PSG_ERR_ARG, "add: null pointer");
    PSG_REQUIRE(rows > 0 && cols > 0 && (cols & 3) == 0 && ((lda | ldb | ldy) & 3) == 0, PSG_ERR_SHAPE, "add: cols/ld must be multiples of 4");
    const int g = grid_for(rows * (cols / 4), 256, 8192);
    DISPATCH_DTYPE(dtype,
        hipLaunchKernelGGL(add_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)a, lda, (const float*)b, ldb, (float*)y, ldy, rows, cols),
        hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)y, ldy, rows, cols));
    PSG_LAUNCH_CHECK("add");
    return PSG_OK;
}

int psg_set_seed_source(const uint64_t* seed_dev) {
    psg::g_seed_source = seed_dev;
    return PSG_OK;
}

int psg_sum_rows(const void* a, int64_t lda, const void* b, int64_t ldb, const void* c, int64_t ldc, void* y, int64_t ldy,
                 int64_t rows, int cols, int dtype, psg_stream_t stream) {
    PSG_REQUIRE(a && y && (b || !c), PSG_ERR_ARG, "sum_rows: null pointer (c needs b)");
    const int CH = dtype == PSG_BF16 ? 8 : 4;
    PSG_REQUIRE(dtype == PSG_F32 || dtype == PSG_BF16, PSG_ERR_DTYPE, "sum_rows: dtype %d", dtype);
    PSG_REQUIRE(rows > 0 && cols > 0 && cols % CH == 0 && lda % CH == 0 && ldy % CH == 0 && (!b || ldb % CH == 0) && (!c || ldc % CH == 0),
                PSG_ERR_SHAPE, "sum_rows: cols / row strides must be multiples of %d", CH);
    PSG_REQUIRE(aligned16(a) && aligned16(y) && (!b || aligned16(b)) && (!c || aligned16(c)), PSG_ERR_ALIGN, "sum_rows: 16-byte alignment");
    const int g = grid_for(rows * (cols / CH), 256, 65536);
    DISPATCH_DTYPE(dtype,
        hipLaunchKernelGGL(sum_rows_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)a, lda, (const float*)b, ldb, (const float*)c, ldc, (float*)y, ldy, rows, cols),
        hipLaunchKernelGGL(sum_rows_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (const bf16_t*)c, ldc, (bf16_t*)y, ldy, rows, cols));
    PSG_LAUNCH_CHECK("sum_rows");
    return PSG_OK;
}

int psg_dropout_apply(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int cols, float p, uint64_t seed,
                      float scale, int dtype, psg_stream_t stream) {
    PSG_REQUIRE(x && y, PSG_ERR_ARG, "dropout_apply: null pointer");
    PSG_REQUIRE(rows > 0 && cols > 0 && (cols & 3) == 0 && ((ldx | ldy) & 3) == 0 && p >= 0.f && p < 1.f, PSG_ERR_SHAPE, "dropout_apply: bad shape");
    const int g = grid_for(rows * (cols / 4), 256, 8192);
    const uint32_t th = drop_thresh(p);
    DISPATCH_DTYPE(dtype,
        hipLaunchKernelGGL(dropout_apply_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (float*)y, ldy, rows, cols, th, seed, scale, seed_source()),
        hipLaunchKernelGGL(dropout_apply_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, rows, cols, th, seed, scale, seed_source()));
    PSG_LAUNCH_CHECK("dropout_apply");
    return PSG_OK;
}

int psg_epilogue_bwd(const void* dy, int64_t lddy, const void* u, int64_t ldu, void* g, int64_t ldg, int64_t rows, int cols,
                     int act, float alpha, float drop_p, uint64_t seed, int dtype, psg_stream_t stream) {
    PSG_REQUIRE(dy && g && (u || act == PSG_ACT_NONE), PSG_ERR_ARG, "epilogue_bwd: null pointer");
    PSG_REQUIRE(rows > 0 && cols > 0 && (cols & 3) == 0 && ((lddy | ldg) & 3) == 0 && (!u || (ldu & 3) == 0) && drop_p >= 0.f && drop_p < 1.f,
                PSG_ERR_SHAPE, "epilogue_bwd: bad shape");
    const int gr = grid_for(rows * (cols / 4), 256, 8192);
    const uint32_t th = drop_p > 0.f ? drop_thresh(drop_p) : 0u;
    const float ds = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    DISPATCH_DTYPE(dtype,
        hipLaunchKernelGGL(epilogue_bwd_kernel<float>, dim3(gr), dim3(256), 0, (hipStream_t)stream, (const float*)dy, lddy, (const float*)u, ldu, (float*)g, ldg, rows, cols, act, alpha, th, ds, seed, seed_source()),
        hipLaunchKernelGGL(epilogue_bwd_kernel<bf16_t>, dim3(gr), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, lddy, (const bf16_t*)u, ldu, (bf16_t*)g, ldg, rows, cols, act, alpha, th, ds, seed, seed_source()));
    PSG_LAUNCH_CHECK("epilogue_bwd");
    return PSG_OK;
}

int64_t psg_kpad(int64_t K, int dtype) {
    const int64_t bk = dtype == PSG_BF16 ? 64 : 32;
    return (K + bk - 1) / bk * bk;
}

int psg_prep_weight(const void* w, int w_dtype, int w_layout, void* wf, void* wd, int O, int I, int ksize, int dtype, psg_stream_t stream) {
    PSG_REQUIRE(w && (wf || wd), PSG_ERR_ARG, "prep_weight: null pointer");
    PSG_REQUIRE(O > 0 && I > 0 && (ksize == 1 || ksize == 3 || ksize == 4), PSG_ERR_SHAPE, "prep_weight: O=%d I=%d k=%d", O, I, ksize);
    PSG_REQUIRE(ksize != 4 || (w_layout == PSG_W_OHWI && O % 4 == 0 && I % 4 == 0 && aligned16(w)), PSG_ERR_ARG,
                "prep_weight: 4x4 weights (VAE encoder) must be OHWI, 16-byte aligned, O and I multiples of 4");
    PSG_REQUIRE(w_layout == PSG_W_OIHW || w_layout == PSG_W_OHWI, PSG_ERR_ARG, "prep_weight: w_layout %d", w_layout);
    PSG_REQUIRE(w_dtype == PSG_F32 || w_dtype == PSG_BF16, PSG_ERR_DTYPE, "prep_weight: w_dtype %d", w_dtype);
    const int ohwi = w_layout == PSG_W_OHWI;
    const int taps = ksize * ksize;
    const int64_t kpf = psg_kpad((int64_t)taps * I, dtype), kpd = psg_kpad((int64_t)taps * O, dtype);
    PSG_REQUIRE((O + 31) / 32 <= 65535, PSG_ERR_SHAPE, "prep_weight: O=%d too large", O);
    hipStream_t s = (hipStream_t)stream;
    if ((ohwi || taps == 1) && O % 4 == 0 && I % 4 == 0 && aligned16(w) && (!wf || aligned16(wf)) && (!wd || aligned16(wd))) {
        const dim3 g((I + 63) / 64, (O + 63) / 64, taps);
        if (w_dtype == PSG_BF16) {
            PSG_REQUIRE(dtype == PSG_BF16, PSG_ERR_DTYPE, "prep_weight: a bf16 source prepares bf16 weights only");
            if (!wf && O % 8 == 0 && I % 8 == 0)
                hipLaunchKernelGGL(prep_wd_bf16_kernel, g, dim3(256), 0, s, (const bf16_t*)w, (bf16_t*)wd, O, I, taps, kpd);
            else
                hipLaunchKernelGGL((prep_weight_ohwi_kernel<bf16_t, bf16_t>), g, dim3(256), 0, s, (const bf16_t*)w, (bf16_t*)wf, (bf16_t*)wd, O, I, taps, kpf, kpd);
        } else {
            DISPATCH_DTYPE(dtype,
                hipLaunchKernelGGL((prep_weight_ohwi_kernel<float, float>), g, dim3(256), 0, s, (const float*)w, (float*)wf, (float*)wd, O, I, taps, kpf, kpd),
                hipLaunchKernelGGL((prep_weight_ohwi_kernel<float, bf16_t>), g, dim3(256), 0, s, (const float*)w, (bf16_t*)wf, (bf16_t*)wd, O, I, taps, kpf, kpd));
        }
        PSG_LAUNCH_CHECK("prep_weight");
        return PSG_OK;
    }
    PSG_REQUIRE(w_dtype == PSG_F32, PSG_ERR_ARG, "prep_weight: a bf16 source must be OHWI (or 1x1), 16-byte aligned, O and I multiples of 4");
    const dim3 g((I + 31) / 32, (O + 31) / 32);
    DISPATCH_DTYPE(dtype,
        hipLaunchKernelGGL(prep_weight_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)w, ohwi, (float*)wf, (float*)wd, O, I, taps, kpf, kpd),
        hipLaunchKernelGGL(prep_weight_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const float*)w, ohwi, (bf16_t*)wf, (bf16_t*)wd, O, I, taps, kpf, kpd));
    PSG_LAUNCH_CHECK("prep_weight");
    return PSG_OK;
}

static inline void colsum_plan(int64_t R, int groups, int cols, int& splits, int64_t& rps) {
    const int64_t cb = (cols + 63) / 64;
    int64_t want = 2048 / (cb * groups);          // ~8 workgroups per CU in flight
    const int64_t maxs = (R + 63) / 64;           // at least 64 rows per split
    if (want > maxs) want = maxs;
    if (want > 256) want = 256;
    if (want < 1) want = 1;
    rps = (R + want - 1) / want;
    splits = (int)((R + rps - 1) / rps);
}

int64_t psg_colsum_workspace_bytes(int64_t R, int groups, int cols) {
    int splits; int64_t rps;
    colsum_plan(R, groups, cols, splits, rps);
    return (int64_t)groups * splits * cols * sizeof(float);
}

int psg_colsum(const void* a, int64_t lda, void* out, int64_t ld_out, int64_t R, int groups, int cols, int dtype,
               int out_dtype, int accumulate, void* ws, int64_t ws_bytes, psg_stream_t stream) {
    PSG_REQUIRE(a && out && ws, PSG_ERR_ARG, "colsum: null pointer");
    PSG_REQUIRE(R > 0 && groups > 0 && cols > 0 && lda >= cols && ld_out >= cols, PSG_ERR_SHAPE, "colsum: bad shape");
    PSG_REQUIRE(ws_bytes >= psg_colsum_workspace_bytes(R, groups, cols), PSG_ERR_WORKSPACE, "colsum: workspace too small");
    PSG_REQUIRE(!(accumulate && out_dtype != PSG_F32), PSG_ERR_ARG, "colsum: accumulate needs fp32 out");
    PSG_REQUIRE(dtype == PSG_F32 || dtype == PSG_BF16, PSG_ERR_DTYPE, "colsum: dtype %d", dtype);
    PSG_REQUIRE(out_dtype == PSG_F32 || out_dtype == PSG_BF16, PSG_ERR_DTYPE, "colsum: out_dtype %d", out_dtype);
    int splits; int64_t rps;
    colsum_plan(R, groups, cols, splits, rps);
    hipStream_t st = (hipStream_t)stream;
    const int cb = (cols + 63) / 64;
    float* wsf = (float*)ws;
    if (splits == 1) {      // single pass straight into the output
        dim3 g1(cb, groups);
        if (dtype == PSG_F32 && out_dtype == PSG_F32) hipLaunchKernelGGL((colsum_tile<float, float>), g1, dim3(256), 0, st, (const float*)a, lda, (float*)out, ld_out, R, cols, 1, R, accumulate);
        else if (dtype == PSG_F32) hipLaunchKernelGGL((colsum_tile<float, bf16_t>), g1, dim3(256), 0, st, (const float*)a, lda, (bf16_t*)out, ld_out, R, cols, 1, R, 0);
        else if (out_dtype == PSG_F32) hipLaunchKernelGGL((colsum_tile<bf16_t, float>), g1, dim3(256), 0, st, (const bf16_t*)a, lda, (float*)out, ld_out, R, cols, 1, R, accumulate);
        else hipLaunchKernelGGL((colsum_tile<bf16_t, bf16_t>), g1, dim3(256), 0, st, (const bf16_t*)a, lda, (bf16_t*)out, ld_out, R, cols, 1, R, 0);
        PSG_LAUNCH_CHECK("colsum");
        return PSG_OK;
    }
    dim3 g1(cb, groups * splits);
    if (dtype == PSG_F32) hipLaunchKernelGGL((colsum_tile<float, float>), g1, dim3(256), 0, st, (const float*)a, lda, wsf, (int64_t)cols, R, cols, splits, rps, 0);
    else hipLaunchKernelGGL((colsum_tile<bf16_t, float>), g1, dim3(256), 0, st, (const bf16_t*)a, lda, wsf, (int64_t)cols, R, cols, splits, rps, 0);
    PSG_LAUNCH_CHECK("colsum_stage1");
    dim3 g2(cb, groups);    // second stage: the partial slab [groups][splits][cols] is itself a column-sum problem
    if (out_dtype == PSG_F32) hipLaunchKernelGGL((colsum_tile<float, float>), g2, dim3(256), 0, st, (const float*)wsf, (int64_t)cols, (float*)out, ld_out, (int64_t)splits, cols, 1, (int64_t)splits, accumulate);
    else hipLaunchKernelGGL((colsum_tile<float, bf16_t>), g2, dim3(256), 0, st, (const float*)wsf, (int64_t)cols, (bf16_t*)out, ld_out, (int64_t)splits, cols, 1, (int64_t)splits, 0);
    PSG_LAUNCH_CHECK("colsum_stage2");
    return PSG_OK;
}

}  // extern "C"
