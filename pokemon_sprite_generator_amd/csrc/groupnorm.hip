// GroupNorm (+ fused SiLU) forward / backward, channels-last [B, HW, C].
// One workgroup per (sample, group): the group's HW x Cg slab (<= 58 KB fp32 at
// the U-Net's shapes) is staged ONCE into LDS, statistics are wave-shuffle /
// LDS reductions, and the normalised output is written from LDS, so HBM traffic
// is the algorithmic minimum: 1 read + 1 write per element forward, 2 reads +
// 1 write backward.  Roofline: HBM.
#include "psg_common.h"

namespace psg {

constexpr int GN_THREADS = 256;
constexpr int GN_LDS_ELEMS = 16384;  // slab cap held in LDS (elements of T)

// element e in [0, HW*Cg) -> (pixel, channel-in-group); Cg even, 2 elements per access
template <typename T>
__device__ __forceinline__ void ld2(const T* p, float& a, float& b);
template <> __device__ __forceinline__ void ld2<float>(const float* p, float& a, float& b) {
    float2 v = *reinterpret_cast<const float2*>(p); a = v.x; b = v.y;
}
template <> __device__ __forceinline__ void ld2<bf16_t>(const bf16_t* p, float& a, float& b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v = *reinterpret_cast<const bf16x2*>(p); a = (float)v[0]; b = (float)v[1];
}
template <typename T>
__device__ __forceinline__ void st2(T* p, float a, float b);
template <> __device__ __forceinline__ void st2<float>(float* p, float a, float b) {
    *reinterpret_cast<float2*>(p) = make_float2(a, b);
}
template <> __device__ __forceinline__ void st2<bf16_t>(bf16_t* p, float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v = {(bf16_t)a, (bf16_t)b};
    *reinterpret_cast<bf16x2*>(p) = v;
}

template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_fwd_kernel(
    const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ mean_out, float* __restrict__ rstd_out, int HW, int C,
    int G, float eps, int silu) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* slab = reinterpret_cast<T*>(smem_raw);
    __shared__ float red[16];
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const int Cg = C / G, half = Cg >> 1;
    const int npair = HW * half;
    const bool in_lds = (HW * Cg) <= GN_LDS_ELEMS;
    const float inv_half = 1.0f / (float)half;
    const T* xb = x + (int64_t)b * HW * ldx + g * Cg;
    T* yb = y + (int64_t)b * HW * ldy + g * Cg;

    // pass 1: load (+stage), sum
    float s = 0.f;
    for (int e = threadIdx.x; e < npair; e += GN_THREADS) {
        const int p = fastdiv(e, half, inv_half), cv = e - p * half;
        float a, c;
        ld2<T>(xb + (int64_t)p * ldx + 2 * cv, a, c);
        if (in_lds) st2<T>(slab + 2 * e, a, c);
        s += a + c;
    }
    const float n = (float)(HW * Cg);
    const float mean = block_sum(s, red) / n;
    // pass 2: variance about the mean (biased)
    float q = 0.f;
    for (int e = threadIdx.x; e < npair; e += GN_THREADS) {
        float a, c;
        if (in_lds) ld2<T>(slab + 2 * e, a, c);
        else { const int p = fastdiv(e, half, inv_half), cv = e - p * half; ld2<T>(xb + (int64_t)p * ldx + 2 * cv, a, c); }
        a -= mean; c -= mean;
        q += a * a + c * c;
    }
    const float var = block_sum(q, red) / n;
    const float rstd = rsqrtf(var + eps);
    if (threadIdx.x == 0) { mean_out[blockIdx.x] = mean; rstd_out[blockIdx.x] = rstd; }
    // pass 3: normalise, affine, SiLU
    for (int e = threadIdx.x; e < npair; e += GN_THREADS) {
        const int p = fastdiv(e, half, inv_half), cv = e - p * half;
        float a, c;
        if (in_lds) ld2<T>(slab + 2 * e, a, c);
        else ld2<T>(xb + (int64_t)p * ldx + 2 * cv, a, c);
        const int ch = g * Cg + 2 * cv;
        a = (a - mean) * rstd * gamma[ch] + beta[ch];
        c = (c - mean) * rstd * gamma[ch + 1] + beta[ch + 1];
        if (silu) { a = silu_f(a); c = silu_f(c); }
        st2<T>(yb + (int64_t)p * ldy + 2 * cv, a, c);
    }
}

// backward: dx, and per-(b, channel) partials of dgamma / dbeta into ws[2][B][C]
template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_bwd_kernel(
    const T* __restrict__ dy, int64_t lddy, const T* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    T* __restrict__ dx, int64_t lddx, float* __restrict__ ws, int B, int HW, int C, int G, int silu) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ float red[16];
    __shared__ float chan_part[2][GN_THREADS];
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const int Cg = C / G, half = Cg >> 1;
    const int npair = HW * half;
    const int nel = HW * Cg;
    const bool in_lds = nel <= GN_LDS_ELEMS;
    T* sx = reinterpret_cast<T*>(smem_raw);
    T* sd = sx + (in_lds ? ((nel + 7) & ~7) : 0);
    const float inv_half = 1.0f / (float)half;
    const T* xb = x + (int64_t)b * HW * ldx + g * Cg;
    const T* dyb = dy + (int64_t)b * HW * lddy + g * Cg;
    T* dxb = dx + (int64_t)b * HW * lddx + g * Cg;
    const float mean = mean_in[blockIdx.x], rstd = rstd_in[blockIdx.x];

    // pass A: stage x, dy; s1 = sum(dxhat), s2 = sum(dxhat * xhat)
    float s1 = 0.f, s2 = 0.f;
    for (int e = threadIdx.x; e < npair; e += GN_THREADS) {
        const int p = fastdiv(e, half, inv_half), cv = e - p * half;
        float xa, xc, da, dc;
        ld2<T>(xb + (int64_t)p * ldx + 2 * cv, xa, xc);
        ld2<T>(dyb + (int64_t)p * lddy + 2 * cv, da, dc);
        if (in_lds) { st2<T>(sx + 2 * e, xa, xc); st2<T>(sd + 2 * e, da, dc); }
        const int ch = g * Cg + 2 * cv;
        const float ga = gamma[ch], gc = gamma[ch + 1];
        const float ha = (xa - mean) * rstd, hc = (xc - mean) * rstd;
        if (silu) { da *= silu_grad(ha * ga + beta[ch]); dc *= silu_grad(hc * gc + beta[ch + 1]); }
        s1 += da * ga + dc * gc;
        s2 += da * ga * ha + dc * gc * hc;
    }
    const float n = (float)nel;
    s1 = block_sum(s1, red) / n;
    s2 = block_sum(s2, red) / n;

    // pass B: dx = rstd * (dxhat - s1 - xhat * s2)
    for (int e = threadIdx.x; e < npair; e += GN_THREADS) {
        const int p = fastdiv(e, half, inv_half), cv = e - p * half;
        float xa, xc, da, dc;
        if (in_lds) { ld2<T>(sx + 2 * e, xa, xc); ld2<T>(sd + 2 * e, da, dc); }
        else { ld2<T>(xb + (int64_t)p * ldx + 2 * cv, xa, xc); ld2<T>(dyb + (int64_t)p * lddy + 2 * cv, da, dc); }
        const int ch = g * Cg + 2 * cv;
        const float ga = gamma[ch], gc = gamma[ch + 1];
        const float ha = (xa - mean) * rstd, hc = (xc - mean) * rstd;
        if (silu) { da *= silu_grad(ha * ga + beta[ch]); dc *= silu_grad(hc * gc + beta[ch + 1]); }
        const float ra = rstd * (da * ga - s1 - ha * s2);
        const float rc = rstd * (dc * gc - s1 - hc * s2);
        st2<T>(dxb + (int64_t)p * lddx + 2 * cv, ra, rc);   // dx may alias dy: this element was consumed above
    }

    // pass C: per-channel sums over HW (deterministic): P threads per channel, then combine
    if (Cg <= GN_THREADS) {
        const int P = GN_THREADS / Cg;               // partial sums per channel
        const int c = threadIdx.x % Cg, part = threadIdx.x / Cg;
        float ag = 0.f, ab = 0.f;
        if (part < P) {
            const int ch = g * Cg + c;
            const float gch = gamma[ch], bch = beta[ch];
            for (int p = part; p < HW; p += P) {
                float xv, dv;
                if (in_lds) { xv = Elem<T>::ld(sx + p * Cg + c); dv = Elem<T>::ld(sd + p * Cg + c); }
                else { xv = Elem<T>::ld(xb + (int64_t)p * ldx + c); dv = Elem<T>::ld(dyb + (int64_t)p * lddy + c); }
                const float h = (xv - mean) * rstd;
                if (silu) dv *= silu_grad(h * gch + bch);
                ag += dv * h;
                ab += dv;
            }
        }
        __syncthreads();
        chan_part[0][threadIdx.x] = ag;
        chan_part[1][threadIdx.x] = ab;
        __syncthreads();
        if (threadIdx.x < Cg) {
            float tg = 0.f, tb = 0.f;
            for (int q = 0; q < P; ++q) { tg += chan_part[0][q * Cg + threadIdx.x]; tb += chan_part[1][q * Cg + threadIdx.x]; }
            const int ch = g * Cg + threadIdx.x;
            ws[(int64_t)b * C + ch] = tg;
            ws[(int64_t)B * C + (int64_t)b * C + ch] = tb;
        }
    }
}

// NOTE (aliasing): when dx aliases dy and the slab is NOT in LDS, pass C would read dy after pass B
// overwrote it; the launcher forbids aliasing in that case.

// dgamma/dbeta[c] (+)= sum_b ws[{0,1}][b][c]: 64 columns x 4 batch lanes per block, fixed order
__global__ __launch_bounds__(256) void gn_param_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int B, int C, int accumulate) {
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float ag = 0.f, ab = 0.f;
    if (c < C)
        for (int b = rl; b < B; b += 4) { ag += ws[(int64_t)b * C + c]; ab += ws[(int64_t)B * C + (int64_t)b * C + c]; }
    red[0][rl][cl] = ag; red[1][rl][cl] = ab;
    __syncthreads();
    if (rl == 0 && c < C) {
        ag = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        ab = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
        if (accumulate) { ag += dgamma[c]; ab += dbeta[c]; }
        dgamma[c] = ag;
        dbeta[c] = ab;
    }
}

}  // namespace psg
using namespace psg;

extern "C" {

int psg_gn_init_attrs(void) {
    const int big = 2 * GN_LDS_ELEMS * 4 + 64;
#define SET_LDS(K) PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, big))
    SET_LDS(gn_fwd_kernel<float>); SET_LDS(gn_fwd_kernel<bf16_t>);
    SET_LDS(gn_bwd_kernel<float>); SET_LDS(gn_bwd_kernel<bf16_t>);
#undef SET_LDS
    return PSG_OK;
}

static int gn_check(int B, int HW, int C, int G, int64_t lda, int64_t ldb) {
    PSG_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, PSG_ERR_SHAPE, "groupnorm: B=%d HW=%d C=%d G=%d", B, HW, C, G);
    PSG_REQUIRE(((C / G) & 1) == 0, PSG_ERR_SHAPE, "groupnorm: channels per group (%d) must be even", C / G);
    PSG_REQUIRE(C / G <= GN_THREADS, PSG_ERR_SHAPE, "groupnorm: channels per group (%d) > %d", C / G, GN_THREADS);
    PSG_REQUIRE(lda >= C && ldb >= C && (lda & 1) == 0 && (ldb & 1) == 0, PSG_ERR_SHAPE, "groupnorm: row strides must be even and >= C");
    PSG_REQUIRE((int64_t)HW * (C / G) < (1 << 24), PSG_ERR_SHAPE, "groupnorm: slab too large");
    return PSG_OK;
}

int psg_groupnorm_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma, const float* beta,
                      float* mean, float* rstd, int B, int HW, int C, int G, float eps, int silu, int dtype,
                      psg_stream_t stream) {
    PSG_REQUIRE(x && y && gamma && beta && mean && rstd, PSG_ERR_ARG, "groupnorm_fwd: null pointer");
    int rc = gn_check(B, HW, C, G, ldx, ldy);
    if (rc) return rc;
    const int nel = HW * (C / G);
    const size_t esz = dtype == PSG_BF16 ? 2 : 4;
    const size_t lds = nel <= GN_LDS_ELEMS ? (size_t)nel * esz : 0;
    ProfScope prof(PROF_GN, 2.0 * (double)B * HW * C * (double)esz, (hipStream_t)stream);
    if (dtype == PSG_F32)
        hipLaunchKernelGGL(gn_fwd_kernel<float>, dim3(B * G), dim3(GN_THREADS), lds, (hipStream_t)stream, (const float*)x, ldx,
                           (float*)y, ldy, gamma, beta, mean, rstd, HW, C, G, eps, silu);
    else if (dtype == PSG_BF16)
        hipLaunchKernelGGL(gn_fwd_kernel<bf16_t>, dim3(B * G), dim3(GN_THREADS), lds, (hipStream_t)stream, (const bf16_t*)x, ldx,
                           (bf16_t*)y, ldy, gamma, beta, mean, rstd, HW, C, G, eps, silu);
    else return set_error(PSG_ERR_DTYPE, "groupnorm_fwd: dtype %d", dtype);
    PSG_LAUNCH_CHECK("groupnorm_fwd");
    return PSG_OK;
}

int64_t psg_groupnorm_bwd_workspace_bytes(int B, int C) { return (int64_t)2 * B * C * sizeof(float); }

int psg_groupnorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma, const float* beta,
                      const float* mean, const float* rstd, void* dx, int64_t lddx, float* dgamma, float* dbeta, int B,
                      int HW, int C, int G, int silu, int accumulate, int dtype, void* ws, psg_stream_t stream) {
    PSG_REQUIRE(dy && x && gamma && beta && mean && rstd && dx && dgamma && dbeta && ws, PSG_ERR_ARG, "groupnorm_bwd: null pointer");
    int rc = gn_check(B, HW, C, G, ldx, lddy);
    if (rc) return rc;
    PSG_REQUIRE(lddx >= C && (lddx & 1) == 0, PSG_ERR_SHAPE, "groupnorm_bwd: lddx");
    const int nel = HW * (C / G);
    const size_t esz = dtype == PSG_BF16 ? 2 : 4;
    const bool in_lds = nel <= GN_LDS_ELEMS;
    PSG_REQUIRE(in_lds || dx != dy, PSG_ERR_ARG, "groupnorm_bwd: dx may alias dy only when the slab fits LDS");
    const size_t lds = in_lds ? (size_t)(((nel + 7) & ~7) + nel) * esz : 0;
    ProfScope prof(PROF_GN, 3.0 * (double)B * HW * C * (double)esz, (hipStream_t)stream);
    if (dtype == PSG_F32)
        hipLaunchKernelGGL(gn_bwd_kernel<float>, dim3(B * G), dim3(GN_THREADS), lds, (hipStream_t)stream, (const float*)dy, lddy,
                           (const float*)x, ldx, gamma, beta, mean, rstd, (float*)dx, lddx, (float*)ws, B, HW, C, G, silu);
    else if (dtype == PSG_BF16)
        hipLaunchKernelGGL(gn_bwd_kernel<bf16_t>, dim3(B * G), dim3(GN_THREADS), lds, (hipStream_t)stream, (const bf16_t*)dy, lddy,
                           (const bf16_t*)x, ldx, gamma, beta, mean, rstd, (bf16_t*)dx, lddx, (float*)ws, B, HW, C, G, silu);
    else return set_error(PSG_ERR_DTYPE, "groupnorm_bwd: dtype %d", dtype);
    PSG_LAUNCH_CHECK("groupnorm_bwd");
    hipLaunchKernelGGL(gn_param_reduce_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const float*)ws, dgamma,
                       dbeta, B, C, accumulate);
    PSG_LAUNCH_CHECK("groupnorm_param_reduce");
    return PSG_OK;
}

}  // extern "C"
