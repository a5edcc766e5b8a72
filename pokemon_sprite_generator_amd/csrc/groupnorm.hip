// GroupNorm (+ fused SiLU) forward / backward, channels-last [B, HW, C].  Roofline: HBM.
//
// Every pass streams whole pixel rows with 16-byte vectors per lane (a thread owns one fixed
// 16-byte channel chunk and walks pixels), so loads/stores are fully coalesced for every
// channels-per-group value (10, 20, 40, 80: a chunk may straddle two groups - handled per
// element).  Work is split over (sample, pixel-split) workgroups to fill the chip:
//   forward : stats kernel -> per-(b, split, group) partial (sum, sum of squares)
//             apply kernel -> combines the partials (double), writes mean/rstd, normalises
//   backward: reduce kernel -> per-(b, split, channel) partial (sum dz*xhat, sum dz)
//             apply kernel  -> combines them per sample, forms the two group sums, writes dx
//             param kernel  -> dgamma/dbeta over the batch (fixed order)
// All reductions are fixed-order (LDS trees, no float atomics): results are run-to-run identical.
// Traffic: x read twice + y written (forward), x and dy read twice + dx written (backward); the
// second read of a <=1 MB per-sample slab is served from L2 / Infinity Cache.
#include "psg_common.h"

namespace psg {

constexpr int GN_MAXSPLIT = 16;
#ifndef GN_U_VALUE
#define GN_U_VALUE 4
#endif
#ifndef GN_UB_VALUE
#define GN_UB_VALUE 2
#endif
constexpr int GN_U = GN_U_VALUE;     // loads in flight per lane, forward passes
constexpr int GN_UB = GN_UB_VALUE;   // backward passes (two operands each)

struct GnP {
    const void *x, *dy, *dres; void *y, *dx;
    const float *gamma, *beta; float *mean, *rstd, *ws, *dgamma, *dbeta;
    int64_t ldx, ldy, lddy, lddx, lddres;
    int B, HW, C, G, Cg, CC, PP, NS, pps;   // CC chunks per row, PP pixel lanes, NS splits, pps pixels per split
    float eps; int silu, accumulate;
};

template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void ld(const float* p, float* v) { f32x4 t = *reinterpret_cast<const f32x4*>(p); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
    static __device__ __forceinline__ void st(float* p, const float* v) { f32x4 t = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(p) = t; }
};
template <> struct Vec<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void ld(const bf16_t* p, float* v) {
        bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float* v) {
        bf16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (bf16_t)v[e];
        *reinterpret_cast<bf16x8*>(p) = t;
    }
};

// ---------------------------------------------------------------- forward: partial statistics
template <typename T>
__global__ void gn_stats_kernel(const GnP p) {
    constexpr int N = Vec<T>::N;
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [PP][C][2]
    const int b = blockIdx.x / p.NS, split = blockIdx.x - b * p.NS;
    const int c = threadIdx.x % p.CC, pl = threadIdx.x / p.CC;
    const int p0 = split * p.pps, p1 = min(p.HW, p0 + p.pps);
    const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.HW * p.ldx + c * N;
    float s[N], q[N];
#pragma unroll
    for (int e = 0; e < N; ++e) { s[e] = 0.f; q[e] = 0.f; }
    for (int px = p0 + pl; px < p1; px += GN_U * p.PP) {      // GN_U independent 16-byte loads in flight per lane
        float v[GN_U][N];
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            const int pu = px + u * p.PP;
            Vec<T>::ld(xb + (int64_t)(pu < p1 ? pu : px) * p.ldx, v[u]);
        }
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            const float w = (px + u * p.PP) < p1 ? 1.f : 0.f;
#pragma unroll
            for (int e = 0; e < N; ++e) { s[e] += w * v[u][e]; q[e] += w * v[u][e] * v[u][e]; }
        }
    }
    float* row = sm + (int64_t)pl * p.C * 2;
#pragma unroll
    for (int e = 0; e < N; ++e) { row[(c * N + e) * 2] = s[e]; row[(c * N + e) * 2 + 1] = q[e]; }
    __syncthreads();
    if (threadIdx.x < p.G) {
        const int g = threadIdx.x;
        float a0 = 0.f, a1 = 0.f;
        for (int l = 0; l < p.PP; ++l)
            for (int ch = g * p.Cg; ch < (g + 1) * p.Cg; ++ch) { a0 += sm[((int64_t)l * p.C + ch) * 2]; a1 += sm[((int64_t)l * p.C + ch) * 2 + 1]; }
        float* o = p.ws + ((int64_t)blockIdx.x * p.G + g) * 2;
        o[0] = a0; o[1] = a1;
    }
}

// ---------------------------------------------------------------- forward: apply
template <typename T>
__global__ void gn_apply_kernel(const GnP p) {
    constexpr int N = Vec<T>::N;
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [G][2] mean, rstd
    const int b = blockIdx.x / p.NS, split = blockIdx.x - b * p.NS;
    if (threadIdx.x < p.G) {
        const int g = threadIdx.x;
        double a0 = 0.0, a1 = 0.0;
        for (int s2 = 0; s2 < p.NS; ++s2) {
            const float* o = p.ws + ((int64_t)(b * p.NS + s2) * p.G + g) * 2;
            a0 += (double)o[0]; a1 += (double)o[1];
        }
        const double n = (double)p.HW * p.Cg;
        const double mean = a0 / n;
        double var = a1 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
        sm[g * 2] = (float)mean; sm[g * 2 + 1] = rstd;
        if (split == 0) { p.mean[b * p.G + g] = (float)mean; p.rstd[b * p.G + g] = rstd; }
    }
    __syncthreads();
    const int c = threadIdx.x % p.CC, pl = threadIdx.x / p.CC;
    float sc[N], sh[N];
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const int ch = c * N + e, g = ch / p.Cg;
        const float rs = sm[g * 2 + 1] * p.gamma[ch];
        sc[e] = rs; sh[e] = p.beta[ch] - sm[g * 2] * rs;
    }
    const int p0 = split * p.pps, p1 = min(p.HW, p0 + p.pps);
    const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.HW * p.ldx + c * N;
    T* yb = reinterpret_cast<T*>(p.y) + (int64_t)b * p.HW * p.ldy + c * N;
    for (int px = p0 + pl; px < p1; px += GN_U * p.PP) {
        float v[GN_U][N];
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            const int pu = px + u * p.PP;
            Vec<T>::ld(xb + (int64_t)(pu < p1 ? pu : px) * p.ldx, v[u]);
        }
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            const int pu = px + u * p.PP;
#pragma unroll
            for (int e = 0; e < N; ++e) { v[u][e] = v[u][e] * sc[e] + sh[e]; if (p.silu) v[u][e] = silu_f(v[u][e]); }
            if (pu < p1) Vec<T>::st(yb + (int64_t)pu * p.ldy, v[u]);
        }
    }
}

// ---------------------------------------------------------------- backward: per-channel partials
template <typename T>
__global__ void gn_bwd_reduce_kernel(const GnP p) {
    constexpr int N = Vec<T>::N;
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [PP][C][2]
    const int b = blockIdx.x / p.NS, split = blockIdx.x - b * p.NS;
    const int c = threadIdx.x % p.CC, pl = threadIdx.x / p.CC;
    float mu[N], rs[N], ga[N], be[N], a0[N], a1[N];
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const int ch = c * N + e, g = ch / p.Cg;
        mu[e] = p.mean[b * p.G + g]; rs[e] = p.rstd[b * p.G + g]; ga[e] = p.gamma[ch]; be[e] = p.beta[ch];
        a0[e] = 0.f; a1[e] = 0.f;
    }
    const int p0 = split * p.pps, p1 = min(p.HW, p0 + p.pps);
    const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.HW * p.ldx + c * N;
    const T* gb = reinterpret_cast<const T*>(p.dy) + (int64_t)b * p.HW * p.lddy + c * N;
    for (int px = p0 + pl; px < p1; px += GN_UB * p.PP) {
        float v[GN_UB][N], d[GN_UB][N];
#pragma unroll
        for (int u = 0; u < GN_UB; ++u) {
            const int pu = px + u * p.PP, ps = pu < p1 ? pu : px;
            Vec<T>::ld(xb + (int64_t)ps * p.ldx, v[u]);
            Vec<T>::ld(gb + (int64_t)ps * p.lddy, d[u]);
        }
#pragma unroll
        for (int u = 0; u < GN_UB; ++u) {
            const float w = (px + u * p.PP) < p1 ? 1.f : 0.f;
#pragma unroll
            for (int e = 0; e < N; ++e) {
                const float h = (v[u][e] - mu[e]) * rs[e];
                float dz = d[u][e] * w;
                if (p.silu) dz *= silu_grad(h * ga[e] + be[e]);
                a0[e] += dz * h; a1[e] += dz;
            }
        }
    }
    float* row = sm + (int64_t)pl * p.C * 2;
#pragma unroll
    for (int e = 0; e < N; ++e) { row[(c * N + e) * 2] = a0[e]; row[(c * N + e) * 2 + 1] = a1[e]; }
    __syncthreads();
    for (int ch = threadIdx.x; ch < p.C; ch += blockDim.x) {
        float t0 = 0.f, t1 = 0.f;
        for (int l = 0; l < p.PP; ++l) { t0 += sm[((int64_t)l * p.C + ch) * 2]; t1 += sm[((int64_t)l * p.C + ch) * 2 + 1]; }
        float* o = p.ws + ((int64_t)blockIdx.x * p.C + ch) * 2;
        o[0] = t0; o[1] = t1;
    }
}

// ---------------------------------------------------------------- backward: dx
template <typename T>
__global__ void gn_bwd_apply_kernel(const GnP p) {
    constexpr int N = Vec<T>::N;
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [C][2] per-sample channel sums, then [G][2]
    float* gs = sm + (int64_t)p.C * 2;
    const int b = blockIdx.x / p.NS, split = blockIdx.x - b * p.NS;
    for (int ch = threadIdx.x; ch < p.C; ch += blockDim.x) {
        float t0 = 0.f, t1 = 0.f;
        for (int s2 = 0; s2 < p.NS; ++s2) {
            const float* o = p.ws + ((int64_t)(b * p.NS + s2) * p.C + ch) * 2;
            t0 += o[0]; t1 += o[1];
        }
        sm[ch * 2] = t0 * p.gamma[ch]; sm[ch * 2 + 1] = t1 * p.gamma[ch];      // gamma-weighted: sum(dxhat*xhat), sum(dxhat)
        if (split == 0) {                                                      // per-sample sums for dgamma / dbeta
            float* o2 = p.ws + ((int64_t)p.B * p.NS * p.C + (int64_t)b * p.C + ch) * 2;
            o2[0] = t0; o2[1] = t1;
        }
    }
    __syncthreads();
    if (threadIdx.x < p.G) {
        const int g = threadIdx.x;
        float s2v = 0.f, s1v = 0.f;
        for (int ch = g * p.Cg; ch < (g + 1) * p.Cg; ++ch) { s2v += sm[ch * 2]; s1v += sm[ch * 2 + 1]; }
        const float inv_n = 1.0f / ((float)p.HW * (float)p.Cg);
        gs[g * 2] = s1v * inv_n; gs[g * 2 + 1] = s2v * inv_n;
    }
    __syncthreads();
    const int c = threadIdx.x % p.CC, pl = threadIdx.x / p.CC;
    float mu[N], rs[N], ga[N], be[N], s1[N], s2[N];
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const int ch = c * N + e, g = ch / p.Cg;
        mu[e] = p.mean[b * p.G + g]; rs[e] = p.rstd[b * p.G + g]; ga[e] = p.gamma[ch]; be[e] = p.beta[ch];
        s1[e] = gs[g * 2]; s2[e] = gs[g * 2 + 1];
    }
    const int p0 = split * p.pps, p1 = min(p.HW, p0 + p.pps);
    const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.HW * p.ldx + c * N;
    const T* gb = reinterpret_cast<const T*>(p.dy) + (int64_t)b * p.HW * p.lddy + c * N;
    T* ob = reinterpret_cast<T*>(p.dx) + (int64_t)b * p.HW * p.lddx + c * N;
    // optional gradient of the tensor's OTHER consumer (the residual / skip path that bypasses the norm): added here
    // instead of by a separate autograd accumulation pass
    const T* rb = p.dres ? reinterpret_cast<const T*>(p.dres) + (int64_t)b * p.HW * p.lddres + c * N : nullptr;
    for (int px = p0 + pl; px < p1; px += GN_UB * p.PP) {
        float v[GN_UB][N], d[GN_UB][N], r[GN_UB][N];
#pragma unroll
        for (int u = 0; u < GN_UB; ++u) {
            const int pu = px + u * p.PP, ps = pu < p1 ? pu : px;
            Vec<T>::ld(xb + (int64_t)ps * p.ldx, v[u]);
            Vec<T>::ld(gb + (int64_t)ps * p.lddy, d[u]);
            if (rb) Vec<T>::ld(rb + (int64_t)ps * p.lddres, r[u]);
        }
#pragma unroll
        for (int u = 0; u < GN_UB; ++u) {
            const int pu = px + u * p.PP;
#pragma unroll
            for (int e = 0; e < N; ++e) {
                const float h = (v[u][e] - mu[e]) * rs[e];
                float dz = d[u][e];
                if (p.silu) dz *= silu_grad(h * ga[e] + be[e]);
                v[u][e] = rs[e] * (dz * ga[e] - s1[e] - h * s2[e]);
                if (rb) v[u][e] += r[u][e];
            }
            if (pu < p1) Vec<T>::st(ob + (int64_t)pu * p.lddx, v[u]);
        }
    }
}

// dgamma/dbeta[c] (+)= sum over rows (b, split) of ws[row][c][{0,1}]: 64 columns x 4 row lanes, fixed order
__global__ __launch_bounds__(1024) void gn_param_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, int rows, int C, int accumulate) {
    __shared__ float red[2][16][32];               // 32 columns x 16 row lanes (x2 values): few rows per lane
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;   // thread -> (column pair lane, row lane of 32)
    const int c = blockIdx.x * 32 + cl;
    float ag = 0.f, ab = 0.f;
    if (c < C)
        for (int r = rl; r < rows; r += 32) { const float2 v = *reinterpret_cast<const float2*>(ws + ((int64_t)r * C + c) * 2); ag += v.x; ab += v.y; }
    // two-level fixed-order combine: 32 row lanes -> 16 -> 1
    if (rl >= 16) { red[0][rl - 16][cl] = ag; red[1][rl - 16][cl] = ab; }
    __syncthreads();
    if (rl < 16) { ag += red[0][rl][cl]; ab += red[1][rl][cl]; }
    __syncthreads();
    if (rl < 16) { red[0][rl][cl] = ag; red[1][rl][cl] = ab; }
    __syncthreads();
    if (rl == 0 && c < C) {
        float tg = 0.f, tb = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { tg += red[0][i][cl]; tb += red[1][i][cl]; }
        if (accumulate) { tg += dgamma[c]; tb += dbeta[c]; }
        dgamma[c] = tg;
        dbeta[c] = tb;
    }
}

static int gn_plan(GnP& p, int dtype) {
    const int N = dtype == PSG_BF16 ? 8 : 4;
    PSG_REQUIRE(p.B > 0 && p.HW > 0 && p.C > 0 && p.G > 0 && p.C % p.G == 0, PSG_ERR_SHAPE, "groupnorm: B=%d HW=%d C=%d G=%d", p.B, p.HW, p.C, p.G);
    PSG_REQUIRE(p.C % N == 0, PSG_ERR_SHAPE, "groupnorm: C=%d must be a multiple of %d", p.C, N);
    PSG_REQUIRE(p.G <= 64, PSG_ERR_SHAPE, "groupnorm: G=%d > 64", p.G);
    p.Cg = p.C / p.G;
    p.CC = p.C / N;
    PSG_REQUIRE(p.CC <= 1024, PSG_ERR_SHAPE, "groupnorm: C=%d too wide", p.C);
    p.PP = 256 / p.CC;
    if (p.PP < 1) p.PP = 1;
    if (p.PP * p.CC < 64) p.PP = (64 + p.CC - 1) / p.CC;
    int ns = (2048 + p.B - 1) / p.B;                       // ~8 workgroups per CU
    const int maxns = (p.HW + 4 * p.PP - 1) / (4 * p.PP);  // about 4 pixels per lane at least
    if (ns > maxns) ns = maxns;
    if (ns > GN_MAXSPLIT) ns = GN_MAXSPLIT;
    if (ns < 1) ns = 1;
    p.pps = (p.HW + ns - 1) / ns;
    p.NS = (p.HW + p.pps - 1) / p.pps;
    return PSG_OK;
}

}  // namespace psg
using namespace psg;

extern "C" {

int psg_gn_init_attrs(void) { return PSG_OK; }

int64_t psg_groupnorm_fwd_workspace_bytes(int B, int G) { return (int64_t)B * GN_MAXSPLIT * G * 2 * sizeof(float); }
int64_t psg_groupnorm_bwd_workspace_bytes(int B, int C) { return (int64_t)B * (GN_MAXSPLIT + 1) * C * 2 * sizeof(float); }

int psg_groupnorm_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma, const float* beta,
                      float* mean, float* rstd, int B, int HW, int C, int G, float eps, int silu, int dtype, void* ws,
                      psg_stream_t stream) {
    PSG_REQUIRE(x && y && gamma && beta && mean && rstd && ws, PSG_ERR_ARG, "groupnorm_fwd: null pointer");
    PSG_REQUIRE(dtype == PSG_F32 || dtype == PSG_BF16, PSG_ERR_DTYPE, "groupnorm_fwd: dtype %d", dtype);
    GnP p = {};
    p.x = x; p.y = y; p.gamma = gamma; p.beta = beta; p.mean = mean; p.rstd = rstd; p.ws = (float*)ws;
    p.ldx = ldx; p.ldy = ldy; p.B = B; p.HW = HW; p.C = C; p.G = G; p.eps = eps; p.silu = silu;
    int rc = gn_plan(p, dtype);
    if (rc) return rc;
    const int N = dtype == PSG_BF16 ? 8 : 4;
    PSG_REQUIRE(ldx >= C && ldy >= C && ldx % N == 0 && ldy % N == 0 && aligned16(x) && aligned16(y), PSG_ERR_ALIGN,
                "groupnorm_fwd: rows must be 16-byte aligned (ld multiple of %d)", N);
    hipStream_t s = (hipStream_t)stream;
    const int threads = p.CC * p.PP, grid = B * p.NS;
    const size_t lds1 = (size_t)p.PP * C * 2 * sizeof(float), lds2 = (size_t)G * 2 * sizeof(float);
    ProfScope prof(PROF_GN, 2.0 * (double)B * HW * C * (double)(dtype == PSG_BF16 ? 2 : 4), s);
    if (dtype == PSG_F32) {
        hipLaunchKernelGGL(gn_stats_kernel<float>, dim3(grid), dim3(threads), lds1, s, p);
        hipLaunchKernelGGL(gn_apply_kernel<float>, dim3(grid), dim3(threads), lds2, s, p);
    } else {
        hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, dim3(grid), dim3(threads), lds1, s, p);
        hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, dim3(grid), dim3(threads), lds2, s, p);
    }
    PSG_LAUNCH_CHECK("groupnorm_fwd");
    return PSG_OK;
}

int psg_groupnorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma, const float* beta,
                      const float* mean, const float* rstd, void* dx, int64_t lddx, float* dgamma, float* dbeta, int B,
                      int HW, int C, int G, int silu, int accumulate, int dtype, void* ws, psg_stream_t stream) {
    return psg_groupnorm_bwd_res(dy, lddy, x, ldx, gamma, beta, mean, rstd, nullptr, 0, dx, lddx, dgamma, dbeta, B, HW, C, G, silu,
                                 accumulate, dtype, ws, stream);
}

int psg_groupnorm_bwd_res(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma, const float* beta,
                          const float* mean, const float* rstd, const void* dres, int64_t lddres, void* dx, int64_t lddx,
                          float* dgamma, float* dbeta, int B, int HW, int C, int G, int silu, int accumulate, int dtype, void* ws,
                          psg_stream_t stream) {
    PSG_REQUIRE(dy && x && gamma && beta && mean && rstd && dx && dgamma && dbeta && ws, PSG_ERR_ARG, "groupnorm_bwd: null pointer");
    PSG_REQUIRE(dtype == PSG_F32 || dtype == PSG_BF16, PSG_ERR_DTYPE, "groupnorm_bwd: dtype %d", dtype);
    GnP p = {};
    p.x = x; p.dy = dy; p.dx = dx; p.gamma = gamma; p.beta = beta; p.mean = const_cast<float*>(mean); p.rstd = const_cast<float*>(rstd);
    p.ws = (float*)ws; p.ldx = ldx; p.lddy = lddy; p.lddx = lddx; p.B = B; p.HW = HW; p.C = C; p.G = G; p.silu = silu;
    p.dres = dres; p.lddres = lddres;
    int rc = gn_plan(p, dtype);
    if (rc) return rc;
    const int N = dtype == PSG_BF16 ? 8 : 4;
    PSG_REQUIRE(ldx >= C && lddy >= C && lddx >= C && ldx % N == 0 && lddy % N == 0 && lddx % N == 0 && aligned16(x) && aligned16(dy) &&
                aligned16(dx), PSG_ERR_ALIGN, "groupnorm_bwd: rows must be 16-byte aligned (ld multiple of %d)", N);
    PSG_REQUIRE(!dres || (lddres >= C && lddres % N == 0 && aligned16(dres)), PSG_ERR_ALIGN, "groupnorm_bwd: dres rows must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int threads = p.CC * p.PP, grid = B * p.NS;
    const size_t lds1 = (size_t)p.PP * C * 2 * sizeof(float), lds2 = ((size_t)C + G) * 2 * sizeof(float);
    ProfScope prof(PROF_GN, 3.0 * (double)B * HW * C * (double)(dtype == PSG_BF16 ? 2 : 4), s);
    if (dtype == PSG_F32) {
        hipLaunchKernelGGL(gn_bwd_reduce_kernel<float>, dim3(grid), dim3(threads), lds1, s, p);
        hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, dim3(grid), dim3(threads), lds2, s, p);
    } else {
        hipLaunchKernelGGL(gn_bwd_reduce_kernel<bf16_t>, dim3(grid), dim3(threads), lds1, s, p);
        hipLaunchKernelGGL(gn_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(threads), lds2, s, p);
    }
    PSG_LAUNCH_CHECK("groupnorm_bwd");
    hipLaunchKernelGGL(gn_param_reduce_kernel, dim3((C + 31) / 32), dim3(1024), 0, s, (const float*)ws + (int64_t)B * p.NS * C * 2, dgamma, dbeta,
                       B, C, accumulate);
    PSG_LAUNCH_CHECK("groupnorm_param_reduce");
    return PSG_OK;
}

}  // extern "C"
