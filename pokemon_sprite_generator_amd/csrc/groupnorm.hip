// GroupNorm (+ fused SiLU) forward / backward, channels-last [B, HW, C].  Roofline: HBM.
//
// FUSED (register-resident) path - every shape of the U-Net: one workgroup owns one (sample, channel slab) - a run of
// whole groups whose width is a multiple of the 16-byte chunk - and keeps the slab's HW x slabC values IN REGISTERS
// (<= 16 packed 16-byte chunks per lane), so the tensor is read from memory exactly once:
//   forward : load -> channel sums -> group mean -> centred second pass over the registers (exact two-pass variance,
//             no E[x^2]-E[x]^2 cancellation) -> normalise (+SiLU) -> store              = 1 read + 1 write
//   backward: load x, dy -> per-channel (sum dz*xhat, sum dz) -> group sums -> dx (+ the bypass gradient) -> store;
//             per-sample channel sums go to the workspace for dgamma / dbeta            = 2(+1) reads + 1 write
// against 2 + 1 and 4 + 1 passes of the split kernels below (kept as the general fallback: shapes whose slab does not
// fit the register budget).  Reductions are fixed-order LDS trees, as below: results are run-to-run identical.
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


// ---------------------------------------------------------------- fused, register-resident kernels
struct GnF {
    const void *x, *dy, *dres; void *y, *dx;
    const float *gamma, *beta; float *mean, *rstd, *chan_ws;
    int64_t ldx, ldy, lddy, lddres, lddx;
    int B, HW, C, G, Cg, slabC, CCs, PP, nslab, SG;   // slabC channels = CCs chunks = SG groups per slab; PP pixel lanes
    float eps; int silu;
};

// A chunk of N consecutive channels as it lies in memory (bf16 stays packed in registers).  The forward kernel moves
// 16-byte chunks; the backward kernel, which holds TWO tensors plus twice the per-channel constants, moves 8-byte
// chunks: the per-element state (constants, temporaries) halves while the slab bytes per lane stay the same.
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <typename T, int N> struct Chk;
template <> struct Chk<float, 4> { typedef f32x4 type; };
template <> struct Chk<float, 2> { typedef f32x2 type; };
template <> struct Chk<bf16_t, 8> { typedef bf16x8 type; };
template <> struct Chk<bf16_t, 4> { typedef bf16x4 type; };
template <typename T, int N> __device__ __forceinline__ float chk_get(const typename Chk<T, N>::type& r, int e) { return (float)r[e]; }
template <typename T, int N> __device__ __forceinline__ void chk_st(void* p, const float* v) {
    typename Chk<T, N>::type t;
#pragma unroll
    for (int e = 0; e < N; ++e) t[e] = (T)v[e];
    *reinterpret_cast<typename Chk<T, N>::type*>(p) = t;
}

// sums over the PP pixel lanes of part[pl][ch][stride] (component comp), then over each group's channels: grp[g], g < SG.
// Two levels so that the whole workgroup works (nseg = blockDim / slabC segments of the pixel lanes in parallel) instead
// of slabC lanes walking PP dependent LDS reads each: the reduction phases are dead time for the memory pipe.
// `seg` (nseg * slabC floats) and `chan` (slabC floats) are scratch.  Fixed order.  Ends with a barrier.
__device__ __forceinline__ void gn_tree(const GnF& p, const float* part, float* seg, float* chan, float* grp, int stride, int comp) {
    const int nseg = max(1, min((int)blockDim.x / p.slabC, p.PP));
    __syncthreads();
    for (int idx = threadIdx.x; idx < nseg * p.slabC; idx += blockDim.x) {      // (one trip unless slabC > blockDim)
        const int ch = idx % p.slabC, sg = idx / p.slabC;
        float t0 = 0.f, t1 = 0.f;
        int l = sg;
        for (; l + nseg < p.PP; l += 2 * nseg) {
            t0 += part[((int64_t)l * p.slabC + ch) * stride + comp];
            t1 += part[((int64_t)(l + nseg) * p.slabC + ch) * stride + comp];
        }
        if (l < p.PP) t0 += part[((int64_t)l * p.slabC + ch) * stride + comp];
        seg[sg * p.slabC + ch] = t0 + t1;
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < p.slabC; ch += blockDim.x) {
        float t = 0.f;
        for (int sg = 0; sg < nseg; ++sg) t += seg[sg * p.slabC + ch];
        chan[ch] = t;
    }
    __syncthreads();
    if (grp && (int)threadIdx.x < p.SG) {
        float t = 0.f;
        for (int ch = threadIdx.x * p.Cg; ch < ((int)threadIdx.x + 1) * p.Cg; ++ch) t += chan[ch];
        grp[threadIdx.x] = t;
    }
    if (grp) __syncthreads();
}

// Both components of part[pl][ch][2] in one walk (the backward kernel's two sums: half the barriers, 8-byte LDS reads).
// `seg` holds 2 * nseg * slabC floats.  Same summation order per component as gn_tree.
__device__ __forceinline__ void gn_tree2(const GnF& p, const float* part, float* seg, float* c0, float* c1) {
    const int nseg = max(1, min((int)blockDim.x / p.slabC, p.PP));
    const f32x2* part2 = reinterpret_cast<const f32x2*>(part);
    f32x2* seg2 = reinterpret_cast<f32x2*>(seg);
    __syncthreads();
    for (int idx = threadIdx.x; idx < nseg * p.slabC; idx += blockDim.x) {
        const int ch = idx % p.slabC, sg = idx / p.slabC;
        f32x2 t0 = {0.f, 0.f}, t1 = {0.f, 0.f};
        int l = sg;
        for (; l + nseg < p.PP; l += 2 * nseg) {
            t0 += part2[l * p.slabC + ch];
            t1 += part2[(l + nseg) * p.slabC + ch];
        }
        if (l < p.PP) t0 += part2[l * p.slabC + ch];
        seg2[sg * p.slabC + ch] = t0 + t1;
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < p.slabC; ch += blockDim.x) {
        f32x2 t = {0.f, 0.f};
        for (int sg = 0; sg < nseg; ++sg) t += seg2[sg * p.slabC + ch];
        c0[ch] = t[0]; c1[ch] = t[1];
    }
    __syncthreads();
}

// Workgroup ids go to the 8 XCDs round-robin; the slabs of ONE sample (consecutive logical ids) share 64-byte sectors
// when a slab row is narrower than a sector pair, so each XCD gets a contiguous range of logical ids: the partner
// slab's half of a sector is then an L2 hit instead of a second fetch through the fabric.
__device__ __forceinline__ int gn_xcd_lid() {
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// The packed chunks are converted to fp32 again in every pass: without this fence hipcc keeps one fp32 copy of the whole
// slab live across the passes (2x the registers of the packed form, which is what bounds the slab a workgroup can hold).
template <typename Raw, int R>
__device__ __forceinline__ void gn_fence(Raw (&raw)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) asm volatile("" : "+v"(raw[r]));
}

template <typename T, int N, int R, bool SILU>
__global__ __launch_bounds__(512, 4) void gn_fwd_fused_kernel(const GnF p) {
    typedef typename Chk<T, N>::type Raw;
    constexpr bool FAST = sizeof(T) == 2;                            // bf16 compute: v_rcp / fma forms (psg_common.h)
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [PP][slabC] partials | [nseg][slabC] | [slabC] | [SG] | [SG]
    float* segb = sm + (int64_t)p.PP * p.slabC;
    float* chan = segb + max((int)blockDim.x, p.slabC);
    float* gmean = chan + p.slabC;
    float* gvar = gmean + p.SG;
    const int lid = gn_xcd_lid();
    const int b = lid / p.nslab, slab = lid - b * p.nslab;
    const int c = threadIdx.x % p.CCs, pl = threadIdx.x / p.CCs;
    const int ch0 = slab * p.slabC + c * N;
    // Addresses = wave-uniform 64-bit base (sample, slab, chunk row r: scalar registers) + ONE 32-bit per-lane offset per
    // tensor.  (Per-chunk 64-bit vector addresses - base + px*ld, or a bumped pointer, which hipcc unrolls back into
    // R independent addresses - cost 2 x R registers per tensor: more than the packed slab itself.)
    constexpr int ESZ = (int)sizeof(T);
    const char* xs = reinterpret_cast<const char*>(p.x) + ((int64_t)b * p.HW * p.ldx + slab * p.slabC) * ESZ;
    const uint32_t xo = (uint32_t)((pl * (int)p.ldx + c * N) * ESZ);
    const uint32_t xstep = (uint32_t)(p.PP * (int)p.ldx * ESZ);      // (32-bit: a sample's slab spans far less than 4 GiB)
    Raw raw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (pl + r * p.PP < p.HW) raw[r] = *reinterpret_cast<const Raw*>(xs + (uint32_t)(xo + r * xstep));
        else { Raw z = {}; raw[r] = z; }
    }
    // A chunk lies in at most two groups (Cg >= N, checked by the plan): group glo for elements e < eb, glo + 1 after.
    // (Per-element group indices would cost a division and an LDS address register per element and per table.)
    const int glo = (c * N) / p.Cg, eb = (glo + 1) * p.Cg - c * N;
    const int ghi = min(glo + 1, p.SG - 1);
    // ---- pass A: means
    float s[N];
#pragma unroll
    for (int e = 0; e < N; ++e) s[e] = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < N; ++e) s[e] += chk_get<T, N>(raw[r], e);   // (absent pixels hold zeros)
    float* row = sm + (int64_t)pl * p.slabC + c * N;
#pragma unroll
    for (int e = 0; e < N; ++e) row[e] = s[e];
    gn_tree(p, sm, segb, chan, gmean, 1, 0);
    gn_fence(raw);
    const float inv_n = 1.0f / ((float)p.HW * (float)p.Cg);
    float mu[N];
    {
        const float m_lo = gmean[glo] * inv_n, m_hi = gmean[ghi] * inv_n;
#pragma unroll
        for (int e = 0; e < N; ++e) mu[e] = e < eb ? m_lo : m_hi;
    }
    // ---- pass B: centred second moment, from the registers
#pragma unroll
    for (int e = 0; e < N; ++e) s[e] = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float w = (pl + r * p.PP) < p.HW ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < N; ++e) { const float d = chk_get<T, N>(raw[r], e) - mu[e]; s[e] = FAST ? __builtin_fmaf(w * d, d, s[e]) : s[e] + w * d * d; }
    }
#pragma unroll
    for (int e = 0; e < N; ++e) row[e] = s[e];
    gn_tree(p, sm, segb, chan, gvar, 1, 0);
    gn_fence(raw);
    if ((int)threadIdx.x < p.SG) {
        const int g = slab * p.SG + threadIdx.x;
        p.mean[b * p.G + g] = gmean[threadIdx.x] * inv_n;
        p.rstd[b * p.G + g] = 1.0f / sqrtf(gvar[threadIdx.x] * inv_n + p.eps);
    }
    // ---- normalise (+SiLU) and store
    float sc[N], sh[N];
    {
        const float r_lo = 1.0f / sqrtf(gvar[glo] * inv_n + p.eps), r_hi = 1.0f / sqrtf(gvar[ghi] * inv_n + p.eps);
#pragma unroll
        for (int e = 0; e < N; ++e) {
            sc[e] = (e < eb ? r_lo : r_hi) * p.gamma[ch0 + e];
            sh[e] = p.beta[ch0 + e] - mu[e] * sc[e];
        }
    }
    char* ys = reinterpret_cast<char*>(p.y) + ((int64_t)b * p.HW * p.ldy + slab * p.slabC) * ESZ;
    const uint32_t yo = (uint32_t)((pl * (int)p.ldy + c * N) * ESZ);
    const uint32_t ystep = (uint32_t)(p.PP * (int)p.ldy * ESZ);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (pl + r * p.PP < p.HW) {
            float v[N];
#pragma unroll
            for (int e = 0; e < N; ++e) { v[e] = mad_t<FAST>(chk_get<T, N>(raw[r], e), sc[e], sh[e]); if (SILU) v[e] = silu_t<FAST>(v[e]); }
            chk_st<T, N>(ys + (uint32_t)(yo + r * ystep), v);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <typename T, int N, int R, bool SILU, bool RES>
__global__ __launch_bounds__(512, 4) void gn_bwd_fused_kernel(const GnF p) {
    typedef typename Chk<T, N>::type Raw;
    constexpr bool FAST = sizeof(T) == 2;
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [PP][slabC][2] partials | [nseg][slabC][2] | [4][slabC] | 4 x [SG]
    float* segb = sm + (int64_t)p.PP * p.slabC * 2;
    float* chan = segb + 2 * max((int)blockDim.x, p.slabC);          // [0],[1]: gamma-weighted; [2],[3]: raw channel totals
    float* g1 = chan + 4 * p.slabC;
    float* g2 = g1 + p.SG;
    float* gmu = g2 + p.SG;          // the slab's group statistics, staged once (per-element global addresses of mean /
    float* grs = gmu + p.SG;         // rstd kept live across the passes cost 32 registers)
    const int lid = gn_xcd_lid();
    const int b = lid / p.nslab, slab = lid - b * p.nslab;
    const int c = threadIdx.x % p.CCs, pl = threadIdx.x / p.CCs;
    const int ch0 = slab * p.slabC + c * N;
    // (uniform 64-bit bases + one 32-bit lane offset per tensor: see the forward kernel)
    constexpr int ESZ = (int)sizeof(T);
    const char* xs = reinterpret_cast<const char*>(p.x) + ((int64_t)b * p.HW * p.ldx + slab * p.slabC) * ESZ;
    const char* gs = reinterpret_cast<const char*>(p.dy) + ((int64_t)b * p.HW * p.lddy + slab * p.slabC) * ESZ;
    const uint32_t xo = (uint32_t)((pl * (int)p.ldx + c * N) * ESZ), go = (uint32_t)((pl * (int)p.lddy + c * N) * ESZ);
    const uint32_t xstep = (uint32_t)(p.PP * (int)p.ldx * ESZ), gstep = (uint32_t)(p.PP * (int)p.lddy * ESZ);
    Raw rx[R], rd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (pl + r * p.PP < p.HW) {
            rx[r] = *reinterpret_cast<const Raw*>(xs + (uint32_t)(xo + r * xstep));
            rd[r] = *reinterpret_cast<const Raw*>(gs + (uint32_t)(go + r * gstep));
        } else { Raw z = {}; rx[r] = z; rd[r] = z; }
    }
    if ((int)threadIdx.x < p.SG) {
        gmu[threadIdx.x] = p.mean[b * p.G + slab * p.SG + threadIdx.x];
        grs[threadIdx.x] = p.rstd[b * p.G + slab * p.SG + threadIdx.x];
    }
    __syncthreads();
    const int glo = (c * N) / p.Cg, eb = (glo + 1) * p.Cg - c * N;      // (two groups per chunk at most: see the forward kernel)
    const int ghi = min(glo + 1, p.SG - 1);
    const float m_lo = gmu[glo], m_hi = gmu[ghi], r_lo = grs[glo], r_hi = grs[ghi];
    float A[N], Bc[N];
#pragma unroll
    for (int e = 0; e < N; ++e) {
        A[e] = (e < eb ? r_lo : r_hi) * p.gamma[ch0 + e];
        Bc[e] = p.beta[ch0 + e] - (e < eb ? m_lo : m_hi) * A[e];
    }
    float a0[N], a1[N];
#pragma unroll
    for (int e = 0; e < N; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const float xv = chk_get<T, N>(rx[r], e);
            float dz = chk_get<T, N>(rd[r], e);                  // (absent pixels hold dy = 0)
            if (SILU) dz *= silu_grad_t<FAST>(mad_t<FAST>(xv, A[e], Bc[e]));
            a0[e] = mad_t<FAST>(dz, xv, a0[e]); a1[e] += dz;
        }
        __builtin_amdgcn_sched_barrier(0);      // one chunk's temporaries at a time (else hipcc interleaves all R: spills)
    }
    float* row = sm + ((int64_t)pl * p.slabC + c * N) * 2;
#pragma unroll
    for (int e = 0; e < N; ++e) { row[2 * e] = a0[e]; row[2 * e + 1] = a1[e]; }
    // channel totals, per-sample (sum dz*xhat, sum dz) out for dgamma / dbeta, gamma-weighted group sums
    gn_tree2(p, sm, segb, chan + 2 * p.slabC, chan + 3 * p.slabC);
    for (int ch = threadIdx.x; ch < p.slabC; ch += blockDim.x) {
        const float t0 = chan[2 * p.slabC + ch], t1 = chan[3 * p.slabC + ch];
        const float m_ = gmu[ch / p.Cg], r_ = grs[ch / p.Cg];
        const float dzh = r_ * t0 - m_ * r_ * t1;          // sum dz * xhat
        float* o = p.chan_ws + ((int64_t)b * p.C + slab * p.slabC + ch) * 2;
        o[0] = dzh; o[1] = t1;
        const float gm = p.gamma[slab * p.slabC + ch];
        chan[ch] = dzh * gm;                // sum(dxhat * xhat) contribution
        chan[p.slabC + ch] = t1 * gm;       // sum(dxhat)
    }
    __syncthreads();
    if ((int)threadIdx.x < p.SG) {
        float t0 = 0.f, t1 = 0.f;
        for (int ch = threadIdx.x * p.Cg; ch < ((int)threadIdx.x + 1) * p.Cg; ++ch) { t0 += chan[ch]; t1 += chan[p.slabC + ch]; }
        const float inv_n = 1.0f / ((float)p.HW * (float)p.Cg);
        g2[threadIdx.x] = t0 * inv_n; g1[threadIdx.x] = t1 * inv_n;
    }
    __syncthreads();
    gn_fence(rx);
    gn_fence(rd);
    float P[N], Q[N];
    {
        const float P_lo = r_lo * r_lo * g2[glo], P_hi = r_hi * r_hi * g2[ghi];
        const float Q_lo = m_lo * P_lo - r_lo * g1[glo], Q_hi = m_hi * P_hi - r_hi * g1[ghi];
#pragma unroll
        for (int e = 0; e < N; ++e) { P[e] = e < eb ? P_lo : P_hi; Q[e] = e < eb ? Q_lo : Q_hi; }
    }
    char* os = reinterpret_cast<char*>(p.dx) + ((int64_t)b * p.HW * p.lddx + slab * p.slabC) * ESZ;
    const char* rb = RES ? reinterpret_cast<const char*>(p.dres) + ((int64_t)b * p.HW * p.lddres + slab * p.slabC) * ESZ : nullptr;
    const uint32_t oo = (uint32_t)((pl * (int)p.lddx + c * N) * ESZ), ro = (uint32_t)((pl * (int)p.lddres + c * N) * ESZ);
    const uint32_t ostep = (uint32_t)(p.PP * (int)p.lddx * ESZ), rstep = (uint32_t)(p.PP * (int)p.lddres * ESZ);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (pl + r * p.PP < p.HW) {
            float v[N];
            Raw res = {};
            if (RES) res = *reinterpret_cast<const Raw*>(rb + (uint32_t)(ro + r * rstep));
#pragma unroll
            for (int e = 0; e < N; ++e) {
                const float xv = chk_get<T, N>(rx[r], e);
                float dz = chk_get<T, N>(rd[r], e);
                if (SILU) dz *= silu_grad_t<FAST>(mad_t<FAST>(xv, A[e], Bc[e]));
                v[e] = FAST ? __builtin_fmaf(A[e], dz, __builtin_fmaf(-xv, P[e], Q[e])) : A[e] * dz - xv * P[e] + Q[e];
                if (RES) v[e] += chk_get<T, N>(res, e);
            }
            chk_st<T, N>(os + (uint32_t)(oo + r * ostep), v);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

static int gcd_i(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

// Slab plan of the fused kernels: the widest run of whole groups (a multiple of the 16-byte chunk that divides C) whose
// HW x slabC values fit rmax chunks per lane within max_threads lanes.  Returns R (4, 8 or 16), or 0 = use the split kernels.
static int gn_fused_plan(GnF& p, int N, int max_threads, int rmax) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("PSG_GN_FUSED"); off = (e && atoi(e) == 0) ? 1 : 0; }
    if (off) return 0;
    { static int mt = -1; if (mt < 0) { const char* e = getenv("PSG_GN_MAXT"); mt = e ? atoi(e) : 0; } if (mt >= 64 && mt < max_threads) max_threads = mt; }
    if (p.C % p.G || p.C % N) return 0;
    p.Cg = p.C / p.G;
    if (p.Cg < N) return 0;                                // (a chunk must not span more than two groups)
    const int U = p.Cg / gcd_i(p.Cg, N) * N;             // lcm(Cg, N): smallest slab of whole groups and whole chunks
    if (p.C % U) return 0;
    const int units = p.C / U;
    int best_k = 0;
    for (int k = 1; k <= units; ++k) {
        if (units % k) continue;
        const int ccs = k * U / N;
        const int pp_need = (p.HW + rmax - 1) / rmax;
        if (ccs * pp_need > max_threads || k * U > 640) break;
        best_k = k;
    }
    if (!best_k) return 0;
    p.slabC = best_k * U; p.CCs = p.slabC / N; p.SG = p.slabC / p.Cg; p.nslab = p.C / p.slabC;
    int pp = (p.HW + rmax - 1) / rmax;
    const int pp_min = (256 + p.CCs - 1) / p.CCs;        // at least ~256 lanes per workgroup
    if (pp < pp_min) pp = pp_min;
    if (pp > p.HW) pp = p.HW;
    if (p.CCs * pp > max_threads) pp = max_threads / p.CCs;
    p.PP = pp;
    const int r = (p.HW + pp - 1) / pp;
    if (r > rmax) return 0;
    return r <= 4 ? 4 : (r <= 8 ? 8 : 16);
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
    ProfScope prof(PROF_GN, 2.0 * (double)B * HW * C * (double)(dtype == PSG_BF16 ? 2 : 4), s, 2.0 * (double)B * HW * C * (double)(dtype == PSG_BF16 ? 2 : 4));
    {
        GnF f = {};
        f.x = x; f.y = y; f.gamma = gamma; f.beta = beta; f.mean = mean; f.rstd = rstd; f.ldx = ldx; f.ldy = ldy;
        f.B = B; f.HW = HW; f.C = C; f.G = G; f.eps = eps; f.silu = silu;
        // 256-lane workgroups first: four or more of them share a CU and their load / reduce / store phases interleave
        // (measured: -9 % GroupNorm time against 512 lanes); 512 lanes where a 256-lane slab cannot hold whole groups
        int R = gn_fused_plan(f, dtype == PSG_BF16 ? 8 : 4, 256, 16);
        if (!R) R = gn_fused_plan(f, dtype == PSG_BF16 ? 8 : 4, 512, 16);
        const size_t lds = ((size_t)f.PP * f.slabC + (f.CCs * f.PP > f.slabC ? f.CCs * f.PP : f.slabC) + f.slabC + 2 * f.SG) * sizeof(float);
        if (R && lds <= 64 * 1024) {
            const dim3 g(B * f.nslab), t(f.CCs * f.PP);
#define PSG_GN_FWD(TT, NN, RR)                                                                     \
    do { if (silu) hipLaunchKernelGGL((gn_fwd_fused_kernel<TT, NN, RR, true>), g, t, lds, s, f);  \
         else hipLaunchKernelGGL((gn_fwd_fused_kernel<TT, NN, RR, false>), g, t, lds, s, f); } while (0)
            if (dtype == PSG_F32) { if (R == 4) PSG_GN_FWD(float, 4, 4); else if (R == 8) PSG_GN_FWD(float, 4, 8); else PSG_GN_FWD(float, 4, 16); }
            else { if (R == 4) PSG_GN_FWD(bf16_t, 8, 4); else if (R == 8) PSG_GN_FWD(bf16_t, 8, 8); else PSG_GN_FWD(bf16_t, 8, 16); }
#undef PSG_GN_FWD
            PSG_LAUNCH_CHECK("groupnorm_fwd_fused");
            return PSG_OK;
        }
    }
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
    ProfScope prof(PROF_GN, (dres ? 4.0 : 3.0) * (double)B * HW * C * (double)(dtype == PSG_BF16 ? 2 : 4), s, (dres ? 4.0 : 3.0) * (double)B * HW * C * (double)(dtype == PSG_BF16 ? 2 : 4));
    {
        GnF f = {};
        f.x = x; f.dy = dy; f.dres = dres; f.dx = dx; f.gamma = gamma; f.beta = beta; f.mean = const_cast<float*>(mean);
        f.rstd = const_cast<float*>(rstd); f.chan_ws = (float*)ws; f.ldx = ldx; f.lddy = lddy; f.lddres = lddres; f.lddx = lddx;
        f.B = B; f.HW = HW; f.C = C; f.G = G; f.silu = silu;
        // 8-byte chunks, <= 8 per lane and tensor (16 fit the register file since the v_rcp forms - 120 VGPRs - and let the
        // 27x27 slabs run in 256 lanes, but measured no faster: 640 vs 648 us over the U-Net's eight shapes)
        int R = gn_fused_plan(f, dtype == PSG_BF16 ? 4 : 2, 256, 8);
        if (!R) R = gn_fused_plan(f, dtype == PSG_BF16 ? 4 : 2, 512, 8);
        // These kernels wait on memory (round 4, rocprofv3: 50 % of the wave cycles; the LDS pipe is active 1 % of them, so its
        // bank conflicts do not matter) and their slab rows are narrower than a 128-byte line (27x27x320: 40-byte rows, 3.1 TB/s;
        // 14x14x640: 80 bytes, 4.5).  16-byte chunks double the row a (lanes x chunks) budget covers: with the bypass gradient
        // as a third input stream -8...-20 % per launch (27x27x640 303 -> 267 us, 14x14x1280 143 -> 122), without it +-0
        // (14x14x640 +10 %): taken for the bypass form only.
        int NN = dtype == PSG_BF16 ? 4 : 2;
        {
            static int wide = -1;                          // PSG_GN_BWD_WIDE=0: 8-byte chunks only (A/B)
            if (wide < 0) { const char* e = getenv("PSG_GN_BWD_WIDE"); wide = (e && atoi(e) == 0) ? 0 : 1; }
            if (wide && dres && dtype == PSG_BF16 && R && f.slabC * 2 < 128) {
                GnF f2 = f;
                int R2 = gn_fused_plan(f2, 8, 256, 8);
                if (!R2 || f2.slabC <= f.slabC) { f2 = f; R2 = gn_fused_plan(f2, 8, 512, 8); }
                if (R2 && f2.slabC > f.slabC) { f = f2; R = R2; NN = 8; }
            }
        }
        const size_t lds = ((size_t)f.PP * f.slabC * 2 + 2 * (f.CCs * f.PP > f.slabC ? f.CCs * f.PP : f.slabC) + 4 * f.slabC + 4 * f.SG) * sizeof(float);
        if (R && lds <= 64 * 1024) {
            const dim3 g(B * f.nslab), t(f.CCs * f.PP);
#define PSG_GN_BWD(TT, NN, RR)                                                                                  \
    do { if (silu && dres) hipLaunchKernelGGL((gn_bwd_fused_kernel<TT, NN, RR, true, true>), g, t, lds, s, f);  \
         else if (silu) hipLaunchKernelGGL((gn_bwd_fused_kernel<TT, NN, RR, true, false>), g, t, lds, s, f);    \
         else if (dres) hipLaunchKernelGGL((gn_bwd_fused_kernel<TT, NN, RR, false, true>), g, t, lds, s, f);    \
         else hipLaunchKernelGGL((gn_bwd_fused_kernel<TT, NN, RR, false, false>), g, t, lds, s, f); } while (0)
            if (dtype == PSG_F32) { if (R == 4) PSG_GN_BWD(float, 2, 4); else PSG_GN_BWD(float, 2, 8); }
            else if (NN == 8) { if (R == 4) PSG_GN_BWD(bf16_t, 8, 4); else PSG_GN_BWD(bf16_t, 8, 8); }
            else { if (R == 4) PSG_GN_BWD(bf16_t, 4, 4); else PSG_GN_BWD(bf16_t, 4, 8); }
#undef PSG_GN_BWD
            PSG_LAUNCH_CHECK("groupnorm_bwd_fused");
            hipLaunchKernelGGL(gn_param_reduce_kernel, dim3((C + 31) / 32), dim3(1024), 0, s, (const float*)ws, dgamma, dbeta, B, C, accumulate);
            PSG_LAUNCH_CHECK("groupnorm_param_reduce");
            return PSG_OK;
        }
    }
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
