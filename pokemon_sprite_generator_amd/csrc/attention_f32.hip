// Exact-fp32 multi-head attention core on the matrix cores (v_mfma_f32_32x32x2_f32): the strict-parity path's
// counterpart of attention_mfma.hip - same one-workgroup-per-(sample, head) structure, same softmax / dropout / lse /
// delta conventions, fp32 operands end to end (every product and sum is an fp32 FMA chain, as on the CPU).
//
// The fp32 MFMA takes ONE value per lane and operand: A[row = lane%32][k = lane/32], B[k = lane/32][col = lane%32], so a
// 32x32 tile advances the reduction by 2 per instruction.  The reduction order is free as long as both operands agree:
//   scores  S^T[key][q] = sum_d K[key][d] Q[q][d]: step s pairs d = s (lanes 0-31) with d = s + d/2 (lanes 32-63), so a
//           lane reads its half of a K row contiguously from LDS (ds_read_b128 = 4 steps) and holds its half of the Q row
//           in registers, straight from memory;
//   O^T[dcol][q] += V^T[dcol][key] P^T[key][q]: step r pairs the two keys that accumulator register r of the score tile
//           holds in the two lane halves (key = acc_row(r, half)): the probability is already in place as the B operand,
//           and the A operand is one coalesced ds_read_b32 of a V row.
// The backward kernels use the same two patterns (dQ: query on the lane; dK / dV: key on the lane, Q and dO in LDS).
// K / V (or Q / dO) live in LDS as fp32 rows of d*4 + 16 bytes: (d + 4) mod 64 dwords in {4, 20, 36} keeps the 16 rows of
// a ds_read_b128 group on distinct banks for every head_dim that is a multiple of 16.
#include "psg_common.h"

namespace psg {

struct AttnFP {
    const float *q, *k, *v, *o, *dout;
    float *out, *dq, *dk, *dv;
    float* lse; float* delta;
    int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    int B, H, L, S, d;
    float scale;
    uint32_t drop_thresh; float drop_scale; uint64_t seed;
    const uint64_t* seed_dev;
};

__device__ __forceinline__ int acc_row32(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// rows [row0, row0 + nrows) of a [*, d] fp32 matrix -> LDS rows of `stride` bytes; rows >= nvalid_end are zero
__device__ __forceinline__ void stage_f32(char* dst, const float* src, int64_t ld, int row0, int nrows, int nvalid_end, int d, int stride,
                                          int tid, int nthreads) {
    const int cpr = d >> 2;                        // 16-byte chunks per row
    const int n = nrows * cpr;
    constexpr int U = 8;
    for (int e0 = tid; e0 < n; e0 += nthreads * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * nthreads;
            const int r = e / cpr, c = e - r * cpr;
            const int gr = row0 + r;
            const bool ok = e < n && gr < nvalid_end;
            const f32x4 t = *reinterpret_cast<const f32x4*>(src + (ok ? (int64_t)gr * ld + c * 4 : 0));
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            v[u] = ok ? t : z;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * nthreads;
            const int r = e / cpr, c = e - r * cpr;
            if (e < n) *reinterpret_cast<f32x4*>(dst + r * stride + c * 16) = v[u];
        }
    }
}

// acc[key][col] += sum over the lane's half row: A from an LDS row image (row = tile row0 + lane%32), B from registers
template <int HD>
__device__ __forceinline__ f32x16 dot_rows(const char* img, int stride, int row, int half, const float (&breg)[HD], f32x16 acc) {
    const char* rp = img + row * stride + half * HD * 4;
#pragma unroll
    for (int s4 = 0; s4 < HD / 4; ++s4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(rp + s4 * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], breg[4 * s4 + e], acc, 0, 0, 0);
    }
    return acc;
}

// out[t][dcol][col] += sum_r img[row0 + acc_row(r, half)][t*32 + lane%32] * b[r]   (the "accumulator as B operand" step)
template <int NDT>
__device__ __forceinline__ void acc_cols(const char* img, int stride, int row0, int fr, int fh, const f32x16& b, f32x16 (&out)[NDT]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const char* rp = img + (row0 + acc_row32(r, fh)) * stride + fr * 4;
#pragma unroll
        for (int t = 0; t < NDT; ++t) {
            const float a = *reinterpret_cast<const float*>(rp + t * 128);
            out[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[r], out[t], 0, 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward
template <int ND>
__global__ __launch_bounds__(256, 1) void attn_fwd_f32(const AttnFP p) {
    constexpr int D = ND * 16, HD = D / 2, NDT = (D + 31) / 32, STR = D * 4 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Sp = (p.S + 31) & ~31;
    char* Ks = smem;
    char* Vs = smem + Sp * STR;
    const int bh = blockIdx.x, b = bh / p.H, hd = bh - b * p.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* qg = p.q + (int64_t)b * p.L * p.ldq + hd * D;
    const float* kg = p.k + (int64_t)b * p.S * p.ldk + hd * D;
    const float* vg = p.v + (int64_t)b * p.S * p.ldv + hd * D;
    float* og = p.out + (int64_t)b * p.L * p.ldo + hd * D;
    stage_f32(Ks, kg, p.ldk, 0, Sp, p.S, D, STR, tid, (int)blockDim.x);
    stage_f32(Vs, vg, p.ldv, 0, Sp, p.S, D, STR, tid, (int)blockDim.x);
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    const int nkt = Sp >> 5;
    for (int qt = wave; qt * 32 < p.L; qt += (int)(blockDim.x >> 6)) {
        const int l = qt * 32 + fr;
        const bool lok = l < p.L;
        float qf[HD];
#pragma unroll
        for (int s4 = 0; s4 < HD / 4; ++s4) {
            f32x4 v = *reinterpret_cast<const f32x4*>(qg + (int64_t)(lok ? l : 0) * p.ldq + fh * HD + 4 * s4);
#pragma unroll
            for (int e = 0; e < 4; ++e) qf[4 * s4 + e] = lok ? v[e] : 0.f;
        }
        f32x16 oacc[NDT];
#pragma unroll
        for (int t = 0; t < NDT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[t][e] = 0.f;
        float m = -INFINITY, lsum = 0.f;
        for (int kt = 0; kt < nkt; ++kt) {
            f32x16 st;
#pragma unroll
            for (int e = 0; e < 16; ++e) st[e] = 0.f;
            st = dot_rows<HD>(Ks, STR, kt * 32 + fr, fh, qf, st);
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + acc_row32(r, fh);
                st[r] = key < p.S ? st[r] * p.scale : -INFINITY;
                mx = fmaxf(mx, st[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m, mx);
            const float alpha = __expf(m - mn);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = __expf(st[r] - mn); ps += st[r]; }
            lsum = lsum * alpha + ps;
            m = mn;
            if (p.drop_thresh) {
                const uint64_t prow = ((uint64_t)bh * p.L + l) * (uint64_t)((p.S + 1) >> 1);
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int key = kt * 32 + acc_row32(r, fh);
                    const uint32_t hh = drop_hash_pair(eff_seed(p.seed, p.seed_dev), prow + (uint32_t)(key >> 1));
                    st[r] = (lok && key < p.S && drop_keep_half(hh, 0, p.drop_thresh)) ? st[r] * p.drop_scale : 0.f;
                    st[r + 1] = (lok && key + 1 < p.S && drop_keep_half(hh, 1, p.drop_thresh)) ? st[r + 1] * p.drop_scale : 0.f;
                }
            }
#pragma unroll
            for (int t = 0; t < NDT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[t][e] *= alpha;
            acc_cols<NDT>(Vs, STR, kt * 32, fr, fh, st, oacc);
        }
        const float ltot = lsum + __shfl_xor(lsum, 32, 64);
        const float inv = 1.0f / ltot;
        if (lok) {
            if (fh == 0) p.lse[(int64_t)bh * p.L + l] = m + __logf(ltot);
#pragma unroll
            for (int t = 0; t < NDT; ++t)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int dd = t * 32 + 8 * g4 + 4 * fh;
                    if (dd < D) {
                        f32x4 v = {oacc[t][4 * g4] * inv, oacc[t][4 * g4 + 1] * inv, oacc[t][4 * g4 + 2] * inv, oacc[t][4 * g4 + 3] * inv};
                        *reinterpret_cast<f32x4*>(og + (int64_t)l * p.ldo + dd) = v;
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------ dQ (query on the lane)
template <int ND>
__global__ __launch_bounds__(256, 1) void attn_dq_f32(const AttnFP p) {
    constexpr int D = ND * 16, HD = D / 2, NDT = (D + 31) / 32, STR = D * 4 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Sp = (p.S + 31) & ~31;
    char* Ks = smem;
    char* Vs = smem + Sp * STR;
    const int bh = blockIdx.x, b = bh / p.H, hd = bh - b * p.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* qg = p.q + (int64_t)b * p.L * p.ldq + hd * D;
    const float* kg = p.k + (int64_t)b * p.S * p.ldk + hd * D;
    const float* vg = p.v + (int64_t)b * p.S * p.ldv + hd * D;
    const float* gg = p.dout + (int64_t)b * p.L * p.lddo + hd * D;
    const float* og = p.o + (int64_t)b * p.L * p.ldo + hd * D;
    float* dqg = p.dq + (int64_t)b * p.L * p.lddq + hd * D;
    stage_f32(Ks, kg, p.ldk, 0, Sp, p.S, D, STR, tid, (int)blockDim.x);
    stage_f32(Vs, vg, p.ldv, 0, Sp, p.S, D, STR, tid, (int)blockDim.x);
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    const int nkt = Sp >> 5;
    for (int qt = wave; qt * 32 < p.L; qt += (int)(blockDim.x >> 6)) {
        const int l = qt * 32 + fr;
        const bool lok = l < p.L;
        const int ls = lok ? l : 0;
        float qf[HD], gf[HD];
        float del = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < HD / 4; ++s4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(qg + (int64_t)ls * p.ldq + fh * HD + 4 * s4);
            const f32x4 w = *reinterpret_cast<const f32x4*>(gg + (int64_t)ls * p.lddo + fh * HD + 4 * s4);
            const f32x4 ov = *reinterpret_cast<const f32x4*>(og + (int64_t)ls * p.ldo + fh * HD + 4 * s4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                qf[4 * s4 + e] = lok ? v[e] : 0.f;
                gf[4 * s4 + e] = lok ? w[e] : 0.f;
                del += (lok ? w[e] : 0.f) * ov[e];                      // delta_l = sum_d dO[l][d] * O[l][d]
            }
        }
        const float lse = lok ? p.lse[(int64_t)bh * p.L + l] : 0.f;
        del += __shfl_xor(del, 32, 64);
        if (!lok) del = 0.f;
        if (lok && fh == 0) p.delta[(int64_t)bh * p.L + l] = del;
        f32x16 dacc[NDT];
#pragma unroll
        for (int t = 0; t < NDT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) dacc[t][e] = 0.f;
        for (int kt = 0; kt < nkt; ++kt) {
            f32x16 st, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { st[e] = 0.f; dp[e] = 0.f; }
            st = dot_rows<HD>(Ks, STR, kt * 32 + fr, fh, qf, st);
            dp = dot_rows<HD>(Vs, STR, kt * 32 + fr, fh, gf, dp);
            const uint64_t prow = ((uint64_t)bh * p.L + l) * (uint64_t)((p.S + 1) >> 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + acc_row32(r, fh);
                float ds = 0.f;
                if (lok && key < p.S) {
                    const float pr = __expf(st[r] * p.scale - lse);
                    float dpr = dp[r];
                    if (p.drop_thresh) {
                        const uint32_t hh = drop_hash_pair(eff_seed(p.seed, p.seed_dev), prow + (uint32_t)(key >> 1));
                        dpr = drop_keep_half(hh, key & 1, p.drop_thresh) ? dpr * p.drop_scale : 0.f;
                    }
                    ds = pr * (dpr - del) * p.scale;
                }
                st[r] = ds;
            }
            acc_cols<NDT>(Ks, STR, kt * 32, fr, fh, st, dacc);          // dQ^T[dcol][q] += K^T[dcol][key] dS^T[key][q]
        }
        if (lok) {
#pragma unroll
            for (int t = 0; t < NDT; ++t)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int dd = t * 32 + 8 * g4 + 4 * fh;
                    if (dd < D) {
                        f32x4 v = {dacc[t][4 * g4], dacc[t][4 * g4 + 1], dacc[t][4 * g4 + 2], dacc[t][4 * g4 + 3]};
                        *reinterpret_cast<f32x4*>(dqg + (int64_t)l * p.lddq + dd) = v;
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------ dK, dV (key on the lane)
template <int ND>
__global__ __launch_bounds__(256, 1) void attn_dkv_f32(const AttnFP p) {
    constexpr int D = ND * 16, HD = D / 2, NDT = (D + 31) / 32, STR = D * 4 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Lp = (p.L + 31) & ~31;
    char* Qs = smem;
    char* Gs = Qs + Lp * STR;
    // lse[l] / delta[l] live in the 16-byte pad slot of row l of the Q image (bytes 4D .. 4D+7)
    const int bh = blockIdx.x, b = bh / p.H, hd = bh - b * p.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* qg = p.q + (int64_t)b * p.L * p.ldq + hd * D;
    const float* kg = p.k + (int64_t)b * p.S * p.ldk + hd * D;
    const float* vg = p.v + (int64_t)b * p.S * p.ldv + hd * D;
    const float* gg = p.dout + (int64_t)b * p.L * p.lddo + hd * D;
    float* dkg = p.dk + (int64_t)b * p.S * p.lddk + hd * D;
    float* dvg = p.dv + (int64_t)b * p.S * p.lddv + hd * D;
    stage_f32(Qs, qg, p.ldq, 0, Lp, p.L, D, STR, tid, (int)blockDim.x);
    stage_f32(Gs, gg, p.lddo, 0, Lp, p.L, D, STR, tid, (int)blockDim.x);
    for (int i = tid; i < Lp; i += (int)blockDim.x) {
        float* pad = reinterpret_cast<float*>(Qs + i * STR + 4 * D);
        pad[0] = i < p.L ? p.lse[(int64_t)bh * p.L + i] : 0.f;
        pad[1] = i < p.L ? p.delta[(int64_t)bh * p.L + i] : 0.f;
    }
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    const int Sp = (p.S + 31) & ~31;
    const int nqt = Lp >> 5;
    for (int kt = wave; kt * 32 < Sp; kt += (int)(blockDim.x >> 6)) {
        const int key = kt * 32 + fr;
        const bool kok = key < p.S;
        const int kr = kok ? key : 0;
        float kreg[HD], vreg[HD];
#pragma unroll
        for (int s4 = 0; s4 < HD / 4; ++s4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(kg + (int64_t)kr * p.ldk + fh * HD + 4 * s4);
            const f32x4 c = *reinterpret_cast<const f32x4*>(vg + (int64_t)kr * p.ldv + fh * HD + 4 * s4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { kreg[4 * s4 + e] = kok ? a[e] : 0.f; vreg[4 * s4 + e] = kok ? c[e] : 0.f; }
        }
        f32x16 dk[NDT], dv[NDT];
#pragma unroll
        for (int t = 0; t < NDT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) { dk[t][e] = 0.f; dv[t][e] = 0.f; }
        for (int qt = 0; qt < nqt; ++qt) {
            f32x16 st, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { st[e] = 0.f; dp[e] = 0.f; }
            st = dot_rows<HD>(Qs, STR, qt * 32 + fr, fh, kreg, st);     // S[q][key]
            dp = dot_rows<HD>(Gs, STR, qt * 32 + fr, fh, vreg, dp);     // dP[q][key]
            f32x16 pd;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int l = qt * 32 + acc_row32(r, fh);
                float pv = 0.f, ds = 0.f;
                if (kok && l < p.L) {
                    const float2 ld2 = *reinterpret_cast<const float2*>(Qs + l * STR + 4 * D);      // (lse, delta) of query l
                    const float pr = __expf(st[r] * p.scale - ld2.x);
                    float dpr = dp[r];
                    pv = pr;
                    if (p.drop_thresh) {
                        const uint32_t hh = drop_hash_pair(eff_seed(p.seed, p.seed_dev), ((uint64_t)bh * p.L + l) * (uint64_t)((p.S + 1) >> 1) + (uint32_t)(key >> 1));
                        const bool keep = drop_keep_half(hh, key & 1, p.drop_thresh);
                        pv = keep ? pr * p.drop_scale : 0.f;
                        dpr = keep ? dpr * p.drop_scale : 0.f;
                    }
                    ds = pr * (dpr - ld2.y) * p.scale;
                }
                pd[r] = pv; st[r] = ds;
            }
            acc_cols<NDT>(Gs, STR, qt * 32, fr, fh, pd, dv);            // dV^T[dcol][key] += dO^T[dcol][q] P[q][key]
            acc_cols<NDT>(Qs, STR, qt * 32, fr, fh, st, dk);            // dK^T[dcol][key] += Q^T[dcol][q] dS[q][key]
        }
        if (kok) {
#pragma unroll
            for (int t = 0; t < NDT; ++t)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int dd = t * 32 + 8 * g4 + 4 * fh;
                    if (dd < D) {
                        f32x4 a = {dk[t][4 * g4], dk[t][4 * g4 + 1], dk[t][4 * g4 + 2], dk[t][4 * g4 + 3]};
                        f32x4 c = {dv[t][4 * g4], dv[t][4 * g4 + 1], dv[t][4 * g4 + 2], dv[t][4 * g4 + 3]};
                        *reinterpret_cast<f32x4*>(dkg + (int64_t)key * p.lddk + dd) = a;
                        *reinterpret_cast<f32x4*>(dvg + (int64_t)key * p.lddv + dd) = c;
                    }
                }
        }
    }
}

constexpr size_t F32_LDS_CAP = 158 * 1024;
static inline int f32_waves(int rows) { const int t = (rows + 31) / 32; return t < 1 ? 1 : (t > 4 ? 4 : t); }
static inline size_t f32_lds(int rows, int d) { return 2 * (size_t)((rows + 31) & ~31) * ((size_t)d * 4 + 16) + 128; }   // (+ slack: the last 32-column tile reads past d)
static inline int f32_nd(int d) { return (d == 16 || d == 32 || d == 64 || d == 80 || d == 160) ? d / 16 : 0; }

// returns 1 when the exact-fp32 MFMA kernels handle this problem (forward AND backward: one predicate for both)
int attn_f32_applicable(int L, int S, int d, int dtype, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo) {
    if (dtype != PSG_F32 || !f32_nd(d)) return 0;
    if (((ldq | ldk | ldv | ldo) & 3) != 0) return 0;                    // 16-byte row fragments
    if (f32_lds(S, d) > F32_LDS_CAP || f32_lds(L, d) > F32_LDS_CAP) return 0;
    return 1;
}

#define F32_DISPATCH(KERNEL, ...)                                        \
    switch (p.d) {                                                       \
        case 16: hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__); break;      \
        case 32: hipLaunchKernelGGL(KERNEL<2>, __VA_ARGS__); break;      \
        case 64: hipLaunchKernelGGL(KERNEL<4>, __VA_ARGS__); break;      \
        case 80: hipLaunchKernelGGL(KERNEL<5>, __VA_ARGS__); break;      \
        default: hipLaunchKernelGGL(KERNEL<10>, __VA_ARGS__); break;     \
    }

int attn_f32_init_attrs() {
#define SET_LDS(K) PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)F32_LDS_CAP))
    SET_LDS(attn_fwd_f32<1>); SET_LDS(attn_fwd_f32<2>); SET_LDS(attn_fwd_f32<4>); SET_LDS(attn_fwd_f32<5>); SET_LDS(attn_fwd_f32<10>);
    SET_LDS(attn_dq_f32<1>); SET_LDS(attn_dq_f32<2>); SET_LDS(attn_dq_f32<4>); SET_LDS(attn_dq_f32<5>); SET_LDS(attn_dq_f32<10>);
    SET_LDS(attn_dkv_f32<1>); SET_LDS(attn_dkv_f32<2>); SET_LDS(attn_dkv_f32<4>); SET_LDS(attn_dkv_f32<5>); SET_LDS(attn_dkv_f32<10>);
#undef SET_LDS
    return PSG_OK;
}

int attn_f32_fwd(const AttnFP& p, hipStream_t s) {
    F32_DISPATCH(attn_fwd_f32, dim3(p.B * p.H), dim3(64 * f32_waves(p.L)), f32_lds(p.S, p.d), s, p);
    PSG_LAUNCH_CHECK("attn_fwd_f32");
    return PSG_OK;
}
int attn_f32_bwd(const AttnFP& p, hipStream_t s) {
    F32_DISPATCH(attn_dq_f32, dim3(p.B * p.H), dim3(64 * f32_waves(p.L)), f32_lds(p.S, p.d), s, p);
    PSG_LAUNCH_CHECK("attn_dq_f32");
    F32_DISPATCH(attn_dkv_f32, dim3(p.B * p.H), dim3(64 * f32_waves(p.S)), f32_lds(p.L, p.d), s, p);
    PSG_LAUNCH_CHECK("attn_dkv_f32");
    return PSG_OK;
}

}  // namespace psg
