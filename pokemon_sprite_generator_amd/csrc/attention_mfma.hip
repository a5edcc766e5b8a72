// bf16 multi-head attention core on the matrix cores (v_mfma_f32_32x32x16_bf16), fused
// QK^T -> softmax -> PV per (sample, head), plus the two backward kernels (dQ; dK+dV).
//
// One workgroup of up to 4 waves (one per 32-row tile, see attn_waves) per (b, head).  K and V of the head ([S][d], S <= 224 here)
// are staged ONCE into LDS in their natural row-major layout (row stride 2*d+16 bytes: an odd
// number of 16-byte slots, so ds_read_b128 row fragments are conflict-free; when d % 32 == 16 the
// last 32-column transposed read of a row runs 16 columns into the pad and the next row - those
// operand rows only feed output rows >= d, which are never stored; 32 bytes of slack end the image).  Each wave owns
// 32-query tiles:
//   S^T[key][q] = K . Q^T      MFMA A = K rows (ds_read_b128), B = Q^T (registers, straight from HBM)
// so every lane holds ONE query column: the row softmax is an in-lane reduction over the 16
// accumulator registers plus one cross-half __shfl_xor(32) — no LDS, no serial lanes.  The
// exponentiated tile is converted to bf16 in registers and fed back as the B operand of
//   O^T[d][q] += V^T . P^T     MFMA A = V^T fragments via ds_read_b64_tr_b16 (hardware transpose)
// using the accumulator-as-operand k-order of the CDNA4 guide (element j of lane-half h is key
// 16s + 8(j>>2) + 4h + (j&3)); the transposed reads pick the same key order.  Online softmax
// (running max / sum per lane) over key tiles of 32.  Backward recomputes P from the saved
// log-sum-exp: the dQ kernel has the same query-on-lane structure (dS^T is directly the B
// operand of dQ^T += K^T . dS^T); the dK/dV kernel puts the KEY on the lane so that P and dS
// are directly the B operands of dV^T += dO^T . P and dK^T += Q^T . dS.
// Dropout: a stateless 16-bit-per-element hash shared by the two keys of a pair (psg_common.h: drop_hash_pair),
// regenerated identically in both backward kernels.
#include "psg_common.h"

#ifndef PSG_STAGE_U
#define PSG_STAGE_U 10      // 16-byte loads in flight per lane while staging (8 / 10 / 12 / 20 measured: 10 takes the d = 160 dK/dV kernel from 70 to 60 us, the others within 2 %)
#endif
namespace psg {

struct AttnMP {     // mirrored in attention.hip
    const bf16_t *q, *k, *v, *o, *dout;
    bf16_t *out, *dq, *dk, *dv;
    float* lse; float* delta;                      // delta = rowsum(dO * O): written by the dQ kernel, read by dK/dV
    int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    int B, H, L, S, d;
    float scale;
    uint32_t drop_thresh; float drop_scale; uint64_t seed;
    const uint64_t* seed_dev;
};

typedef __attribute__((ext_vector_type(8))) short s16x8;
#define LDS_TR(ptr) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ptr))

__device__ __forceinline__ bf16x8 cat_tr(s16x4 a, s16x4 b) {
    s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return *reinterpret_cast<bf16x8*>(&v);
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& x, int s2) {   // registers 8*s2 .. 8*s2+7 -> bf16x8
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16_t)x[8 * s2 + j];
    return r;
}
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// stage rows [0, nrows) of a [*, d] bf16 matrix into LDS rows of `stride` bytes, zero-filling rows >= nvalid
// and columns [d, d32)
__device__ __forceinline__ void stage_tile(char* dst, const bf16_t* src, int64_t ld, int row0, int nrows, int nvalid_end,
                                           int d, int d32, int stride, int tid, int nthreads) {
    const int cpr = d32 >> 3;                      // 16-byte chunks per row
    const int n = nrows * cpr;
    constexpr int U = PSG_STAGE_U;                          // independent 16-byte loads in flight per lane
    for (int e0 = tid; e0 < n; e0 += nthreads * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * nthreads;
            const int r = e / cpr, c = e - r * cpr;
            const int gr = row0 + r;
            const bool ok = e < n && gr < nvalid_end && c * 8 < d;
            const uint4 t = *reinterpret_cast<const uint4*>(src + (ok ? (int64_t)gr * ld + c * 8 : 0));   // unconditional load (no per-load branch + wait)
            v[u] = ok ? t : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * nthreads;
            const int r = e / cpr, c = e - r * cpr;
            if (e < n) *reinterpret_cast<uint4*>(dst + r * stride + c * 16) = v[u];
        }
    }
}

// transposed A-operand fragment for k-step s2 of a 32-row tile starting at row `row0` of an LDS image
// [rows][stride]: output tile rows = columns colbase .. colbase+31 of the image; element j <-> image row
// row0 + 16*s2 + 8*(j>>2) + 4*h + (j&3)
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int stride, int row0, int s2, int colbase, int lane) {
    const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
    const int h = g >> 1;
    const char* base = img + (row0 + 16 * s2 + 4 * h + q4) * stride + (colbase + (g & 1) * 16 + 4 * p4) * 2;
    s16x4 a = LDS_TR(base);
    s16x4 b = LDS_TR(base + 8 * stride);
    return cat_tr(a, b);
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int ND>
__global__ __launch_bounds__(256, (ND <= 5 ? 2 : 1)) void attn_fwd_mfma(const AttnMP p) {
    constexpr int NDT = (ND + 1) / 2;
    constexpr int D32 = NDT * 32;
    constexpr int STR = 2 * ND * 16 + 16;      // row = d values + one 16-byte slot (odd slot count: conflict-free row reads)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Sp = (p.S + 31) & ~31;
    char* Ks = smem;
    char* Vs = smem + Sp * STR;
    const int bh = blockIdx.x, b = bh / p.H, hd = bh - b * p.H;
    const uint64_t seed = eff_seed(p.seed, p.drop_thresh ? p.seed_dev : nullptr);     // wave-uniform
    const int d = ND * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16_t* qg = p.q + (int64_t)b * p.L * p.ldq + hd * d;
    const bf16_t* kg = p.k + (int64_t)b * p.S * p.ldk + hd * d;
    const bf16_t* vg = p.v + (int64_t)b * p.S * p.ldv + hd * d;
    bf16_t* og = p.out + (int64_t)b * p.L * p.ldo + hd * d;
    stage_tile(Ks, kg, p.ldk, 0, Sp, p.S, d, d, STR, tid, (int)blockDim.x);
    stage_tile(Vs, vg, p.ldv, 0, Sp, p.S, d, d, STR, tid, (int)blockDim.x);
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    const int nkt = Sp >> 5;
    for (int qt = wave; qt * 32 < p.L; qt += (int)(blockDim.x >> 6)) {
        const int l = qt * 32 + fr;
        const bool lok = l < p.L;
        bf16x8 qf[ND];
#pragma unroll
        for (int ks = 0; ks < ND; ++ks) {
            uint4 v = *reinterpret_cast<const uint4*>(qg + (int64_t)(lok ? l : 0) * p.ldq + 16 * ks + 8 * fh);   // unconditional
            if (!lok) v = make_uint4(0, 0, 0, 0);
            qf[ks] = *reinterpret_cast<bf16x8*>(&v);
        }
        f32x16 oacc[NDT];
#pragma unroll
        for (int t = 0; t < NDT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[t][e] = 0.f;
        float m = -INFINITY, lsum = 0.f;
        const DropRow dr = drop_row(seed, ((uint64_t)bh * p.L + l) * (uint64_t)((p.S + 1) >> 1));
        const uint32_t t16 = p.drop_thresh >> 16;
        for (int kt = 0; kt < nkt; ++kt) {
            f32x16 st;
#pragma unroll
            for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < ND; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (kt * 32 + fr) * STR + (16 * ks + 8 * fh) * 2);
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st, 0, 0, 0);
            }
            float mx = -INFINITY;
            if (kt * 32 + 32 <= p.S) {                     // (wave-uniform) a full key tile: nothing to mask
#pragma unroll
                for (int r = 0; r < 16; ++r) { st[r] = st[r] * p.scale; mx = fmaxf(mx, st[r]); }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 32 + acc_row(r, fh);
                    st[r] = key < p.S ? st[r] * p.scale : -INFINITY;
                    mx = fmaxf(mx, st[r]);
                }
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
                // Branch-free (as in the dQ kernel): an absent key's probability is already exp(-inf) = 0 and an absent query's
                // column is never stored, so neither needs a test here.
#pragma unroll
                for (int r = 0; r < 16; r += 2) {          // registers r, r+1 hold keys 2k, 2k+1: one hash for both
                    const uint32_t hh = drop_hash_row(dr, (uint32_t)(kt * 16 + (acc_row(r, fh) >> 1)));
                    st[r] = (hh & 0xFFFFu) >= t16 ? st[r] * p.drop_scale : 0.f;
                    st[r + 1] = (hh >> 16) >= t16 ? st[r + 1] * p.drop_scale : 0.f;
                }
            }
#pragma unroll
            for (int t = 0; t < NDT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[t][e] *= alpha;
            const bf16x8 pf0 = pack8(st, 0), pf1 = pack8(st, 1);
#pragma unroll
            for (int t = 0; t < NDT; ++t) {
                const bf16x8 v0 = tr_frag(Vs, STR, kt * 32, 0, t * 32, lane);
                const bf16x8 v1 = tr_frag(Vs, STR, kt * 32, 1, t * 32, lane);
                oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pf0, oacc[t], 0, 0, 0);
                oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pf1, oacc[t], 0, 0, 0);
            }
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
                    if (dd < d) {
                        f32x4 v = {oacc[t][4 * g4] * inv, oacc[t][4 * g4 + 1] * inv, oacc[t][4 * g4 + 2] * inv, oacc[t][4 * g4 + 3] * inv};
                        store4<bf16_t>(og + (int64_t)l * p.ldo + dd, v);
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dQ: same query-on-lane structure
// ------------------------------------------------------------------------------------------------
template <int ND>
__global__ __launch_bounds__(256, (ND <= 5 ? 2 : 1)) void attn_dq_mfma(const AttnMP p) {
    constexpr int NDT = (ND + 1) / 2;
    constexpr int D32 = NDT * 32;
    constexpr int STR = 2 * ND * 16 + 16;      // row = d values + one 16-byte slot (odd slot count: conflict-free row reads)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Sp = (p.S + 31) & ~31;
    char* Ks = smem;
    char* Vs = smem + Sp * STR;
    const int bh = blockIdx.x, b = bh / p.H, hd = bh - b * p.H;
    const uint64_t seed = eff_seed(p.seed, p.drop_thresh ? p.seed_dev : nullptr);     // wave-uniform
    const int d = ND * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16_t* qg = p.q + (int64_t)b * p.L * p.ldq + hd * d;
    const bf16_t* kg = p.k + (int64_t)b * p.S * p.ldk + hd * d;
    const bf16_t* vg = p.v + (int64_t)b * p.S * p.ldv + hd * d;
    const bf16_t* gg = p.dout + (int64_t)b * p.L * p.lddo + hd * d;
    const bf16_t* og = p.o + (int64_t)b * p.L * p.ldo + hd * d;
    bf16_t* dqg = p.dq + (int64_t)b * p.L * p.lddq + hd * d;
    stage_tile(Ks, kg, p.ldk, 0, Sp, p.S, d, d, STR, tid, (int)blockDim.x);
    stage_tile(Vs, vg, p.ldv, 0, Sp, p.S, d, d, STR, tid, (int)blockDim.x);
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    const int nkt = Sp >> 5;
    for (int qt = wave; qt * 32 < p.L; qt += (int)(blockDim.x >> 6)) {
        const int l = qt * 32 + fr;
        const bool lok = l < p.L;
        bf16x8 qf[ND], gf[ND];
#pragma unroll
        for (int ks = 0; ks < ND; ++ks) {
            const int ls = lok ? l : 0;
            uint4 v = *reinterpret_cast<const uint4*>(qg + (int64_t)ls * p.ldq + 16 * ks + 8 * fh);    // unconditional loads
            uint4 w = *reinterpret_cast<const uint4*>(gg + (int64_t)ls * p.lddo + 16 * ks + 8 * fh);
            if (!lok) { v = make_uint4(0, 0, 0, 0); w = make_uint4(0, 0, 0, 0); }
            qf[ks] = *reinterpret_cast<bf16x8*>(&v);
            gf[ks] = *reinterpret_cast<bf16x8*>(&w);
        }
        const float lse = lok ? p.lse[(int64_t)bh * p.L + l] : 0.f;
        // delta_l = sum_d dO[l][d] * O[l][d]: this lane already holds its half of the dO row; O comes the same way
        float del = 0.f;
#pragma unroll
        for (int ks = 0; ks < ND; ++ks) {
            const uint4 ov = *reinterpret_cast<const uint4*>(og + (int64_t)(lok ? l : 0) * p.ldo + 16 * ks + 8 * fh);
            const bf16x8 of = *reinterpret_cast<const bf16x8*>(&ov);
#pragma unroll
            for (int j = 0; j < 8; ++j) del += (float)gf[ks][j] * (float)of[j];
        }
        del += __shfl_xor(del, 32, 64);
        if (!lok) del = 0.f;
        if (lok && fh == 0) p.delta[(int64_t)bh * p.L + l] = del;
        f32x16 dacc[NDT];
#pragma unroll
        for (int t = 0; t < NDT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) dacc[t][e] = 0.f;
        const DropRow dr = drop_row(seed, ((uint64_t)bh * p.L + l) * (uint64_t)((p.S + 1) >> 1));
        const uint32_t t16 = p.drop_thresh >> 16;
        for (int kt = 0; kt < nkt; ++kt) {
            f32x16 st, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { st[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < ND; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (kt * 32 + fr) * STR + (16 * ks + 8 * fh) * 2);
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vs + (kt * 32 + fr) * STR + (16 * ks + 8 * fh) * 2);
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, gf[ks], dp, 0, 0, 0);
            }
            // Branch-free: a key >= S meets a zero K row in dQ^T += K^T . dS^T, a query >= L is a column that is never
            // stored, and every value here is finite either way (lse = del = 0 for absent queries, scores of absent keys 0).
            if (p.drop_thresh) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {          // registers r, r+1 hold keys 2k, 2k+1: one hash for both
                    const uint32_t hh = drop_hash_row(dr, (uint32_t)(kt * 16 + (acc_row(r, fh) >> 1)));
                    dp[r] = (hh & 0xFFFFu) >= t16 ? dp[r] * p.drop_scale : 0.f;
                    dp[r + 1] = (hh >> 16) >= t16 ? dp[r + 1] * p.drop_scale : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pr = __expf(st[r] * p.scale - lse);
                st[r] = pr * (dp[r] - del) * p.scale;
            }
            const bf16x8 sf0 = pack8(st, 0), sf1 = pack8(st, 1);
#pragma unroll
            for (int t = 0; t < NDT; ++t) {
                const bf16x8 k0 = tr_frag(Ks, STR, kt * 32, 0, t * 32, lane);
                const bf16x8 k1 = tr_frag(Ks, STR, kt * 32, 1, t * 32, lane);
                dacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, sf0, dacc[t], 0, 0, 0);
                dacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, sf1, dacc[t], 0, 0, 0);
            }
        }
        if (lok) {
#pragma unroll
            for (int t = 0; t < NDT; ++t)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int dd = t * 32 + 8 * g4 + 4 * fh;
                    if (dd < d) {
                        f32x4 v = {dacc[t][4 * g4], dacc[t][4 * g4 + 1], dacc[t][4 * g4 + 2], dacc[t][4 * g4 + 3]};
                        store4<bf16_t>(dqg + (int64_t)l * p.lddq + dd, v);
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dK, dV: key on the lane.  Q and dO of the whole head live in LDS; each wave stages its own 32-key
// K/V tile (row reads for the B operands) into a private LDS region.
// ------------------------------------------------------------------------------------------------
template <int ND>
__global__ __launch_bounds__(256, (ND <= 5 ? 2 : 1)) void attn_dkv_mfma(const AttnMP p) {
    constexpr int NDT = (ND + 1) / 2;
    constexpr int D32 = NDT * 32;
    constexpr int STR = 2 * ND * 16 + 16;      // row = d values + one 16-byte slot (odd slot count: conflict-free row reads)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool KV_REG = ND <= 5;                          // K/V B-operands straight from HBM into registers
    const int Lp = (p.L + 31) & ~31;
    char* Qs = smem;
    char* Gs = Qs + Lp * STR;
    char* KVw = Gs + Lp * STR + 32;                          // (ND > 5) per wave: K tile [32][STR] | V tile [32][STR]
    // lse[l] / delta[l] live in the 16-byte pad slot of row l of the Q image (bytes 2d .. 2d+7)
    const int bh = blockIdx.x, b = bh / p.H, hd = bh - b * p.H;
    const uint64_t seed = eff_seed(p.seed, p.drop_thresh ? p.seed_dev : nullptr);     // wave-uniform
    const int d = ND * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16_t* qg = p.q + (int64_t)b * p.L * p.ldq + hd * d;
    const bf16_t* kg = p.k + (int64_t)b * p.S * p.ldk + hd * d;
    const bf16_t* vg = p.v + (int64_t)b * p.S * p.ldv + hd * d;
    const bf16_t* gg = p.dout + (int64_t)b * p.L * p.lddo + hd * d;
    bf16_t* dkg = p.dk + (int64_t)b * p.S * p.lddk + hd * d;
    bf16_t* dvg = p.dv + (int64_t)b * p.S * p.lddv + hd * d;
    stage_tile(Qs, qg, p.ldq, 0, Lp, p.L, d, d, STR, tid, (int)blockDim.x);
    stage_tile(Gs, gg, p.lddo, 0, Lp, p.L, d, d, STR, tid, (int)blockDim.x);
    for (int i = tid; i < Lp; i += (int)blockDim.x) {
        float* pad = reinterpret_cast<float*>(Qs + i * STR + 2 * d);
        pad[0] = i < p.L ? p.lse[(int64_t)bh * p.L + i] : 0.f;
        pad[1] = i < p.L ? p.delta[(int64_t)bh * p.L + i] : 0.f;
    }
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    const int Sp = (p.S + 31) & ~31;
    const int nqt = Lp >> 5;
    char* Kw = KVw + wave * (2 * 32 * STR);
    char* Vw = Kw + 32 * STR;
    // Waves = KW key tiles x QW query groups.  With more waves than key tiles (cross-attention: S = 32 is ONE key tile, and the
    // whole workgroup used to be one wave walking all query tiles) the query tiles are dealt to QW waves per key tile; their
    // partial dK / dV meet in LDS - the Q / dO images are dead by then - and query group 0 adds them in fixed order.
    // QW > 1 only when every wave has exactly one key tile (KW == number of key tiles): the barriers below are uniform.
    const int W = (int)(blockDim.x >> 6), nkt_all = Sp >> 5;
    const int KW = min(nkt_all, W), QW = max(1, W / KW);
    const int kw = wave % KW, qw = wave / KW;
    const bool active = qw < QW;
    for (int kt = kw; kt * 32 < Sp; kt += KW) {
        const int key = kt * 32 + fr;
        const bool kok = key < p.S;
        bf16x8 kreg[KV_REG ? ND : 1], vreg[KV_REG ? ND : 1];
        if constexpr (KV_REG) {
#pragma unroll
            for (int ks = 0; ks < ND; ++ks) {
                const int kr = kok ? key : 0;
                uint4 a = *reinterpret_cast<const uint4*>(kg + (int64_t)kr * p.ldk + 16 * ks + 8 * fh);     // unconditional loads
                uint4 c = *reinterpret_cast<const uint4*>(vg + (int64_t)kr * p.ldv + 16 * ks + 8 * fh);
                if (!kok) { a = make_uint4(0, 0, 0, 0); c = make_uint4(0, 0, 0, 0); }
                kreg[ks] = *reinterpret_cast<bf16x8*>(&a);
                vreg[ks] = *reinterpret_cast<bf16x8*>(&c);
            }
        } else {
            // private K/V tile (only this wave touches Kw/Vw: wave-level ordering suffices)
            stage_tile(Kw, kg, p.ldk, kt * 32, 32, p.S, d, d, STR, lane, 64);
            stage_tile(Vw, vg, p.ldv, kt * 32, 32, p.S, d, d, STR, lane, 64);
            __builtin_amdgcn_s_waitcnt(0);     // all counters: the wave's own LDS writes have landed
            __builtin_amdgcn_wave_barrier();
        }
        // head_dim 320: the 2 x 10 output tiles of dK and dV (320 accumulator registers) do not fit next to the score
        // tiles, so the head dimension is produced in two halves, each recomputing the (cheap) score / dP tiles
        constexpr int NH = ND > 10 ? 2 : 1, TH = NDT / NH;
        static_assert(NDT % NH == 0, "head-dim halves");
#pragma unroll 1
        for (int hp = 0; hp < NH; ++hp) {
        f32x16 dk[TH], dv[TH];
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) { dk[t][e] = 0.f; dv[t][e] = 0.f; }
        for (int qt = active ? qw : nqt; qt < nqt; qt += QW) {
            f32x16 st, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { st[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < ND; ++ks) {
                const int co = (16 * ks + 8 * fh) * 2;
                const bf16x8 qa = *reinterpret_cast<const bf16x8*>(Qs + (qt * 32 + fr) * STR + co);     // A: Q rows
                const bf16x8 ga = *reinterpret_cast<const bf16x8*>(Gs + (qt * 32 + fr) * STR + co);     // A: dO rows
                bf16x8 kb, vb;                                                                           // B: K^T, V^T (key on lane)
                if constexpr (KV_REG) { kb = kreg[ks]; vb = vreg[ks]; }
                else {
                    kb = *reinterpret_cast<const bf16x8*>(Kw + fr * STR + co);
                    vb = *reinterpret_cast<const bf16x8*>(Vw + fr * STR + co);
                }
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kb, st, 0, 0, 0);    // S[q][key]
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, vb, dp, 0, 0, 0);    // dP[q][key]
            }
            // Per element this loop is the kernel's time (49 tile pairs x 16 elements per lane at L = 196: the vector ALU, not
            // memory, bounds it), so it is branch-free and hashes along the query row: an absent query (l >= L) has zero Q / dO
            // rows and (lse, delta) = 0 in the image - its P is finite and meets zero operands; an absent key's column is
            // never stored.  Mask index (bh L + l) ceil(S/2) + key/2 = row base of the tile + (acc_row ceil(S/2) + key/2).
            f32x16 pd;
            const uint32_t hS = (uint32_t)((p.S + 1) >> 1), koff = (uint32_t)(key >> 1);
            const DropRow dr = drop_row(seed, ((uint64_t)bh * p.L + (uint64_t)(qt * 32)) * (uint64_t)hS);
            const uint32_t t16 = p.drop_thresh >> 16;
            const int odd = key & 1, hsh = odd ? 16 : 0;
            // The lanes of keys 2k, 2k+1 (lane ^ 1: same key pair, same query rows) need the SAME 16 pair hashes: each computes
            // eight (rows 8 odd ... 8 odd + 7 of the register order) and takes the other eight from its neighbour (DPP quad
            // permute [1,0,3,2]) - the hash is three integer multiplies, the exchange one move.
            uint32_t hmine[8], hpeer[8];
            if (p.drop_thresh) {                           // (wave-uniform)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    hmine[j] = drop_hash_row(dr, (uint32_t)(acc_row(j, fh) + 16 * odd) * hS + koff);   // acc_row(8 odd + j) = acc_row(j) + 16 odd
                    hpeer[j] = (uint32_t)__builtin_amdgcn_mov_dpp((int)hmine[j], 0xB1, 0xF, 0xF, false);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int l = qt * 32 + acc_row(r, fh);
                const float2 ld2 = *reinterpret_cast<const float2*>(Qs + l * STR + 2 * d);      // (lse, delta) of query l
                const float pr = __expf(st[r] * p.scale - ld2.x);
                float dpr = dp[r], pv = pr;
                if (p.drop_thresh) {                       // (wave-uniform)
                    const uint32_t hh = ((r >> 3) == odd) ? hmine[r & 7] : hpeer[r & 7];
                    const bool keep = ((hh >> hsh) & 0xFFFFu) >= t16;
                    pv = keep ? pr * p.drop_scale : 0.f;
                    dpr = keep ? dpr * p.drop_scale : 0.f;
                }
                pd[r] = pv; st[r] = pr * (dpr - ld2.y) * p.scale;
            }
            const bf16x8 pf0 = pack8(pd, 0), pf1 = pack8(pd, 1), sf0 = pack8(st, 0), sf1 = pack8(st, 1);
#pragma unroll
            for (int t = 0; t < TH; ++t) {
                const int tc = (hp * TH + t) * 32;
                const bf16x8 g0 = tr_frag(Gs, STR, qt * 32, 0, tc, lane);
                const bf16x8 g1 = tr_frag(Gs, STR, qt * 32, 1, tc, lane);
                dv[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g0, pf0, dv[t], 0, 0, 0);   // dV^T[d][key] += dO^T . P
                dv[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1, pf1, dv[t], 0, 0, 0);
                const bf16x8 q0 = tr_frag(Qs, STR, qt * 32, 0, tc, lane);
                const bf16x8 q1 = tr_frag(Qs, STR, qt * 32, 1, tc, lane);
                dk[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q0, sf0, dk[t], 0, 0, 0);   // dK^T[d][key] += Q^T . dS
                dk[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q1, sf1, dk[t], 0, 0, 0);
            }
        }
        if (QW > 1) {                            // (workgroup-uniform)
            __syncthreads();                     // every wave is done with the Q / dO images
            float* red = reinterpret_cast<float*>(smem);
            if (active && qw > 0) {
                float* dst = red + (int64_t)((qw - 1) * KW + kw) * (TH * 2 * 16 * 64) + lane;
#pragma unroll
                for (int t = 0; t < TH; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) { dst[((t * 2) * 16 + e) * 64] = dk[t][e]; dst[((t * 2 + 1) * 16 + e) * 64] = dv[t][e]; }
            }
            __syncthreads();
            if (qw == 0) {
                for (int q2 = 1; q2 < QW; ++q2) {
                    const float* src = red + (int64_t)((q2 - 1) * KW + kw) * (TH * 2 * 16 * 64) + lane;
#pragma unroll
                    for (int t = 0; t < TH; ++t)
#pragma unroll
                        for (int e = 0; e < 16; ++e) { dk[t][e] += src[((t * 2) * 16 + e) * 64]; dv[t][e] += src[((t * 2 + 1) * 16 + e) * 64]; }
                }
            }
        }
        if (kok && qw == 0) {
#pragma unroll
            for (int t = 0; t < TH; ++t)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int dd = (hp * TH + t) * 32 + 8 * g4 + 4 * fh;
                    if (dd < d) {
                        f32x4 a = {dk[t][4 * g4], dk[t][4 * g4 + 1], dk[t][4 * g4 + 2], dk[t][4 * g4 + 3]};
                        f32x4 c = {dv[t][4 * g4], dv[t][4 * g4 + 1], dv[t][4 * g4 + 2], dv[t][4 * g4 + 3]};
                        store4<bf16_t>(dkg + (int64_t)key * p.lddk + dd, a);
                        store4<bf16_t>(dvg + (int64_t)key * p.lddv + dd, c);
                    }
                }
        }
        }   // hp
        __builtin_amdgcn_wave_barrier();    // tile reads done before the next restage (same wave, program order)
    }
}

static inline int nd_supported(int d) { return d == 16 || d == 32 || d == 64 || d == 80 || d == 160 || d == 320; }
static inline size_t str_bytes(int d) { return 2 * (size_t)d + 16; }
static inline size_t fwd_lds_m(int S, int d) { return 2 * (size_t)((S + 31) & ~31) * str_bytes(d) + 32; }
// one wave per 32-row tile of the loop the waves share (queries for fwd/dQ, keys for dK/dV), at most 4: a 7x7 map
// (49 rows) gets 2 waves per workgroup instead of 4 half-idle ones, and twice the workgroups fit a CU
static inline int attn_waves(int rows) { const int t = (rows + 31) / 32; return t < 1 ? 1 : (t > 4 ? 4 : t); }
constexpr size_t MFMA_LDS_CAP = 158 * 1024;
static inline size_t dkv_lds_w(int L, int d, int waves) {
    const size_t Lp = (L + 31) & ~31;
    return 2 * Lp * str_bytes(d) + 32 + (d > 80 ? (size_t)waves * 2 * 32 * str_bytes(d) + 32 : 0);
}
// waves of the dK/dV kernel: one per 32-key tile, fewer when their private K/V tiles would not fit LDS next to the
// Q / dO images (head_dim 320 = the reference CLI's 4 heads at 1280 channels: 7x7 self-attention runs 1 wave)
static inline int dkv_waves(int L, int S, int d) {
    const int nkt = (S + 31) / 32, nqt = (L + 31) / 32;
    int kw = attn_waves(S);
    // query groups per key tile (round 3): only when the key tiles do not fill the workgroup (S = 32: one tile; 49: two), the
    // partial sums fit the dead Q / dO images, and K / V ride in registers (head_dim <= 80: at 160 every extra wave stages a
    // private K / V tile, and the 7x7 cross-attention backward went from 125 to 154 us with two query groups)
    int qw = 1;
    if (d <= 80 && nkt <= 2 && kw == nkt) {
        qw = 4 / kw;
        if (qw > nqt) qw = nqt;
        if (qw < 1) qw = 1;
        const size_t th = (size_t)((d / 16 + 1) / 2);
        while (qw > 1 && ((size_t)(qw - 1) * kw * th * 8192 > dkv_lds_w(L, d, kw * qw) || dkv_lds_w(L, d, kw * qw) > MFMA_LDS_CAP)) --qw;
    }
    int w = kw * qw;
    if (qw == 1) while (w > 1 && dkv_lds_w(L, d, w) > MFMA_LDS_CAP) --w;
    return w;
}
static inline size_t dkv_lds_m(int L, int S, int d) { return dkv_lds_w(L, d, dkv_waves(L, S, d)); }

// returns 1 when the MFMA path handles this problem
int attn_mfma_applicable(int L, int S, int d, int dtype, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo) {
    if (dtype != PSG_BF16 || !nd_supported(d)) return 0;
    if (((ldq | ldk | ldv | ldo) & 7) != 0) return 0;                    // 16-byte row fragments
    if (fwd_lds_m(S, d) > MFMA_LDS_CAP || dkv_lds_m(L, S, d) > MFMA_LDS_CAP) return 0;
    return 1;
}

#define ND_DISPATCH(KERNEL, ...)                                                                      \
    switch (d) {                                                                                      \
        case 16: hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__); break;                                   \
        case 32: hipLaunchKernelGGL(KERNEL<2>, __VA_ARGS__); break;                                   \
        case 64: hipLaunchKernelGGL(KERNEL<4>, __VA_ARGS__); break;                                   \
        case 80: hipLaunchKernelGGL(KERNEL<5>, __VA_ARGS__); break;                                   \
        case 160: hipLaunchKernelGGL(KERNEL<10>, __VA_ARGS__); break;                                 \
        default: hipLaunchKernelGGL(KERNEL<20>, __VA_ARGS__); break;                                  \
    }

int attn_mfma_init_attrs() {
#define SET_LDS(K) PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)MFMA_LDS_CAP))
    SET_LDS(attn_fwd_mfma<1>); SET_LDS(attn_fwd_mfma<2>); SET_LDS(attn_fwd_mfma<4>); SET_LDS(attn_fwd_mfma<5>); SET_LDS(attn_fwd_mfma<10>); SET_LDS(attn_fwd_mfma<20>);
    SET_LDS(attn_dq_mfma<1>); SET_LDS(attn_dq_mfma<2>); SET_LDS(attn_dq_mfma<4>); SET_LDS(attn_dq_mfma<5>); SET_LDS(attn_dq_mfma<10>); SET_LDS(attn_dq_mfma<20>);
    SET_LDS(attn_dkv_mfma<1>); SET_LDS(attn_dkv_mfma<2>); SET_LDS(attn_dkv_mfma<4>); SET_LDS(attn_dkv_mfma<5>); SET_LDS(attn_dkv_mfma<10>); SET_LDS(attn_dkv_mfma<20>);
#undef SET_LDS
    return PSG_OK;
}

int attn_mfma_fwd(const AttnMP& p, hipStream_t s) {
    const int d = p.d;
    ND_DISPATCH(attn_fwd_mfma, dim3(p.B * p.H), dim3(64 * attn_waves(p.L)), fwd_lds_m(p.S, d), s, p);
    PSG_LAUNCH_CHECK("attn_fwd_mfma");
    return PSG_OK;
}
int attn_mfma_bwd(const AttnMP& p, hipStream_t s) {
    const int d = p.d;
    ND_DISPATCH(attn_dq_mfma, dim3(p.B * p.H), dim3(64 * attn_waves(p.L)), fwd_lds_m(p.S, d), s, p);
    PSG_LAUNCH_CHECK("attn_dq_mfma");
    ND_DISPATCH(attn_dkv_mfma, dim3(p.B * p.H), dim3(64 * dkv_waves(p.L, p.S, d)), dkv_lds_m(p.L, p.S, d), s, p);
    PSG_LAUNCH_CHECK("attn_dkv_mfma");
    return PSG_OK;
}

}  // namespace psg
