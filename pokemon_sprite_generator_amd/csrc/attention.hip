// Multi-head attention core: softmax((q*scale) k^T) v and its backward.
// Shape-general fp32-accumulate kernels (any L, S, head_dim % 4 == 0 up to 320, fp32 or bf16 I/O):
// K/V chunks of 32 keys are staged through LDS (row stride d+1: conflict-free), the [16 x S]
// score slab of a 16-query tile lives in LDS, row softmax is a wave-shuffle reduction, and the
// probabilities never touch HBM.  lse (log-sum-exp) is saved for backward, which recomputes P.
// Problems are tiny and independent (B*heads of them per call): latency/LDS-bound, not MFMA-bound.
#include "psg_common.h"

namespace psg {

constexpr int AT_Q = 16;      // query rows per workgroup (fwd, dq)
constexpr int AT_KC = 32;     // keys per staged chunk
constexpr int AT_MAXC = 20;   // head_dim <= 16 * AT_MAXC = 320

struct AttnP {
    const void *q, *k, *v, *o, *dout;
    void *out, *dq, *dk, *dv;
    float* lse; float* delta;
    int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    int B, H, L, S, d;
    float scale;
    uint32_t drop_thresh; float drop_scale; uint64_t seed;
    const uint64_t* seed_dev;
};

// dropout element index of P[bh][l][s]: each query row owns ceil(S/2) hash PAIRS (keys 2k, 2k+1 share one 32-bit hash),
// exactly as attention_mfma.hip lays them out - so the VALU and the MFMA kernels draw the SAME mask for every S (odd S
// too: 7x7 self-attention has S = 49) and a forward on one path can be differentiated on the other
__device__ __forceinline__ uint64_t attn_idx(const AttnP& p, int bh, int l, int s) {
    return ((uint64_t)bh * p.L + l) * (uint64_t)(2 * ((p.S + 1) >> 1)) + s;
}

// stage `rows` rows of d elements (global row r at base + r*ld) into LDS with row stride d+1, scaled
template <typename T>
__device__ __forceinline__ void stage_rows(float* dst, const T* base, int64_t ld, int r0, int rows, int rmax, int d, float scl) {
    const int d4 = d >> 2;
    for (int e = threadIdx.x; e < rows * d4; e += blockDim.x) {
        const int r = e / d4, c = (e - r * d4) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + r < rmax) v = load4<T>(base + (int64_t)(r0 + r) * ld + c);
        float* o = dst + r * (d + 1) + c;
        o[0] = v[0] * scl; o[1] = v[1] * scl; o[2] = v[2] * scl; o[3] = v[3] * scl;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int d = p.d, dp = d + 1;
    const int Sp = (p.S + 3) & ~3;
    float* Qs = sm;                       // [AT_Q][dp]
    float* Ss = Qs + AT_Q * dp;           // [AT_Q][Sp]
    float* KV = Ss + AT_Q * Sp;           // [AT_KC][dp]
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int l0 = blockIdx.x * AT_Q;
    const T* qg = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.L * p.ldq + h * d;
    const T* kg = reinterpret_cast<const T*>(p.k) + (int64_t)b * p.S * p.ldk + h * d;
    const T* vg = reinterpret_cast<const T*>(p.v) + (int64_t)b * p.S * p.ldv + h * d;
    T* og = reinterpret_cast<T*>(p.out) + (int64_t)b * p.L * p.ldo + h * d;
    const int tid = threadIdx.x;

    stage_rows<T>(Qs, qg, p.ldq, l0, AT_Q, p.L, d, p.scale);
    // scores
    for (int s0 = 0; s0 < p.S; s0 += AT_KC) {
        __syncthreads();
        stage_rows<T>(KV, kg, p.ldk, s0, AT_KC, p.S, d, 1.0f);
        __syncthreads();
        const int kj = tid & 31, qa = tid >> 5, qb = qa + 8;
        const float* kr = KV + kj * dp;
        const float* q0 = Qs + qa * dp;
        const float* q1 = Qs + qb * dp;
        float a0 = 0.f, a1 = 0.f;
        for (int e = 0; e < d; ++e) { const float kv = kr[e]; a0 += q0[e] * kv; a1 += q1[e] * kv; }
        if (s0 + kj < p.S) { Ss[qa * Sp + s0 + kj] = a0; Ss[qb * Sp + s0 + kj] = a1; }
    }
    __syncthreads();
    // row softmax: wave w handles rows 4w..4w+3
    {
        const int lane = tid & 63, wv = tid >> 6;
        for (int r = wv * 4; r < wv * 4 + 4; ++r) {
            const int l = l0 + r;
            float* row = Ss + r * Sp;
            float mx = -INFINITY;
            for (int s = lane; s < p.S; s += 64) mx = fmaxf(mx, row[s]);
            mx = wave_max(mx);
            float sum = 0.f;
            for (int s = lane; s < p.S; s += 64) { const float e = __expf(row[s] - mx); row[s] = e; sum += e; }
            sum = wave_sum(sum);
            const float inv = 1.0f / sum;
            for (int s = lane; s < p.S; s += 64) {
                float pv = row[s] * inv;
                if (p.drop_thresh && l < p.L) pv = drop_keep(eff_seed(p.seed, p.seed_dev), attn_idx(p, bh, l, s), p.drop_thresh) ? pv * p.drop_scale : 0.f;
                row[s] = pv;
            }
            if (lane == 0 && l < p.L) p.lse[(int64_t)bh * p.L + l] = mx + __logf(sum);
        }
    }
    // O = P V
    const int qi = tid >> 4, dd0 = tid & 15;
    const int nc = (d + 15) >> 4;
    float acc[AT_MAXC];
#pragma unroll
    for (int c = 0; c < AT_MAXC; ++c) acc[c] = 0.f;
    for (int s0 = 0; s0 < p.S; s0 += AT_KC) {
        __syncthreads();
        stage_rows<T>(KV, vg, p.ldv, s0, AT_KC, p.S, d, 1.0f);
        __syncthreads();
        const int jn = min(AT_KC, p.S - s0);
        for (int j = 0; j < jn; ++j) {
            const float pv = Ss[qi * Sp + s0 + j];
            const float* vr = KV + j * dp + dd0;
#pragma unroll
            for (int c = 0; c < AT_MAXC; ++c)
                if (c < nc && dd0 + 16 * c < d) acc[c] += pv * vr[16 * c];
        }
    }
    if (l0 + qi < p.L) {
#pragma unroll
        for (int c = 0; c < AT_MAXC; ++c)
            if (c < nc && dd0 + 16 * c < d) Elem<T>::st(og + (int64_t)(l0 + qi) * p.ldo + dd0 + 16 * c, acc[c]);
    }
}

// delta[bh, l] = sum_d dO * O
template <typename T>
__global__ void attn_delta_kernel(const AttnP p) {
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // (b*L + l)*H + h order? use bh-major
    const int lane = threadIdx.x & 63;
    const int64_t total = (int64_t)p.B * p.H * p.L;
    if (row >= total) return;
    const int bh = (int)(row / p.L), l = (int)(row - (int64_t)bh * p.L);
    const int b = bh / p.H, h = bh - b * p.H;
    const T* o = reinterpret_cast<const T*>(p.o) + ((int64_t)b * p.L + l) * p.ldo + h * p.d;
    const T* g = reinterpret_cast<const T*>(p.dout) + ((int64_t)b * p.L + l) * p.lddo + h * p.d;
    float a = 0.f;
    for (int e = lane; e < p.d; e += 64) a += Elem<T>::ld(o + e) * Elem<T>::ld(g + e);
    a = wave_sum(a);
    if (lane == 0) p.delta[row] = a;
}

// dQ for a 16-query tile
template <typename T>
__global__ __launch_bounds__(256) void attn_dq_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int d = p.d, dp = d + 1;
    const int Sp = (p.S + 3) & ~3;
    float* Qs = sm;                       // [AT_Q][dp]  (scaled)
    float* Gs = Qs + AT_Q * dp;           // [AT_Q][dp]  dO
    float* Ss = Gs + AT_Q * dp;           // [AT_Q][Sp]  dS
    float* Ks = Ss + AT_Q * Sp;           // [AT_KC][dp]
    float* Vs = Ks + AT_KC * dp;          // [AT_KC][dp]
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int l0 = blockIdx.x * AT_Q;
    const T* qg = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.L * p.ldq + h * d;
    const T* kg = reinterpret_cast<const T*>(p.k) + (int64_t)b * p.S * p.ldk + h * d;
    const T* vg = reinterpret_cast<const T*>(p.v) + (int64_t)b * p.S * p.ldv + h * d;
    const T* gg = reinterpret_cast<const T*>(p.dout) + (int64_t)b * p.L * p.lddo + h * d;
    T* dqg = reinterpret_cast<T*>(p.dq) + (int64_t)b * p.L * p.lddq + h * d;
    const int tid = threadIdx.x;

    stage_rows<T>(Qs, qg, p.ldq, l0, AT_Q, p.L, d, p.scale);
    stage_rows<T>(Gs, gg, p.lddo, l0, AT_Q, p.L, d, 1.0f);
    const int kj = tid & 31, qa = tid >> 5, qb = qa + 8;
    const int la = l0 + qa, lb = l0 + qb;
    const float lse_a = la < p.L ? p.lse[(int64_t)bh * p.L + la] : 0.f;
    const float lse_b = lb < p.L ? p.lse[(int64_t)bh * p.L + lb] : 0.f;
    const float del_a = la < p.L ? p.delta[(int64_t)bh * p.L + la] : 0.f;
    const float del_b = lb < p.L ? p.delta[(int64_t)bh * p.L + lb] : 0.f;
    for (int s0 = 0; s0 < p.S; s0 += AT_KC) {
        __syncthreads();
        stage_rows<T>(Ks, kg, p.ldk, s0, AT_KC, p.S, d, 1.0f);
        stage_rows<T>(Vs, vg, p.ldv, s0, AT_KC, p.S, d, 1.0f);
        __syncthreads();
        const float* kr = Ks + kj * dp; const float* vr = Vs + kj * dp;
        const float* q0 = Qs + qa * dp; const float* q1 = Qs + qb * dp;
        const float* g0 = Gs + qa * dp; const float* g1 = Gs + qb * dp;
        float s_a = 0.f, s_b = 0.f, dp_a = 0.f, dp_b = 0.f;
        for (int e = 0; e < d; ++e) {
            const float kv = kr[e], vv = vr[e];
            s_a += q0[e] * kv; s_b += q1[e] * kv;
            dp_a += g0[e] * vv; dp_b += g1[e] * vv;
        }
        const int s = s0 + kj;
        if (s < p.S) {
            float pa = __expf(s_a - lse_a), pb = __expf(s_b - lse_b);
            if (p.drop_thresh) {
                dp_a = (la < p.L && drop_keep(eff_seed(p.seed, p.seed_dev), attn_idx(p, bh, la, s), p.drop_thresh)) ? dp_a * p.drop_scale : 0.f;
                dp_b = (lb < p.L && drop_keep(eff_seed(p.seed, p.seed_dev), attn_idx(p, bh, lb, s), p.drop_thresh)) ? dp_b * p.drop_scale : 0.f;
            }
            Ss[qa * Sp + s] = la < p.L ? pa * (dp_a - del_a) : 0.f;
            Ss[qb * Sp + s] = lb < p.L ? pb * (dp_b - del_b) : 0.f;
        }
    }
    // dQ = scale * dS K
    const int qi = tid >> 4, dd0 = tid & 15;
    const int nc = (d + 15) >> 4;
    float acc[AT_MAXC];
#pragma unroll
    for (int c = 0; c < AT_MAXC; ++c) acc[c] = 0.f;
    for (int s0 = 0; s0 < p.S; s0 += AT_KC) {
        __syncthreads();
        stage_rows<T>(Ks, kg, p.ldk, s0, AT_KC, p.S, d, 1.0f);
        __syncthreads();
        const int jn = min(AT_KC, p.S - s0);
        for (int j = 0; j < jn; ++j) {
            const float ds = Ss[qi * Sp + s0 + j];
            const float* kr = Ks + j * dp + dd0;
#pragma unroll
            for (int c = 0; c < AT_MAXC; ++c)
                if (c < nc && dd0 + 16 * c < d) acc[c] += ds * kr[16 * c];
        }
    }
    if (l0 + qi < p.L) {
#pragma unroll
        for (int c = 0; c < AT_MAXC; ++c)
            if (c < nc && dd0 + 16 * c < d) Elem<T>::st(dqg + (int64_t)(l0 + qi) * p.lddq + dd0 + 16 * c, acc[c] * p.scale);
    }
}

// dK, dV for a 16-key tile: loops over all queries in chunks of 32
template <typename T>
__global__ __launch_bounds__(256) void attn_dkv_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int d = p.d, dp = d + 1;
    float* Ks = sm;                        // [16][dp]
    float* Vs = Ks + AT_Q * dp;            // [16][dp]
    float* Qs = Vs + AT_Q * dp;            // [32][dp] scaled
    float* Gs = Qs + AT_KC * dp;           // [32][dp]
    float* Pt = Gs + AT_KC * dp;           // [16][33] P_dropped^T
    float* Dt = Pt + AT_Q * 33;            // [16][33] dS^T
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int s0 = blockIdx.x * AT_Q;
    const T* qg = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.L * p.ldq + h * d;
    const T* kg = reinterpret_cast<const T*>(p.k) + (int64_t)b * p.S * p.ldk + h * d;
    const T* vg = reinterpret_cast<const T*>(p.v) + (int64_t)b * p.S * p.ldv + h * d;
    const T* gg = reinterpret_cast<const T*>(p.dout) + (int64_t)b * p.L * p.lddo + h * d;
    T* dkg = reinterpret_cast<T*>(p.dk) + (int64_t)b * p.S * p.lddk + h * d;
    T* dvg = reinterpret_cast<T*>(p.dv) + (int64_t)b * p.S * p.lddv + h * d;
    const int tid = threadIdx.x;
    stage_rows<T>(Ks, kg, p.ldk, s0, AT_Q, p.S, d, 1.0f);
    stage_rows<T>(Vs, vg, p.ldv, s0, AT_Q, p.S, d, 1.0f);

    const int kj = tid >> 4, dd0 = tid & 15;       // accumulate phase mapping
    const int nc = (d + 15) >> 4;
    float accK[AT_MAXC], accV[AT_MAXC];
#pragma unroll
    for (int c = 0; c < AT_MAXC; ++c) { accK[c] = 0.f; accV[c] = 0.f; }

    for (int l0 = 0; l0 < p.L; l0 += AT_KC) {
        __syncthreads();
        stage_rows<T>(Qs, qg, p.ldq, l0, AT_KC, p.L, d, p.scale);
        stage_rows<T>(Gs, gg, p.lddo, l0, AT_KC, p.L, d, 1.0f);
        __syncthreads();
        {   // scores for (query qi = tid&31, keys ka = tid>>5, kb = ka+8)
            const int qi = tid & 31, ka = tid >> 5, kb = ka + 8;
            const float* qr = Qs + qi * dp; const float* gr = Gs + qi * dp;
            const float* k0 = Ks + ka * dp; const float* k1 = Ks + kb * dp;
            const float* v0 = Vs + ka * dp; const float* v1 = Vs + kb * dp;
            float s_a = 0.f, s_b = 0.f, dp_a = 0.f, dp_b = 0.f;
            for (int e = 0; e < d; ++e) {
                const float qv = qr[e], gv = gr[e];
                s_a += qv * k0[e]; s_b += qv * k1[e];
                dp_a += gv * v0[e]; dp_b += gv * v1[e];
            }
            const int l = l0 + qi;
            float pa = 0.f, pb = 0.f, da = 0.f, db = 0.f;
            if (l < p.L) {
                const float lse = p.lse[(int64_t)bh * p.L + l], del = p.delta[(int64_t)bh * p.L + l];
                const int sa = s0 + ka, sb = s0 + kb;
                if (sa < p.S) {
                    float pr = __expf(s_a - lse), pd = pr;
                    if (p.drop_thresh) {
                        const bool keep = drop_keep(eff_seed(p.seed, p.seed_dev), attn_idx(p, bh, l, sa), p.drop_thresh);
                        pd = keep ? pr * p.drop_scale : 0.f; dp_a = keep ? dp_a * p.drop_scale : 0.f;
                    }
                    pa = pd; da = pr * (dp_a - del);
                }
                if (sb < p.S) {
                    float pr = __expf(s_b - lse), pd = pr;
                    if (p.drop_thresh) {
                        const bool keep = drop_keep(eff_seed(p.seed, p.seed_dev), attn_idx(p, bh, l, sb), p.drop_thresh);
                        pd = keep ? pr * p.drop_scale : 0.f; dp_b = keep ? dp_b * p.drop_scale : 0.f;
                    }
                    pb = pd; db = pr * (dp_b - del);
                }
            }
            Pt[ka * 33 + qi] = pa; Pt[kb * 33 + qi] = pb;
            Dt[ka * 33 + qi] = da; Dt[kb * 33 + qi] = db;
        }
        __syncthreads();
        const int jn = min(AT_KC, p.L - l0);
        for (int j = 0; j < jn; ++j) {
            const float pv = Pt[kj * 33 + j], ds = Dt[kj * 33 + j];
            const float* gr = Gs + j * dp + dd0;
            const float* qr = Qs + j * dp + dd0;
#pragma unroll
            for (int c = 0; c < AT_MAXC; ++c)
                if (c < nc && dd0 + 16 * c < d) { accV[c] += pv * gr[16 * c]; accK[c] += ds * qr[16 * c]; }
        }
    }
    if (s0 + kj < p.S) {
#pragma unroll
        for (int c = 0; c < AT_MAXC; ++c)
            if (c < nc && dd0 + 16 * c < d) {
                Elem<T>::st(dkg + (int64_t)(s0 + kj) * p.lddk + dd0 + 16 * c, accK[c]);   // Qs already carries `scale`
                Elem<T>::st(dvg + (int64_t)(s0 + kj) * p.lddv + dd0 + 16 * c, accV[c]);
            }
    }
}

static size_t fwd_lds(int S, int d) { return sizeof(float) * ((size_t)AT_Q * (d + 1) + (size_t)AT_Q * ((S + 3) & ~3) + (size_t)AT_KC * (d + 1)); }
static size_t dq_lds(int S, int d) { return sizeof(float) * ((size_t)2 * AT_Q * (d + 1) + (size_t)AT_Q * ((S + 3) & ~3) + (size_t)2 * AT_KC * (d + 1)); }
static size_t dkv_lds(int d) { return sizeof(float) * ((size_t)2 * AT_Q * (d + 1) + (size_t)2 * AT_KC * (d + 1) + 2 * AT_Q * 33); }

static int attn_check(const char* who, int B, int heads, int L, int S, int d, int dtype, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo) {
    PSG_REQUIRE(dtype == PSG_F32 || dtype == PSG_BF16, PSG_ERR_DTYPE, "%s: dtype %d", who, dtype);
    PSG_REQUIRE(B > 0 && heads > 0 && L > 0 && S > 0 && d > 0, PSG_ERR_SHAPE, "%s: non-positive dimension", who);
    PSG_REQUIRE(d % 4 == 0 && d <= 16 * AT_MAXC, PSG_ERR_SHAPE, "%s: head_dim %d must be a multiple of 4 and <= %d", who, d, 16 * AT_MAXC);
    PSG_REQUIRE(ldq >= heads * d && ldk >= heads * d && ldv >= heads * d && ldo >= heads * d, PSG_ERR_SHAPE, "%s: row stride < heads*d", who);
    PSG_REQUIRE(((ldq | ldk | ldv | ldo) & 3) == 0, PSG_ERR_ALIGN, "%s: row strides must be multiples of 4", who);
    PSG_REQUIRE(S <= 4096, PSG_ERR_SHAPE, "%s: S=%d too long for the LDS score slab", who, S);
    PSG_REQUIRE((int64_t)B * heads <= 65535, PSG_ERR_SHAPE, "%s: B*heads=%ld exceeds grid.y", who, (long)B * heads);
    return PSG_OK;
}

// attention_mfma.hip (bf16 matrix-core path)
struct AttnMP {
    const bf16_t *q, *k, *v, *o, *dout;
    bf16_t *out, *dq, *dk, *dv;
    float* lse; float* delta;
    int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    int B, H, L, S, d;
    float scale;
    uint32_t drop_thresh; float drop_scale; uint64_t seed;
    const uint64_t* seed_dev;
};
int attn_mfma_applicable(int L, int S, int d, int dtype, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo);
int attn_mfma_init_attrs();
int attn_mfma_fwd(const AttnMP& p, hipStream_t s);
int attn_mfma_bwd(const AttnMP& p, hipStream_t s);

// attention_f32.hip (exact-fp32 matrix-core path)
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
int attn_f32_applicable(int L, int S, int d, int dtype, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo);
int attn_f32_init_attrs();
int attn_f32_fwd(const AttnFP& p, hipStream_t s);
int attn_f32_bwd(const AttnFP& p, hipStream_t s);

static AttnFP to_f32(const AttnP& a) {
    AttnFP m;
    m.q = (const float*)a.q; m.k = (const float*)a.k; m.v = (const float*)a.v; m.o = (const float*)a.o; m.dout = (const float*)a.dout;
    m.out = (float*)a.out; m.dq = (float*)a.dq; m.dk = (float*)a.dk; m.dv = (float*)a.dv;
    m.lse = a.lse; m.delta = a.delta;
    m.ldq = a.ldq; m.ldk = a.ldk; m.ldv = a.ldv; m.ldo = a.ldo; m.lddo = a.lddo; m.lddq = a.lddq; m.lddk = a.lddk; m.lddv = a.lddv;
    m.B = a.B; m.H = a.H; m.L = a.L; m.S = a.S; m.d = a.d; m.scale = a.scale;
    m.drop_thresh = a.drop_thresh; m.drop_scale = a.drop_scale; m.seed = a.seed; m.seed_dev = a.seed_dev;
    return m;
}

static AttnMP to_mfma(const AttnP& a) {
    AttnMP m;
    m.q = (const bf16_t*)a.q; m.k = (const bf16_t*)a.k; m.v = (const bf16_t*)a.v; m.o = (const bf16_t*)a.o; m.dout = (const bf16_t*)a.dout;
    m.out = (bf16_t*)a.out; m.dq = (bf16_t*)a.dq; m.dk = (bf16_t*)a.dk; m.dv = (bf16_t*)a.dv;
    m.lse = a.lse; m.delta = a.delta;
    m.ldq = a.ldq; m.ldk = a.ldk; m.ldv = a.ldv; m.ldo = a.ldo; m.lddo = a.lddo; m.lddq = a.lddq; m.lddk = a.lddk; m.lddv = a.lddv;
    m.B = a.B; m.H = a.H; m.L = a.L; m.S = a.S; m.d = a.d; m.scale = a.scale;
    m.drop_thresh = a.drop_thresh; m.drop_scale = a.drop_scale; m.seed = a.seed; m.seed_dev = a.seed_dev;
    return m;
}

}  // namespace psg
using namespace psg;

extern "C" {

int psg_attn_init_attrs(void) {
    { int rc = attn_mfma_init_attrs(); if (rc) return rc; }
    { int rc = attn_f32_init_attrs(); if (rc) return rc; }
    const int big = 150 * 1024;
#define SET_LDS(K) PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, big))
    SET_LDS(attn_fwd_kernel<float>); SET_LDS(attn_fwd_kernel<bf16_t>);
    SET_LDS(attn_dq_kernel<float>); SET_LDS(attn_dq_kernel<bf16_t>);
    SET_LDS(attn_dkv_kernel<float>); SET_LDS(attn_dkv_kernel<bf16_t>);
#undef SET_LDS
    return PSG_OK;
}

static int64_t g_attn_paths[3] = {0, 0, 0};         // launches taken by the bf16 MFMA / the VALU / the fp32 MFMA kernels
static int f32_mfma_on() {
    static int on = -1;                              // PSG_ATTN_F32_MFMA=0: the strict path runs the VALU kernels (A/B)
    if (on < 0) { const char* e = getenv("PSG_ATTN_F32_MFMA"); on = (e && atoi(e) == 0) ? 0 : 1; }
    return on;
}
static int g_attn_allow = 3;                         // psg_attn_set_paths: bit 0 bf16 MFMA, bit 1 exact-fp32 MFMA
int psg_attn_set_paths(int allow_mask) { g_attn_allow = allow_mask & 3; return PSG_OK; }
int psg_attn_path_counts(int64_t* mfma, int64_t* valu, int64_t* mfma_f32) {
    if (mfma) *mfma = g_attn_paths[0];
    if (valu) *valu = g_attn_paths[1];
    if (mfma_f32) *mfma_f32 = g_attn_paths[2];
    return PSG_OK;
}

int psg_attn_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo,
                 float* lse, int B, int heads, int L, int S, int d, float scale, float drop_p, uint64_t seed, int dtype,
                 psg_stream_t stream) {
    PSG_REQUIRE(q && k && v && o && lse, PSG_ERR_ARG, "attn_fwd: null pointer");
    int rc = attn_check("attn_fwd", B, heads, L, S, d, dtype, ldq, ldk, ldv, ldo);
    if (rc) return rc;
    PSG_REQUIRE(drop_p >= 0.f && drop_p < 1.f, PSG_ERR_ARG, "attn_fwd: drop_p");
    const size_t lds = fwd_lds(S, d);
    PSG_REQUIRE(lds <= 150 * 1024, PSG_ERR_SHAPE, "attn_fwd: LDS need %zu too large", lds);
    AttnP p = {};
    p.q = q; p.k = k; p.v = v; p.out = o; p.lse = lse; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.B = B; p.H = heads; p.L = L; p.S = S; p.d = d; p.scale = scale;
    p.drop_thresh = drop_p > 0.f ? drop_thresh(drop_p) : 0u; p.drop_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f; p.seed = seed; p.seed_dev = seed_source();
    dim3 grid((L + AT_Q - 1) / AT_Q, B * heads);
    ProfScope prof(PROF_ATTN, 4.0 * (double)B * heads * L * S * d, (hipStream_t)stream, (double)B * heads * d * (2.0 * L + 2.0 * S) * (dtype == PSG_BF16 ? 2.0 : 4.0));
    if ((g_attn_allow & 1) && attn_mfma_applicable(L, S, d, dtype, ldq, ldk, ldv, ldo) && aligned16(q) && aligned16(k) && aligned16(v) && aligned8(o)) {
        ++g_attn_paths[0];
        return attn_mfma_fwd(to_mfma(p), (hipStream_t)stream);
    }
    if ((g_attn_allow & 2) && f32_mfma_on() && attn_f32_applicable(L, S, d, dtype, ldq, ldk, ldv, ldo) && aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o)) {
        ++g_attn_paths[2];
        return attn_f32_fwd(to_f32(p), (hipStream_t)stream);
    }
    ++g_attn_paths[1];
    if (dtype == PSG_F32) hipLaunchKernelGGL(attn_fwd_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(attn_fwd_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, p);
    PSG_LAUNCH_CHECK("attn_fwd");
    return PSG_OK;
}

int psg_attn_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o,
                 int64_t ldo, const void* dout, int64_t lddo, const float* lse, float* delta, void* dq, int64_t lddq,
                 void* dk, int64_t lddk, void* dv, int64_t lddv, int B, int heads, int L, int S, int d, float scale,
                 float drop_p, uint64_t seed, int dtype, psg_stream_t stream) {
    PSG_REQUIRE(q && k && v && o && dout && lse && delta && dq && dk && dv, PSG_ERR_ARG, "attn_bwd: null pointer");
    int rc = attn_check("attn_bwd", B, heads, L, S, d, dtype, ldq, ldk, ldv, ldo);
    if (rc) return rc;
    PSG_REQUIRE(lddo >= heads * d && lddq >= heads * d && lddk >= heads * d && lddv >= heads * d && ((lddo | lddq | lddk | lddv) & 3) == 0,
                PSG_ERR_SHAPE, "attn_bwd: gradient row strides");
    PSG_REQUIRE(drop_p >= 0.f && drop_p < 1.f, PSG_ERR_ARG, "attn_bwd: drop_p");
    const size_t l1 = dq_lds(S, d), l2 = dkv_lds(d);
    PSG_REQUIRE(l1 <= 150 * 1024 && l2 <= 150 * 1024, PSG_ERR_SHAPE, "attn_bwd: LDS need too large");
    AttnP p = {};
    p.q = q; p.k = k; p.v = v; p.o = o; p.dout = dout; p.lse = const_cast<float*>(lse); p.delta = delta;
    p.dq = dq; p.dk = dk; p.dv = dv;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
    p.B = B; p.H = heads; p.L = L; p.S = S; p.d = d; p.scale = scale;
    p.drop_thresh = drop_p > 0.f ? drop_thresh(drop_p) : 0u; p.drop_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f; p.seed = seed; p.seed_dev = seed_source();
    hipStream_t s = (hipStream_t)stream;
    const int64_t rows = (int64_t)B * heads * L;
    const int gdelta = (int)((rows + 3) / 4);
    dim3 gq((L + AT_Q - 1) / AT_Q, B * heads), gkv((S + AT_Q - 1) / AT_Q, B * heads);
    ProfScope prof(PROF_ATTN, 10.0 * (double)B * heads * L * S * d, s, (double)B * heads * d * (4.0 * L + 4.0 * S) * (dtype == PSG_BF16 ? 2.0 : 4.0));
    if ((g_attn_allow & 1) && attn_mfma_applicable(L, S, d, dtype, ldq, ldk, ldv, ldo) && ((lddo | lddq | lddk | lddv) & 7) == 0 && aligned16(q) && aligned16(k) &&
        aligned16(v) && aligned16(o) && aligned16(dout) && aligned8(dq) && aligned8(dk) && aligned8(dv)) {
        ++g_attn_paths[0];
        return attn_mfma_bwd(to_mfma(p), s);       // (delta is produced inside the dQ kernel)
    }
    if ((g_attn_allow & 2) && f32_mfma_on() && attn_f32_applicable(L, S, d, dtype, ldq, ldk, ldv, ldo) && ((lddo | lddq | lddk | lddv) & 3) == 0 && aligned16(q) &&
        aligned16(k) && aligned16(v) && aligned16(o) && aligned16(dout) && aligned16(dq) && aligned16(dk) && aligned16(dv)) {
        ++g_attn_paths[2];
        return attn_f32_bwd(to_f32(p), s);          // (delta is produced inside the dQ kernel)
    }
    ++g_attn_paths[1];
    if (dtype == PSG_F32) {
        hipLaunchKernelGGL(attn_delta_kernel<float>, dim3(gdelta), dim3(256), 0, s, p);
        hipLaunchKernelGGL(attn_dq_kernel<float>, gq, dim3(256), l1, s, p);
        hipLaunchKernelGGL(attn_dkv_kernel<float>, gkv, dim3(256), l2, s, p);
    } else {
        hipLaunchKernelGGL(attn_delta_kernel<bf16_t>, dim3(gdelta), dim3(256), 0, s, p);
        hipLaunchKernelGGL(attn_dq_kernel<bf16_t>, gq, dim3(256), l1, s, p);
        hipLaunchKernelGGL(attn_dkv_kernel<bf16_t>, gkv, dim3(256), l2, s, p);
    }
    PSG_LAUNCH_CHECK("attn_bwd");
    return PSG_OK;
}

}  // extern "C"
