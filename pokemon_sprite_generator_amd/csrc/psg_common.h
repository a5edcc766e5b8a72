// Shared device/host helpers for libpsg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <math.h>
#include <stdlib.h>

#include "../../include/psg_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace psg {

int set_error(int code, const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define PSG_HIP_CHECK(expr)                                   \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return psg::hip_fail(_e, #expr); \
    } while (0)
#define PSG_LAUNCH_CHECK(name)                                 \
    do {                                                       \
        hipError_t _e = hipGetLastError();                     \
        if (_e != hipSuccess) return psg::hip_fail(_e, name);  \
    } while (0)
#define PSG_REQUIRE(cond, code, ...)                              \
    do {                                                          \
        if (!(cond)) return psg::set_error(code, __VA_ARGS__);    \
    } while (0)

// bench-only timing of a kernel family (profile.hip); a no-op unless psg_profile_begin() was called
enum ProfKind { PROF_CONV_FWD = 0, PROF_CONV_DGRAD = 1, PROF_WGRAD = 2, PROF_ATTN = 3, PROF_GN = 4, PROF_KINDS = 5 };
struct ProfScope {
    ProfScope(int kind, double work, hipStream_t stream, double bytes = 0.0);   // bytes: the launch's ALGORITHMIC operand bytes
    ~ProfScope();
    int idx_;
    hipStream_t stream_;
};

// CUs the tile choosers may count on (psg_set_available_cus; 256 unless part of the chip is taken, e.g. by an overlapped
// RCCL all-reduce): a launch planned as ONE round of workgroups over 256 CUs becomes two when 16 of them are busy
int avail_cus();
// ... for a launch that would take `rounds_full` rounds of workgroups on the whole chip: with psg_set_reserve_rounds(r > 0) only
// launches of at most r rounds plan around the reserve (a one-round grid becomes two when CUs are taken: 2x; a ten-round
// grid loses 14 % either way, and planning it for fewer CUs costs that 14 % also while the CUs are free)
int avail_cus_for(double rounds_full);

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

// ---- element access -------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int CH = 4;  // elements per 16-byte chunk
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int CH = 8;
    static __device__ __forceinline__ float ld(const bf16_t* p) { return (float)*p; }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = (bf16_t)v; }
};

// 4 consecutive elements
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 load4<bf16_t>(const bf16_t* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
}
template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, f32x4 v) {
    bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *reinterpret_cast<bf16x4*>(p) = r;
}

// ---- activations ------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float silu_grad(float x) {
    float s = sigmoidf_(x);
    return s * (1.0f + x * (1.0f - s));
}
// bf16 compute path (FAST): v_rcp_f32 (1 ulp) instead of the IEEE division (v_div_scale x2 / v_div_fmas / v_div_fixup + 4
// fma: the division was 30-40 % of the vector instructions of the fused GroupNorm kernels) and fused multiply-adds; the
// exact-fp32 path keeps the division and the unfused forms (its results do not change).
template <bool FAST> __device__ __forceinline__ float sigmoid_t(float x) {
    if (FAST) return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
    return sigmoidf_(x);
}
template <bool FAST> __device__ __forceinline__ float silu_t(float x) { return x * sigmoid_t<FAST>(x); }
template <bool FAST> __device__ __forceinline__ float silu_grad_t(float x) {
    const float s = sigmoid_t<FAST>(x);
    if (FAST) return s * __builtin_fmaf(x, 1.0f - s, 1.0f);
    return s * (1.0f + x * (1.0f - s));
}
template <bool FAST> __device__ __forceinline__ float mad_t(float a, float b, float c) {      // a * b + c
    if (FAST) return __builtin_fmaf(a, b, c);
    return a * b + c;
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float x) {
    float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
// GELU and its derivative from ONE exponential: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute - far below a
// bf16 ulp; the exact-fp32 path keeps erff), whose exp(-z^2) with z = x/sqrt(2) is also the Gaussian of the derivative.
// ~16 VALU operations instead of ~60 for erff + expf: the GELU epilogues of the FFN GEMMs were as long as their K loops.
__device__ __forceinline__ void gelu_both_fast(float x, float& y, float& dy) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);   // v_rcp_f32, 1 ulp (__frcp_rn is the 11-instruction IEEE division)
    const float e = __expf(-z * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float h = 0.5f - 0.5f * poly * e;                      // 0.5 * erf(|z|)
    const float cdf = x >= 0.f ? 0.5f + h : 0.5f - h;
    y = x * cdf;
    dy = cdf + x * 0.39894228040143268f * e;
}
// (value, derivative) of an activation; FAST selects the bf16-grade GELU above
template <bool FAST>
__device__ __forceinline__ void act_both(float x, int act, float& y, float& dy) {
    if (act == PSG_ACT_GELU) {
        if (FAST) { gelu_both_fast(x, y, dy); return; }
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
        y = x * cdf;
        dy = cdf + x * 0.39894228040143268f * __expf(-0.5f * x * x);
    } else if (act == PSG_ACT_SILU) {
        const float s = sigmoidf_(x);
        y = x * s;
        dy = s * (1.0f + x * (1.0f - s));
    } else if (act == PSG_ACT_RELU) {
        y = fmaxf(x, 0.f); dy = x > 0.f ? 1.0f : 0.f;
    } else if (act == PSG_ACT_TANH) {
        y = tanhf(x); dy = 1.0f - y * y;
    } else { y = x; dy = 1.0f; }
}
__device__ __forceinline__ float act_f(float x, int act) {
    return act == PSG_ACT_SILU ? silu_f(x) : (act == PSG_ACT_GELU ? gelu_f(x) : (act == PSG_ACT_RELU ? fmaxf(x, 0.f) : (act == PSG_ACT_TANH ? tanhf(x) : x)));
}
__device__ __forceinline__ float act_grad(float x, int act) {
    if (act == PSG_ACT_RELU) return x > 0.f ? 1.0f : 0.f;
    if (act == PSG_ACT_TANH) { const float t = tanhf(x); return 1.0f - t * t; }
    return act == PSG_ACT_SILU ? silu_grad(x) : (act == PSG_ACT_GELU ? gelu_grad(x) : 1.0f);
}

// ---- stateless dropout mask ---------------------------------------------------
// keep(idx) = 16-bit half of hash(seed, idx >> 1) >= p * 2^16 ; the same function regenerates the mask in backward.
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
// One 32-bit hash serves the two elements 2k, 2k+1 (16 bits each, resolution 1/65536): half the hash rounds of a
// per-element hash, and adjacent elements of a lane (vector epilogues, key pairs of an attention row) share it.
__device__ __forceinline__ uint32_t drop_hash_pair(uint64_t seed, uint64_t pair) {
    const uint32_t lo = (uint32_t)pair, hi = (uint32_t)(pair >> 32);
    return mix32(lo * 0x9E3779B1u + (uint32_t)seed + ((hi * 0x85EBCA77u) ^ (uint32_t)(seed >> 32)));
}
// Pair hashes along ONE row of a mask: drop_hash_pair(seed, base + off), off < 2^32, with the 64-bit half of the index
// done once per row (the attention kernels hash 8-16 pairs per lane and key tile: the 64-bit index arithmetic per pair
// was half of their vector instructions).  Same bits as drop_hash_pair, carry into the high word included.
struct DropRow { uint32_t lo, k0, k1; };
__device__ __forceinline__ DropRow drop_row(uint64_t seed, uint64_t base) {
    const uint32_t hi = (uint32_t)(base >> 32), s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
    DropRow r;
    r.lo = (uint32_t)base;
    r.k0 = s0 + ((hi * 0x85EBCA77u) ^ s1);
    r.k1 = s0 + (((hi + 1u) * 0x85EBCA77u) ^ s1);
    return r;
}
__device__ __forceinline__ uint32_t drop_hash_row(const DropRow& r, uint32_t off) {
    const uint32_t lo = r.lo + off;
    return mix32(lo * 0x9E3779B1u + (lo < r.lo ? r.k1 : r.k0));
}
__device__ __forceinline__ bool drop_keep_half(uint32_t h, int odd, uint32_t thresh) {
    return ((odd ? (h >> 16) : (h & 0xFFFFu)) >= (thresh >> 16));
}
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t idx, uint32_t thresh) {
    return drop_keep_half(drop_hash_pair(seed, idx >> 1), (int)(idx & 1), thresh);
}
// (attention_mfma.hip pairs the keys 2k, 2k+1 of a query row: pair = row * ceil(S/2) + (key >> 1))
// Optional device word added to every dropout seed of a launch (psg_set_seed_source): the seeds themselves are launch
// arguments, frozen when a train step is captured into a hipGraph - the word, advanced on the device between replays,
// keeps the masks changing from step to step.  NULL (the default): seeds are used as passed.
const uint64_t* seed_source();
static __device__ __forceinline__ uint64_t eff_seed(uint64_t seed, const uint64_t* dev) { return seed + (dev ? *dev : 0ull); }

static inline uint32_t drop_thresh(float p) {
    double t = (double)p * 4294967296.0;
    if (t < 0) t = 0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (uint32_t)t;
}

// ---- wave / block reductions ---------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block sum for blockDim.x multiple of 64, <= 1024; red: >= 16 floats of LDS; result broadcast.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}

// ---- LDS-DMA (buffer_load_dwordx4 ... lds) issued from inline asm ---------------------------------------
// hipcc tracks the builtin form as an LDS store and drains it with vmcnt(0) before the next ds_read, which
// defeats multi-stage rings; the asm form is invisible to its waitcnt pass, so EVERY wait for these loads is
// placed by hand (s_waitcnt vmcnt(N) + barrier before the tile is read).  Each lane moves 16 bytes from
// `rsrc` base + voff (out-of-range => zeros) to LDS address lds_base + lane*16; lds_base must be wave-uniform.
// M0 is written in the same statement that uses it (the compiler does not preserve it across statements).
__device__ __forceinline__ u32x4 make_rsrc(const void* base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    u32x4 r = {(uint32_t)a, (uint32_t)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
    return r;
}
__device__ __forceinline__ void lds_dma16(const u32x4& rsrc, uint32_t lds_base, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_base), "v"(voff), "s"(rsrc) : "memory");
}
// same, with a wave-uniform byte offset in the instruction's scalar-offset operand (added to voff by the address
// unit; the range check covers the sum), so lanes that step through memory by a uniform stride share ONE voff register
__device__ __forceinline__ void lds_dma16s(const u32x4& rsrc, uint32_t lds_base, uint32_t voff, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// exact n / d for n < 2^24 using a float reciprocal and one correction step
__device__ __forceinline__ int fastdiv(int n, int d, float inv_d) {
    int q = (int)((float)n * inv_d);
    int r = n - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) ++q;
    return q;
}

}  // namespace psg
