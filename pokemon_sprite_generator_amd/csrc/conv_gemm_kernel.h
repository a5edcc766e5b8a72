// Implicit-GEMM convolution / linear on the gfx950 matrix cores.
//
//   acc[n, m] = sum_k W[n, k] * X[pix(m, k), ci(k)]        (MFMA A = weights, B = activations)
//
// The product is computed "swapped" (weights as the MFMA A operand) so that each
// lane ends up holding 4 CONSECUTIVE OUTPUT CHANNELS of one pixel per accumulator
// register group: the channels-last store is then an 8-byte (bf16) / 16-byte
// (fp32) vector per lane and the fused epilogue (bias, per-sample add, activation,
// dropout, gate, residual) reads its operands with the same vectors.
//
// Tiling: BM pixels x BN channels per 256-thread workgroup (4 waves, 2x2), K step
// of 8 sixteen-byte chunks (64 bf16 / 32 fp32 values, always inside one filter
// tap when Cin % chunk-run == 0).  Operand tiles go global -> LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds issued from inline asm, hand-placed s_waitcnt vmcnt) into a
// double-buffered, XOR-swizzled LDS image (128-byte rows; chunk ^= (row>>1)&7
// makes the ds_read_b128 fragment reads conflict-free), one barrier per K step.
// bf16: v_mfma_f32_16x16x32_bf16; fp32: v_mfma_f32_32x32x2_f32 (exact fp32).
// Workgroup ids are remapped XCD-aware (each XCD's L2 sees a contiguous range of
// tiles) and rasterised in groups of 8 M-tiles sharing weight panels.
#pragma once
#include "psg_common.h"
#include <type_traits>

namespace psg {

struct ConvP {
    const void* x; const void* w; void* y;
    const float* bias; const void* rowadd; const void* residual; void* preact; const void* dact_u;
    int64_t ldx, ldy, ldra, ldres, ldpre, lddact, ldw;
    int B, Hi, Wi, Cin, Ho, Wo, N;
    int ks, stride, pad, transposed;
    int M, taps, cpt, KT, Kpad, fast, tpt;
    int mtiles, ntiles;
    int act; float alpha;
    int flags;                     // PSG_CONV_SAVE_DACT / PSG_CONV_DACT_MUL
    int epi_lds;                   // bf16: stage the output tile through LDS and store whole 16-byte row chunks
    int epi_generic;               // PSG_EPI_KINDS=0: run-time epilogue form for every launch (A/B of the per-kind copies)
    uint32_t drop_thresh; float drop_scale; uint64_t drop_seed;
    const uint64_t* seed_dev;      // optional device word added to drop_seed (psg_set_seed_source): per-step masks under hipGraph replay
    uint32_t x_bytes, w_bytes;     // extents for the bounds-checked buffer loads
    // MODE 3 (one parity class of a stride-2 data gradient): result pixels (sub_h0 + 2i, sub_w0 + 2j), i < sub_nH,
    // j < sub_nW; only the ntap filter taps that reach them: source pixel (i + tap_dh, j + tap_dw), weight tap tap_wi
    int sub_h0, sub_w0, sub_nH, sub_nW, ntap;
    int tap_dh[4], tap_dw[4], tap_wi[4];
    // split-K (small M: sampling, small training batches): `splits` workgroups share one output tile, each reduces
    // kt_per_split K steps and writes its fp32 partial tile to ws[split][M][N]; conv_splitk_finish_kernel sums the partials
    // in fixed order and applies the epilogue.  splits == 1: the kernel's own epilogue, no workspace.
    int splits, kt_per_split;
    float* ws;
    int64_t ws_bytes;
};

template <typename T> struct Mma;           // 32x32 fragment path (fp32 only; bf16 uses mma16 below)
template <> struct Mma<float> {
    static __device__ __forceinline__ f32x16 run(const uint4& a, const uint4& b, f32x16 c) {
        const float* fa = reinterpret_cast<const float*>(&a);
        const float* fb = reinterpret_cast<const float*>(&b);
#pragma unroll
        for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], c, 0, 0, 0);
        return c;
    }
};

// bf16 main loop: v_mfma_f32_16x16x32_bf16.  Same LDS bytes per FLOP as 32x32x16 (a fragment is 1 KiB either way)
// but the chip sustains a higher clock on this shape under load, and the 16-pixel x 16-channel accumulator tile
// (lane: pixel = lane&15, 4 consecutive channels 4*(lane>>4)..+3) stores 32-byte runs per pixel.
__device__ __forceinline__ f32x4 mma16(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}

__device__ __forceinline__ int lds_off(int row, int chunk) {   // byte offset inside a tile of 128-byte rows
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// Fused epilogue for 4 consecutive output channels n..n+3 of tile pixel mt (shared by both kernels):
//   y = residual + alpha * drop(act(acc + bias + rowadd[b]))   or, with dact_u, the backward form * act'(u)
// Tile pixel mt -> row m of y and its sample index b (one decode per PIXEL, shared by its channel groups)
template <int MODE>
__device__ __forceinline__ void conv_out_row(const ConvP& p, int mt, int& m, int& b) {
    if (MODE == 3) {                                      // parity class: tile pixel -> row of y
        const int hw = p.sub_nH * p.sub_nW;
        b = mt / hw;
        const int rm = mt - b * hw, i = rm / p.sub_nW, jj = rm - i * p.sub_nW;
        m = (b * p.Ho + p.sub_h0 + 2 * i) * p.Wo + p.sub_w0 + 2 * jj;
    } else {
        m = mt;
        b = p.rowadd ? mt / (p.Ho * p.Wo) : 0;
    }
}

// Fused epilogue for 4 consecutive output channels n..n+3 of y row m (sample b):
//   y = residual + alpha * drop(act(acc + bias + rowadd[b]))   or, with dact_u, the backward form * act'(u)
// `bias4`, `ra4` (per-sample add) and `res4` are loaded by the caller BEFORE the stores they would otherwise trail: vmcnt retires in order, so a load
// issued after a store cannot complete before that store has - with the loads inside this function every channel
// group waited a full store round trip (16 per lane), which made every epilogue with a bias 1.4x slower than one
// without (fwd vs dgrad of the same GEMM).
// value(s) of the fused epilogue for 4 consecutive channels: `v` becomes the output value; returns in `pre` what the
// `preact` buffer receives (the pre-activation u, or with PSG_CONV_SAVE_DACT the epilogue's derivative)
template <typename T>
__device__ __forceinline__ void conv_value(const ConvP& p, int m, int n, f32x4& v, f32x4& pre, f32x4 bias4, f32x4 ra4, f32x4 aux4, uint64_t dseed) {
    // aux4: the residual, or (backward form, dact_u set - the two are mutually exclusive) the saved pre-activation u /
    // the saved epilogue derivative
    constexpr bool FAST = sizeof(T) == 2;                  // bf16 compute: bf16-grade GELU (exact fp32 path keeps erff)
    v += bias4 + ra4;
    const bool save_d = (p.flags & PSG_CONV_SAVE_DACT) != 0;
    f32x4 d = {1.f, 1.f, 1.f, 1.f};                        // d(epilogue value)/d(accumulator), apart from alpha
    pre = v;
    if (p.dact_u) {
        if (p.flags & PSG_CONV_DACT_MUL) v *= aux4;       // backward form, derivative saved by the forward launch
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= act_grad(aux4[e], p.act);
        }
    } else if (p.act != PSG_ACT_NONE) {
        if (save_d || FAST) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float y, dy; act_both<FAST>(v[e], p.act, y, dy); v[e] = y; d[e] = dy; }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_f(v[e], p.act);
        }
    }
    if (p.drop_thresh) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool keep = drop_keep(dseed, (uint64_t)m * p.N + n + e, p.drop_thresh);
            v[e] = keep ? v[e] * p.drop_scale : 0.f;
            d[e] = keep ? d[e] * p.drop_scale : 0.f;
        }
    }
    if (save_d) pre = d;
    v *= p.alpha;
    if (!p.dact_u) v += aux4;
}

// Epilogue kinds of the bf16 kernels.  conv_value decides everything per element at run time (which activation, dropout,
// backward form): unrolled over a lane's 64+ accumulator cells that is a maze of scalar branches around code for five
// activations - tens of thousands of instructions per kernel, of which a launch executes a thin, scattered slice.  The
// LDS-staged epilogue is therefore instantiated once per KIND, chosen once per launch: each copy holds only its own
// arithmetic, branch-free.  EK_GENERIC keeps the run-time form (SiLU / ReLU / tanh, activation-gradient backward forms,
// and every fp32 launch: the exact path's arithmetic order does not change).
enum { EK_PLAIN = 0, EK_DROP = 1, EK_GELU = 2, EK_GELU_DROP = 3, EK_DMUL = 4, EK_GENERIC = 5 };
// AUX_LATER: stop before the aux operand (the residual, or EK_DMUL's saved derivative) enters - conv_aux_k applies it
// later to the same fp32 value, so the result is the one-step form's bit for bit.
template <typename T, int EK, bool AUX_LATER = false>
__device__ __forceinline__ void conv_value_k(const ConvP& p, int m, int n, f32x4& v, f32x4& pre, f32x4 bias4, f32x4 ra4, f32x4 aux4, uint64_t dseed) {
    if constexpr (EK == EK_GENERIC) { conv_value<T>(p, m, n, v, pre, bias4, ra4, aux4, dseed); return; }
    else {
        v += bias4 + ra4;
        pre = v;
        if constexpr (EK == EK_DMUL) {                    // backward form, derivative saved by the forward launch; no dropout
            if constexpr (!AUX_LATER) {
                v *= aux4;
                v *= p.alpha;
            }
            return;
        }
        f32x4 d = {1.f, 1.f, 1.f, 1.f};
        if constexpr (EK == EK_GELU || EK == EK_GELU_DROP) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float y, dy; gelu_both_fast(v[e], y, dy); v[e] = y; d[e] = dy; }
        }
        if constexpr (EK == EK_DROP || EK == EK_GELU_DROP) {
            // n is a multiple of 4 and N of 8: the 4 elements are the pairs (idx>>1, idx>>1 + 1) of drop_keep
            const uint64_t pair = ((uint64_t)m * p.N + n) >> 1;
            const uint32_t h0 = drop_hash_pair(dseed, pair), h1 = drop_hash_pair(dseed, pair + 1);
            const bool k[4] = {drop_keep_half(h0, 0, p.drop_thresh), drop_keep_half(h0, 1, p.drop_thresh),
                               drop_keep_half(h1, 0, p.drop_thresh), drop_keep_half(h1, 1, p.drop_thresh)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = k[e] ? v[e] * p.drop_scale : 0.f;
                d[e] = k[e] ? d[e] * p.drop_scale : 0.f;
            }
        }
        if (p.flags & PSG_CONV_SAVE_DACT) pre = d;
        v *= p.alpha;
        if constexpr (!AUX_LATER) v += aux4;
    }
}
// the aux step of conv_value_k<.., AUX_LATER = true> on one value
template <int EK>
__device__ __forceinline__ float conv_aux_k(const ConvP& p, float v, float aux) {
    if constexpr (EK == EK_DMUL) return (v * aux) * p.alpha;
    else return v + aux;
}

template <typename T>
__device__ __forceinline__ void conv_emit(const ConvP& p, int m, int n, f32x4 v, f32x4 bias4, f32x4 ra4, f32x4 aux4, uint64_t dseed) {
    T* yg = reinterpret_cast<T*>(p.y);
    T* preg = reinterpret_cast<T*>(p.preact);
    f32x4 pre;
    conv_value<T>(p, m, n, v, pre, bias4, ra4, aux4, dseed);
    if (preg) store4<T>(preg + (int64_t)m * p.ldpre + n, pre);
    store4<T>(yg + (int64_t)m * p.ldy + n, v);
}

// y row of tile pixel mt (no sample index): the coalesced store pass of the LDS-staged epilogue
template <int MODE>
__device__ __forceinline__ int conv_out_m(const ConvP& p, int mt) {
    if (MODE == 3) {
        const int hw = p.sub_nH * p.sub_nW;
        const int b = mt / hw, rm = mt - b * hw, i = rm / p.sub_nW, jj = rm - i * p.sub_nW;
        return (b * p.Ho + p.sub_h0 + 2 * i) * p.Wo + p.sub_w0 + 2 * jj;
    }
    return mt;
}

// MODE 0: forward gather, K step inside one tap (Cin % K-step == 0)   [every 3x3 / 1x1 layer of the U-Net body]
// MODE 1: data-gradient gather of a stride-1 conv, same fast decode
// MODE 2: generic (Cin = 8 first/last convs, odd channel counts): per-thread tap decode
// MODE 3: stride-2 data gradient, one launch per output-pixel parity class: on the class's own (i, j) grid the
//         gather is a stride-1 gather with 1, 2 or 4 taps (table in ConvP) instead of 9 taps of which 3/4 miss
// 256 threads = 4 waves (2 along m x 2 along n), two LDS tile buffers: the tile of step kt+1 is requested at the top
// of step kt (plain __syncthreads at the end of the step).
// SPLITK: the split-K form (small output grids) is its OWN instantiation - it has no epilogue at all, and the epilogue-heavy
// plain kernels keep their register allocation (as a run-time branch inside them it cost 274 more spilled registers and
// 17 % of the forward / data-gradient families).
template <typename T, int BM, int BN, int MODE, bool SPLITK = false>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const ConvP p) {
#if defined(__HIP_DEVICE_COMPILE__)      // the LDS-DMA builtin exists only in the device pass
    constexpr int NT = 256;
    constexpr int CH = Elem<T>::CH;
    constexpr int WM = BM / 2, WN = BN / 2;           // pixels / channels per wave (BN = 160 -> 80 = 5 x 16: bf16 only)
    constexpr int RPP = NT / 8;                 // rows per staging pass (8 lanes x 16 B per 128-byte row)
    constexpr int JX = BM / RPP, JW = BN / RPP; // staging passes
    constexpr int PASS_BYTES = RPP * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [buf][W tile BN rows | X tile BM rows], 128 B per row
    constexpr int BUF_BYTES = (BM + BN) * 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, wm = wave >> 1;
    // Ablation / timeline builds (tools/build_abl.sh; never the shipped library): PSG_ABL bit 0 = one K step only, bit 1 = no
    // epilogue, bit 3 = per-workgroup phase timestamps into the buffer PSG_DBG_PTR names (tools/conv_timeline.py)
#ifndef PSG_ABL
#define PSG_ABL 0
#endif
    uint64_t* dbg = nullptr;
    if ((PSG_ABL & 8) && !SPLITK && p.ws) { dbg = reinterpret_cast<uint64_t*>(p.ws) + (size_t)blockIdx.x * 8; if (tid == 0) { dbg[0] = wall_clock64(); dbg[5] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); dbg[6] = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (7 << 11)); } }

    // ---- XCD-aware tile id + grouped raster -------------------------------
    int mt, nt, split = 0;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7;
        int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        if constexpr (SPLITK) { const int tiles = p.mtiles * p.ntiles; split = lid / tiles; lid -= split * tiles; }   // split slowest
        constexpr int GM = 8;
        const int per_group = GM * p.ntiles;
        const int g = lid / per_group, rem = lid - g * per_group;
        const int gm = min(GM, p.mtiles - g * GM);
        nt = rem / gm;
        mt = g * GM + (rem - nt * gm);
    }
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- per-thread staging coordinates ------------------------------------
    // Tiles go global -> LDS by LDS-DMA (buffer_load ... lds): the destination of a wave-instruction is
    // wave-uniform base + lane*16, i.e. 8 consecutive 128-byte rows, so the XOR swizzle is applied on the SOURCE
    // side: the lane sitting at physical chunk (tid & 7) of row sr fetches logical chunk sc.
    const int sr = tid >> 3;                          // row within a 32-row pass
    const int sc = (tid & 7) ^ ((sr >> 1) & 7);       // logical chunk this lane fetches ((row>>1)&7 is pass-invariant)

    // Branch-free gathers: raw buffer loads return 0 for an out-of-range offset, so padding taps,
    // rows beyond M / N and the K tail cost no control flow (and no per-load s_waitcnt).
    constexpr uint32_t OOB = 0x80000000u;          // extents are < 2 GiB (checked on the host)
    constexpr int ESZ = (int)sizeof(T);
    const u32x4 xrs = make_rsrc(p.x, p.x_bytes);
    const u32x4 wrs = make_rsrc(p.w, p.w_bytes);
    uint32_t w_off[JW];              // byte offset of (row, chunk sc), or OOB
#pragma unroll
    for (int j = 0; j < JW; ++j) {
        const int n = n0 + sr + RPP * j;
        w_off[j] = n < p.N ? (uint32_t)(((int64_t)n * p.ldw + sc * CH) * ESZ) : OOB;
    }
    int x_hb[JX], x_wb[JX], x_base[JX];   // tap-0 source row / col, and byte offset of that pixel (+ chunk sc)
    {
        const int HoWo = p.Ho * p.Wo;
        const bool small_m = p.M < (1 << 24);
        const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)p.Wo;
#pragma unroll
        for (int j = 0; j < JX; ++j) {
            const int m = m0 + sr + RPP * j;
            if (MODE == 3) {
                if (m < p.M) {
                    const int hw = p.sub_nH * p.sub_nW;
                    const int b = m / hw, rm = m - b * hw;
                    const int i = rm / p.sub_nW, jj = rm - i * p.sub_nW;
                    x_hb[j] = i; x_wb[j] = jj;
                    x_base[j] = ((b * p.Hi + i) * p.Wi + jj) * (int)p.ldx * ESZ + sc * 16;
                } else { x_hb[j] = -100000; x_wb[j] = -100000; x_base[j] = 0; }
            } else if (MODE != 2 && p.taps == 1 && p.stride == 1 && m < p.M) {
                // pointwise (every Linear, the 1x1 convs): the source pixel IS the output pixel - no (b, ho, wo) decode
                // (two integer divisions per staging pass in a prologue that a K <= 1280 tile cannot amortise)
                x_hb[j] = 0; x_wb[j] = 0;
                x_base[j] = m * (int)p.ldx * ESZ + sc * 16;
            } else if (m < p.M) {
                // (float-reciprocal division, exact below 2^24: the two integer divisions per staging pass were a visible part
                //  of every 3x3 tile's start - tools/conv_timeline.py)
                const int b = small_m ? fastdiv(m, HoWo, inv_howo) : m / HoWo, rm = m - b * HoWo;
                const int ho = small_m ? fastdiv(rm, p.Wo, inv_wo) : rm / p.Wo, wo = rm - ho * p.Wo;
                if (p.transposed) { x_hb[j] = ho + p.pad; x_wb[j] = wo + p.pad; }
                else { x_hb[j] = ho * p.stride - p.pad; x_wb[j] = wo * p.stride - p.pad; }
                if (MODE == 2) x_base[j] = b * p.Hi * p.Wi;                       // pixel base only
                else x_base[j] = ((b * p.Hi + x_hb[j]) * p.Wi + x_wb[j]) * (int)p.ldx * ESZ + sc * 16;
            } else { x_hb[j] = -100000; x_wb[j] = -100000; x_base[j] = 0; }
        }
    }

    // this workgroup's K steps [kt0, kt1) (the whole K axis unless split-K)
    const int kt0 = SPLITK ? split * p.kt_per_split : 0;
    const int kt1 = (PSG_ABL & 1) ? kt0 + 1 : (SPLITK ? min(p.KT, kt0 + p.kt_per_split) : p.KT);
    // uniform tap state of the NEXT K step to load (fast modes): chunk offset inside the tap, kh, kw - at step kt0 of the
    // taps-innermost walk (MODE 3: t_kh indexes the class's tap table)
    int t_c0 = 0, t_kh = 0, t_kw = 0;
    if (SPLITK && kt0 > 0) {
        if (MODE == 3) { t_c0 = (kt0 / p.ntap) * 8; t_kh = kt0 % p.ntap; }
        else if (MODE != 2) { const int tp = kt0 % p.taps; t_c0 = (kt0 / p.taps) * 8; t_kh = tp / p.ks; t_kw = tp - t_kh * p.ks; }
    }
    // wave-uniform LDS byte offset of this wave's 1 KiB slot in pass 0 (rows 8*wave .. 8*wave+7)
    typedef __attribute__((address_space(3))) char* lds_ptr_t;
    const uint32_t lds_wave = (uint32_t)(size_t)(lds_ptr_t)smem + (uint32_t)__builtin_amdgcn_readfirstlane(wave) * 1024u;
    auto load_tiles = [&](int kt, int buf) {          // issue the LDS-DMA of K step kt into LDS buffer buf
        // weight K offset: 8 chunks of 16 B per K step; MODE 3 walks only its own taps of the 9-tap K axis
        // fast modes 0/1 walk K with the TAPS INNERMOST: for one 64-channel slice the 9 taps re-read the same pixels
        // shifted by a row / a column, back to back, so 8 of the 9 gathers hit L2 (tap-outermost spreads them a whole
        // channel sweep apart and every one misses): +2..9 % on the 3x3 layers.  The weight K offset follows the walk.
        const uint32_t kbytes = MODE == 3 ? (uint32_t)(p.tap_wi[t_kh] * p.cpt + t_c0) * 16u
                              : (MODE == 2 ? (uint32_t)kt * 128u : (uint32_t)((t_kh * p.ks + t_kw) * p.cpt + t_c0) * 16u);
        const uint32_t wdst = lds_wave + (uint32_t)buf * BUF_BYTES;
        const uint32_t xdst = wdst + BN * 128;
#pragma unroll
        for (int j = 0; j < JW; ++j)
            lds_dma16(wrs, wdst + j * PASS_BYTES, w_off[j] + kbytes);
        if (MODE == 3) {                                   // t_kh is the index into the class's tap table
            const int dh = p.tap_dh[t_kh], dw = p.tap_dw[t_kh];
            const int delta = ((dh * p.Wi + dw) * (int)p.ldx + t_c0 * CH) * ESZ;                 // wave-uniform
#pragma unroll
            for (int j = 0; j < JX; ++j) {
                // (inside-the-image test without compares: the sign bits of the four differences, OR-ed, go into bit 31 of
                //  the offset = out of range for the buffer load; no VALU -> VCC -> select chain per piece)
                const int sh = x_hb[j] + dh, sw = x_wb[j] + dw;
                const int bad = sh | (p.Hi - 1 - sh) | sw | (p.Wi - 1 - sw);
                const uint32_t off = (uint32_t)(x_base[j] + delta) | ((uint32_t)bad & OOB);
                lds_dma16(xrs, xdst + j * PASS_BYTES, off);
            }
            if (++t_kh == p.ntap) { t_kh = 0; t_c0 += 8; }       // taps innermost, as in modes 0/1
        } else if (MODE != 2) {
            const int sgn = MODE == 0 ? 1 : -1;
            const int delta = (sgn * (t_kh * p.Wi + t_kw) * (int)p.ldx + t_c0 * CH) * ESZ;      // wave-uniform
#pragma unroll
            for (int j = 0; j < JX; ++j) {
                const int sh = x_hb[j] + sgn * t_kh, sw = x_wb[j] + sgn * t_kw;
                const int bad = sh | (p.Hi - 1 - sh) | sw | (p.Wi - 1 - sw);
                const uint32_t off = (uint32_t)(x_base[j] + delta) | ((uint32_t)bad & OOB);
                lds_dma16(xrs, xdst + j * PASS_BYTES, off);
            }
            if (++t_kw == p.ks) { t_kw = 0; if (++t_kh == p.ks) { t_kh = 0; t_c0 += 8; } }
        } else {
            const int qi = kt * 8 + sc;
            const int tap = qi / p.cpt, cc = qi - tap * p.cpt;
            const bool tap_ok = tap < p.taps;
            const int kh = tap / p.ks, kw = tap - kh * p.ks;
#pragma unroll
            for (int j = 0; j < JX; ++j) {
                int sh, sw; bool ok = tap_ok;
                if (p.transposed) {
                    const int th = x_hb[j] - kh, tw = x_wb[j] - kw;
                    if (p.stride == 1) { sh = th; sw = tw; }
                    else { sh = th >> 1; sw = tw >> 1; ok = ok && ((th | tw) & 1) == 0; }
                    ok = ok && th >= 0 && tw >= 0 && sh < p.Hi && sw < p.Wi;
                } else {
                    sh = x_hb[j] + kh; sw = x_wb[j] + kw;
                    ok = ok && sh >= 0 && sw >= 0 && sh < p.Hi && sw < p.Wi;
                }
                uint32_t off = (uint32_t)(((x_base[j] + sh * p.Wi + sw) * (int)p.ldx + cc * CH) * ESZ);
                off = ok ? off : OOB;
                lds_dma16(xrs, xdst + j * PASS_BYTES, off);
            }
        }
    };

    constexpr bool FT16 = sizeof(T) == 2;                 // bf16: 16x16x32 tiles; fp32: 32x32x2 tiles
    constexpr int FT = FT16 ? 16 : 32;
    static_assert(WN % FT == 0 && WM % FT == 0, "wave tile must be a whole number of MFMA tiles");
    constexpr int NA = WN / FT, NB = WM / FT;             // fragment tiles per wave along n / m
    constexpr int AE = FT16 ? 4 : 16;
    typedef float AccT __attribute__((ext_vector_type(AE)));
    AccT acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < AE; ++e) acc[i][j][e] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;             // 32x32 fragment coordinates (fp32 path)
    const int l16 = lane & 15, kq = lane >> 4;            // 16x16x32 fragment coordinates (bf16 path)
    // fragment row bases (bytes) and chunk swizzle key; tile rows step by 16 / 32, which leaves (row>>1)&7 unchanged
    const int frow = FT16 ? l16 : fr;
    static_assert((WN / 2) % 8 == 0 && (WM / 2) % 8 == 0, "wave row bases must keep the (row>>1)&7 swizzle key");
    const int rd_w = (wn * WN + frow) * 128, rd_x = BN * 128 + (wm * WM + frow) * 128;
    const int swz = (frow >> 1) & 7;
    constexpr int FSTEP = (FT16 ? 16 : 32) * 128;         // byte step between fragment tiles

    auto compute = [&](int buf) {
        const char* tb = smem + buf * BUF_BYTES;
        if constexpr (FT16) {
            // all 2 x (NA + NB) fragment reads of the K tile are issued up front (64 VGPRs); the MFMAs start as the
            // first ones land and the rest of the LDS latency hides under the 2 x NA x NB MFMA stream
            uint4 wf[2][NA], xf[2][NB];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int co = ((4 * s2 + kq) ^ swz) << 4;
#pragma unroll
                for (int i = 0; i < NA; ++i) wf[s2][i] = *reinterpret_cast<const uint4*>(tb + rd_w + i * FSTEP + co);
#pragma unroll
                for (int j = 0; j < NB; ++j) xf[s2][j] = *reinterpret_cast<const uint4*>(tb + rd_x + j * FSTEP + co);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the reads ABOVE the MFMA stream (hipcc otherwise sinks them)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < NA; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc[i][j] = mma16(wf[s2][i], xf[s2][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // fragment reads of sub-step s+1 are issued BEFORE the MFMAs of sub-step s (two fragment register sets),
            // so the ~128-cycle LDS latency hides under the MFMA cluster instead of idling the matrix pipe
            uint4 wf[2][NA], xf[2][NB];
#pragma unroll
            for (int i = 0; i < NA; ++i) wf[0][i] = *reinterpret_cast<const uint4*>(tb + rd_w + i * FSTEP + ((fh ^ swz) << 4));
#pragma unroll
            for (int j = 0; j < NB; ++j) xf[0][j] = *reinterpret_cast<const uint4*>(tb + rd_x + j * FSTEP + ((fh ^ swz) << 4));
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int cur = s & 1, nxt = cur ^ 1;
                if (s < 3) {
                    const int c = 2 * (s + 1) + fh;
#pragma unroll
                    for (int i = 0; i < NA; ++i) wf[nxt][i] = *reinterpret_cast<const uint4*>(tb + rd_w + i * FSTEP + ((c ^ swz) << 4));
#pragma unroll
                    for (int j = 0; j < NB; ++j) xf[nxt][j] = *reinterpret_cast<const uint4*>(tb + rd_x + j * FSTEP + ((c ^ swz) << 4));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NA; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc[i][j] = Mma<T>::run(wf[cur][i], xf[cur][j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    load_tiles(kt0, 0);
    // Epilogue operands that do not depend on the GEMM (bias, residual, output-row decode) are requested NOW, right
    // behind the first tile's DMA: their latency disappears under the K loop instead of being exposed after it.
    // (A residual that aliases y is still read before this workgroup - the only writer of these rows - stores.)
    // (aux operand of the epilogue: the residual, or the saved pre-activation of the backward form)
    const T* resg = reinterpret_cast<const T*>(p.dact_u ? p.dact_u : p.residual);
    const int64_t ldaux = p.dact_u ? p.lddact : p.ldres;
    const T* rag = reinterpret_cast<const T*>(p.rowadd);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    int rows[FT16 ? NB : 1], smp[FT16 ? NB : 1];
    f32x4 bias4[FT16 ? NA : 1];
    bf16x4 res_raw[FT16 ? NA : 1][FT16 ? NB : 1];
    // LDS-staged epilogue: the residual / saved derivative is fetched ROW-major instead (16 bytes per lane, whole 128-byte
    // lines per instruction) and turned into the accumulator layout through the wave's LDS region - exact, it is bf16
    constexpr int E_CPRW = FT16 ? WN / 8 : 1, E_NCH = FT16 ? WM * E_CPRW : 1, E_NIT = (E_NCH + 63) / 64;
    uint4 res_row[FT16 ? E_NIT : 1];
    if constexpr (FT16) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int mt_ = m0 + wm * WM + j * 16 + l16;
            rows[j] = -1; smp[j] = 0;
            if (mt_ < p.M) conv_out_row<MODE>(p, mt_, rows[j], smp[j]);
            if (!SPLITK && resg && !p.epi_lds) {
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int n = n0 + wn * WN + i * 16 + 4 * kq;
                    if (rows[j] >= 0 && n < p.N) res_raw[i][j] = *reinterpret_cast<const bf16x4*>(resg + (int64_t)rows[j] * ldaux + n);
                }
            }
        }
    }
    wait_vmcnt<0>();
    __syncthreads();                       // the DMA of tile 0 has landed for every wave
    if ((PSG_ABL & 8) && dbg && tid == 0) dbg[1] = wall_clock64();
    for (int kt = kt0; kt + 1 < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        // buffer buf^1 was last read in step kt-1, which every wave finished before the barrier below
        load_tiles(kt + 1, buf ^ 1);
        compute(buf);
        wait_vmcnt<0>();                   // (asm LDS-DMA is invisible to hipcc: the wait is ours)
        __syncthreads();                   // tile kt+1 is in LDS, tile kt no longer needed
    }
    // Last K step (peeled): no tile is left to request, so the staged epilogue's row-major residual / saved-derivative
    // fetch goes out NOW and flies under the step's MFMAs.  vmcnt retires in order: issued any earlier these HBM-latency
    // loads sit in front of a tile DMA and the wait for that tile waits for them too (behind the first tile, as before,
    // they stretched every workgroup's prologue: a residual cost a K<=1280 Linear 35-50 % more than its bandwidth).
    if constexpr (SPLITK) {
        // split-K: this workgroup's share of the reduction goes out as an fp32 partial tile; the epilogue runs in the
        // finishing kernel (fixed summation order over the splits: deterministic)
        compute((kt1 - 1 - kt0) & 1);
        float* wsb = p.ws + (int64_t)split * p.M * p.N;
        if constexpr (FT16) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int mt_ = m0 + wm * WM + j * 16 + l16;
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int n = n0 + wn * WN + i * 16 + 4 * kq;
                    if (mt_ < p.M && n < p.N) *reinterpret_cast<f32x4*>(wsb + (int64_t)mt_ * p.N + n) = acc[i][j];
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int mt_ = m0 + wm * WM + j * 32 + fr;
#pragma unroll
                for (int i = 0; i < NA; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = n0 + wn * WN + i * 32 + 8 * g + 4 * fh;
                        if (mt_ < p.M && n < p.N) {
                            const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                            *reinterpret_cast<f32x4*>(wsb + (int64_t)mt_ * p.N + n) = v;
                        }
                    }
            }
        }
        return;
    } else {
    if constexpr (FT16) {
        // (the bias too: even as L2 hits its loads, issued behind the first tile, held every workgroup's first wait for
        //  0.5-1 us - a bias cost a K = 640 Linear 12-18 %)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int n = n0 + wn * WN + i * 16 + 4 * kq;
            bias4[i] = (p.bias && n < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + n) : zero4;
        }
        if (resg && p.epi_lds) {
#pragma unroll
            for (int it = 0; it < E_NIT; ++it) {
                const int idx = it * 64 + lane;
                const int row = idx / E_CPRW, chk = idx - row * E_CPRW;
                const int mt_ = m0 + wm * WM + row, n = n0 + wn * WN + chk * 8;
                // (an unconditional load from a clamped address + a select on the VALUE: as `ok ? *ptr : zero` hipcc selected
                //  between the pointer and the address of a zero in scratch memory and issued a flat load)
                const bool ok = idx < E_NCH && mt_ < p.M && n < p.N;
                const uint4 got = *reinterpret_cast<const uint4*>(resg + (ok ? (int64_t)conv_out_m<MODE>(p, mt_) * ldaux + n : (int64_t)0));
                res_row[it].x = ok ? got.x : 0u; res_row[it].y = ok ? got.y : 0u; res_row[it].z = ok ? got.z : 0u; res_row[it].w = ok ? got.w : 0u;
            }
        }
    }
    const uint64_t dseed = p.drop_seed + ((p.drop_thresh && p.seed_dev) ? *p.seed_dev : 0ull);     // wave-uniform (requested here: flies under the last step)
    compute((kt1 - 1 - kt0) & 1);
    if (PSG_ABL & 2) {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int e = 0; e < AE; ++e) sacc += acc[i][j][e];
        if (sacc == 12345.678f) reinterpret_cast<float*>(p.y)[tid] = sacc;
        return;
    }
    __syncthreads();                       // every wave is done with the tile buffers: the epilogue reuses them
    if ((PSG_ABL & 8) && dbg && tid == 0) dbg[2] = wall_clock64();

    // ---- fused epilogue: phase 1 loads every bias / residual operand of the tile, phase 2 computes and stores ----
    if constexpr (FT16) {
        if (p.epi_lds) {
            // LDS-staged stores.  In the accumulator layout a lane owns 4 channels of one pixel, so a store instruction
            // writes 64 x 8 bytes in 32-byte pieces of 16 different rows: the store pipe, not the bandwidth, bounds the
            // epilogue (and the epilogue, not the K loop, bounds every short-K layer).  Each wave instead writes its
            // WM x WN sub-tile to its own LDS region (bf16, padded rows: conflict-free ds_write_b64) and reads it back
            // row-major, 16 bytes per lane: a store instruction then writes whole 128-byte lines, half as many of them.
            // Values are final (bias, activation, dropout, gate, residual all applied in fp32 BEFORE the staging: one
            // rounding to bf16, bit-identical to the direct form).  `preact` takes the same route in a first pass.
            constexpr int CPRW = WN / 8;                       // 16-byte chunks per sub-tile row
            constexpr int PITCH = WN * 2 + 16;                 // bytes; (PITCH/4) mod 64 = 36 / 20 / 44: distinct banks for 16 rows
            constexpr int NCH = WM * CPRW;                     // chunks per sub-tile
            constexpr int NIT = (NCH + 63) / 64;
            char* reg = smem + wave * (WM * PITCH);
            // The epilogue runs once per workgroup as straight-line code with nothing to hide behind (both workgroups of a CU
            // reach it together: tools/conv_timeline.py), so its INSTRUCTION COUNT is its time: ~850 instructions took 2.6 us of
            // a 13 us K = 640 tile.  Two things made most of them: a per-cell exec-mask branch around each per-sample-add load
            // (taken or not, ~10 instructions per cell, and the taken form waited for every load on its own), and a flush loop
            // that predicated, read LDS, waited and built a 64-bit address once per store.
            // (a) per-sample add of pixel row j: ONE wave-uniform branch; inside it the loads are unconditional (clamped to a
            //     valid address) and batched, the validity test is a select afterwards
            auto load_ra = [&](int j, f32x4 (&ra4)[NA]) {
                if (rag) {
                    const bool rok = rows[j] >= 0;
                    const T* rp = rag + (int64_t)(rok ? smp[j] : 0) * p.ldra;
#pragma unroll
                    for (int i = 0; i < NA; ++i) {
                        const int n = n0 + wn * WN + i * 16 + 4 * kq;
                        ra4[i] = load4<T>(rp + (n < p.N ? n : 0));
                    }
#pragma unroll
                    for (int i = 0; i < NA; ++i) {
                        const int n = n0 + wn * WN + i * 16 + 4 * kq;
                        if (!(rok && n < p.N)) ra4[i] = zero4;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < NA; ++i) ra4[i] = zero4;
                }
            };
            // (b) flush of NITX x 64 row-major 16-byte chunks of the wave's region (rows of PITCHX bytes, the first one tile row
            //     `rbase` of the wave's sub-tile) to `dst`: a tile that lies wholly inside y (wave-uniform test; every tile of
            //     the U-Net's training shapes) reads all chunks first and stores them through one 64-bit row base + 32-bit
            //     offsets, without predicates
            const bool tile_full = MODE != 3 && m0 + BM <= p.M && n0 + BN <= p.N;
            auto flush_rows = [&](auto nchx_c, const char* src, int rbase, T* dst, int64_t ldd) {
                constexpr int NCHX = decltype(nchx_c)::value, NITX = (NCHX + 63) / 64;
                if (tile_full && ldd < (1 << 22)) {                 // (32-bit element offsets inside the tile: 128 rows x ldd)
                    T* base = dst + (int64_t)(m0 + wm * WM + rbase) * ldd + (n0 + wn * WN);
                    constexpr int G = 4;                            // chunks in flight per lane (registers: the accumulators may still be live)
                    constexpr bool REG = 64 % CPRW == 0;            // (64-wide sub-tiles: iteration `it` is rows it*8 .. it*8+7, same chunk per lane)
                    T* lane0 = base + (uint32_t)((lane / CPRW) * (int)ldd + (lane % CPRW) * 8);
#pragma unroll
                    for (int g0 = 0; g0 < NITX; g0 += G) {
                        uint4 v0, v1, v2, v3;
                        auto rd = [&](int it) {
                            const int idx = it * 64 + lane;
                            const int row = idx / CPRW, chk = idx - row * CPRW;
                            return *reinterpret_cast<const uint4*>(src + (NCHX % 64 == 0 || idx < NCHX ? row : 0) * PITCH + chk * 16);
                        };
                        auto wr = [&](int it, const uint4& v) {
                            const int idx = it * 64 + lane;
                            const int row = idx / CPRW, chk = idx - row * CPRW;
                            T* d = REG ? lane0 + (int64_t)(it * (64 / CPRW)) * ldd : base + (uint32_t)(row * (int)ldd + chk * 8);
                            if (NCHX % 64 == 0 || idx < NCHX) *reinterpret_cast<uint4*>(d) = v;
                        };
                        if (g0 + 0 < NITX) v0 = rd(g0 + 0);
                        if (g0 + 1 < NITX) v1 = rd(g0 + 1);
                        if (g0 + 2 < NITX) v2 = rd(g0 + 2);
                        if (g0 + 3 < NITX) v3 = rd(g0 + 3);
                        if (g0 + 0 < NITX) wr(g0 + 0, v0);
                        if (g0 + 1 < NITX) wr(g0 + 1, v1);
                        if (g0 + 2 < NITX) wr(g0 + 2, v2);
                        if (g0 + 3 < NITX) wr(g0 + 3, v3);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < NITX; ++it) {
                        const int idx = it * 64 + lane;
                        const int row = idx / CPRW, chk = idx - row * CPRW;
                        const int mt_ = m0 + wm * WM + rbase + row, n = n0 + wn * WN + chk * 8;
                        if (idx < NCHX && mt_ < p.M && n < p.N) {
                            const uint4 val = *reinterpret_cast<const uint4*>(src + row * PITCH + chk * 16);
                            *reinterpret_cast<uint4*>(dst + (int64_t)conv_out_m<MODE>(p, mt_) * ldd + n) = val;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            auto epilogue = [&](auto ek_c) {
            constexpr int EK = decltype(ek_c)::value;
            auto stage_and_flush = [&](auto pass_c) {
                constexpr int PASS = decltype(pass_c)::value;      // 0: the `preact` tensor, 1: y
                if (resg) {                               // residual rows -> LDS (read back per accumulator cell below)
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int idx = it * 64 + lane;
                        const int row = idx / CPRW, chk = idx - row * CPRW;
                        if (idx < NCH) *reinterpret_cast<uint4*>(reg + row * PITCH + chk * 16) = res_row[it];
                    }
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    f32x4 ra4[NA];
                    load_ra(j, ra4);
#pragma unroll
                    for (int i = 0; i < NA; ++i) {
                        const int n = n0 + wn * WN + i * 16 + 4 * kq;
                        f32x4 v = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                        f32x4 r4 = zero4, pre;
                        char* cell = reg + (j * 16 + l16) * PITCH + (i * 16 + 4 * kq) * 2;     // this lane's 4 channels of pixel j*16+l16
                        if (resg) {
                            const bf16x4 rr = *reinterpret_cast<const bf16x4*>(cell);
                            r4[0] = (float)rr[0]; r4[1] = (float)rr[1]; r4[2] = (float)rr[2]; r4[3] = (float)rr[3];
                        }
                        conv_value_k<T, EK>(p, rows[j], n, v, pre, bias4[i], ra4[i], r4, dseed);
                        const f32x4 o = PASS == 0 ? pre : v;
                        bf16x4 ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
                        *reinterpret_cast<bf16x4*>(cell) = ob;
                    }
                    __builtin_amdgcn_sched_barrier(0);          // one pixel row of tiles at a time (register pressure)
                }
                if ((PSG_ABL & 8) && dbg && tid == 0) dbg[7] = wall_clock64();
                flush_rows(std::integral_constant<int, NCH>{}, reg, 0, reinterpret_cast<T*>(PASS == 0 ? p.preact : p.y), PASS == 0 ? p.ldpre : p.ldy);
            };
            // `preact` AND y without a residual (the FFN's first GEMM: GELU value + saved derivative): one evaluation of the
            // epilogue per element instead of one per pass - the sub-tile goes out in two halves of WM/2 pixels, each half
            // staging both tensors side by side in the wave's region.
            auto stage_both = [&]() {
                constexpr int HR = WM / 2, NBH = NB / 2;           // pixels / fragment rows per half
                constexpr int NCHH = HR * CPRW, NITH = (NCHH + 63) / 64;
                char* regB = reg + HR * PITCH;
                T* dpre = reinterpret_cast<T*>(p.preact);
                T* dy = reinterpret_cast<T*>(p.y);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int jj = 0; jj < NBH; ++jj) {
                        const int j = h * NBH + jj;
                        f32x4 ra4[NA];
                        load_ra(j, ra4);
#pragma unroll
                        for (int i = 0; i < NA; ++i) {
                            const int n = n0 + wn * WN + i * 16 + 4 * kq;
                            f32x4 v = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                            f32x4 pre;
                            conv_value_k<T, EK>(p, rows[j], n, v, pre, bias4[i], ra4[i], zero4, dseed);
                            const int off = (jj * 16 + l16) * PITCH + (i * 16 + 4 * kq) * 2;
                            bf16x4 pb = {(bf16_t)pre[0], (bf16_t)pre[1], (bf16_t)pre[2], (bf16_t)pre[3]};
                            bf16x4 vb = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                            *reinterpret_cast<bf16x4*>(reg + off) = pb;
                            *reinterpret_cast<bf16x4*>(regB + off) = vb;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    flush_rows(std::integral_constant<int, NCHH>{}, reg, h * HR, dpre, p.ldpre);
                    flush_rows(std::integral_constant<int, NCHH>{}, regB, h * HR, dy, p.ldy);
                }
            };
            // A residual (or EK_DMUL's saved derivative) WITHOUT `preact`: the operand was fetched row-major (res_row, issued in
            // the last K step).  Turning it into the accumulator layout through LDS put a second dependent round trip
            // (rows -> LDS -> cells) in FRONT of the arithmetic, and the wait for the fetch with it: on the K <= 1280 layers a
            // residual cost 1.4-1.6x a bias-only epilogue.  Here the cells go out first, as fp32 (two halves of WM/2 pixels so
            // that the 4-byte image fits the wave's region), and the operand joins in the ROW-major domain, in fp32, right
            // before the one rounding to bf16: same bits, no round trip, and the fetch has the whole staging to land.
            auto stage_aux = [&]() {
                constexpr int HR = WM / 2, NBH = NB / 2;
                constexpr int PITCHF = WN * 4 + 16;                 // bytes; (PITCHF/4) mod 64 = 4: the 8-lane groups of ds_write_b128 cover 32 banks
                constexpr int NITH = NIT / 2;                        // row-major chunks (8 channels) per lane and half
                static_assert(HR * PITCHF <= WM * PITCH, "fp32 half image must fit the wave's staging region");
                T* dy = reinterpret_cast<T*>(p.y);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int jj = 0; jj < NBH; ++jj) {
                        const int j = h * NBH + jj;
                        f32x4 ra4[NA];
                        load_ra(j, ra4);
#pragma unroll
                        for (int i = 0; i < NA; ++i) {
                            const int n = n0 + wn * WN + i * 16 + 4 * kq;
                            f32x4 v = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                            f32x4 pre;
                            conv_value_k<T, EK, true>(p, rows[j], n, v, pre, bias4[i], ra4[i], zero4, dseed);
                            *reinterpret_cast<f32x4*>(reg + (jj * 16 + l16) * PITCHF + (i * 16 + 4 * kq) * 4) = v;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    auto join = [&](const f32x4 lo, const f32x4 hi, const uint4 a) {
                        const uint32_t aw[4] = {a.x, a.y, a.z, a.w};
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float a0 = __uint_as_float(aw[e] << 16), a1 = __uint_as_float(aw[e] & 0xFFFF0000u);
                            const float v0 = e < 2 ? lo[2 * e] : hi[2 * e - 4], v1 = e < 2 ? lo[2 * e + 1] : hi[2 * e - 3];
                            o[2 * e] = (bf16_t)conv_aux_k<EK>(p, v0, a0);
                            o[2 * e + 1] = (bf16_t)conv_aux_k<EK>(p, v1, a1);
                        }
                        return o;
                    };
                    if (tile_full && p.ldy < (1 << 22)) {            // (as flush_rows: all reads first, one row base, no predicates)
                        T* base = dy + (int64_t)(m0 + wm * WM + h * HR) * p.ldy + (n0 + wn * WN);
                        f32x4 lo[NITH], hi[NITH];
#pragma unroll
                        for (int it = 0; it < NITH; ++it) {
                            const int idx = it * 64 + lane;
                            const int row = idx / CPRW, chk = idx - row * CPRW;
                            lo[it] = *reinterpret_cast<const f32x4*>(reg + row * PITCHF + chk * 32);
                            hi[it] = *reinterpret_cast<const f32x4*>(reg + row * PITCHF + chk * 32 + 16);
                        }
#pragma unroll
                        for (int it = 0; it < NITH; ++it) {
                            const int idx = it * 64 + lane;
                            const int row = idx / CPRW, chk = idx - row * CPRW;
                            *reinterpret_cast<bf16x8*>(base + (uint32_t)(row * (int)p.ldy + chk * 8)) = join(lo[it], hi[it], res_row[h * NITH + it]);
                        }
                    } else {
#pragma unroll
                        for (int it = 0; it < NITH; ++it) {
                            const int idx = it * 64 + lane;              // chunk inside the half: same (row, chunk) walk as res_row
                            const int row = idx / CPRW, chk = idx - row * CPRW;
                            const int mt_ = m0 + wm * WM + h * HR + row, n = n0 + wn * WN + chk * 8;
                            if (mt_ < p.M && n < p.N) {
                                const f32x4 lo = *reinterpret_cast<const f32x4*>(reg + row * PITCHF + chk * 32);
                                const f32x4 hi = *reinterpret_cast<const f32x4*>(reg + row * PITCHF + chk * 32 + 16);
                                *reinterpret_cast<bf16x8*>(dy + (int64_t)conv_out_m<MODE>(p, mt_) * p.ldy + n) = join(lo, hi, res_row[h * NITH + it]);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            };
            if constexpr (NB % 2 == 0 && EK != EK_GENERIC && MODE != 3 && NIT % 2 == 0 && ((WM / 2) * CPRW) % 64 == 0) {
                if (resg && !p.preact) { stage_aux(); return; }
            }
            if constexpr (NB % 2 == 0 && BN != 160 && MODE != 3) {   // (160-wide and parity-class kernels have no scalar registers to spare)
                if (p.preact && !resg) { stage_both(); return; }
            }
            if (p.preact) stage_and_flush(std::integral_constant<int, 0>{});
            stage_and_flush(std::integral_constant<int, 1>{});
            };
            int ek = EK_GENERIC;
            if (!p.epi_generic) {
                if (p.dact_u) { if ((p.flags & PSG_CONV_DACT_MUL) && !p.drop_thresh) ek = EK_DMUL; }
                else if (p.act == PSG_ACT_NONE) ek = p.drop_thresh ? EK_DROP : EK_PLAIN;
                else if (p.act == PSG_ACT_GELU) ek = p.drop_thresh ? EK_GELU_DROP : EK_GELU;
            }
            // (the parity-class kernels keep the run-time form: their tap tables leave no scalar registers for more copies)
            if constexpr (MODE == 3) { epilogue(std::integral_constant<int, EK_GENERIC>{}); return; }
            else switch (ek) {
                case EK_PLAIN: epilogue(std::integral_constant<int, EK_PLAIN>{}); break;
                case EK_DROP: epilogue(std::integral_constant<int, EK_DROP>{}); break;
                case EK_GELU: epilogue(std::integral_constant<int, EK_GELU>{}); break;
                case EK_GELU_DROP: epilogue(std::integral_constant<int, EK_GELU_DROP>{}); break;
                case EK_DMUL: epilogue(std::integral_constant<int, EK_DMUL>{}); break;
                default: epilogue(std::integral_constant<int, EK_GENERIC>{}); break;
            }
            if ((PSG_ABL & 8) && dbg) { if (tid == 0) dbg[3] = wall_clock64(); wait_vmcnt<0>(); __syncthreads(); if (tid == 0) dbg[4] = wall_clock64(); }
            return;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (rows[j] < 0) continue;
            f32x4 ra4[NA];                                 // per-sample add of this pixel's sample: loads before its stores
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int n = n0 + wn * WN + i * 16 + 4 * kq;
                ra4[i] = (rag && n < p.N) ? load4<T>(rag + (int64_t)smp[j] * p.ldra + n) : zero4;
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int n = n0 + wn * WN + i * 16 + 4 * kq;
                if (n >= p.N) continue;
                f32x4 v = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                f32x4 r4 = zero4;
                if (resg) { r4[0] = (float)res_raw[i][j][0]; r4[1] = (float)res_raw[i][j][1]; r4[2] = (float)res_raw[i][j][2]; r4[3] = (float)res_raw[i][j][3]; }
                conv_emit<T>(p, rows[j], n, v, bias4[i], ra4[i], r4, dseed);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int mt_ = m0 + wm * WM + j * 32 + fr;
            if (mt_ >= p.M) continue;
            int m, b;
            conv_out_row<MODE>(p, mt_, m, b);
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                f32x4 b4[4], r4[4], a4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {          // the 4 channel groups of this 32x32 tile: loads first
                    const int n = n0 + wn * WN + i * 32 + 8 * g + 4 * fh;
                    b4[g] = (p.bias && n < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + n) : zero4;
                    r4[g] = (resg && n < p.N) ? load4<T>(resg + (int64_t)m * ldaux + n) : zero4;
                    a4[g] = (rag && n < p.N) ? load4<T>(rag + (int64_t)b * p.ldra + n) : zero4;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + wn * WN + i * 32 + 8 * g + 4 * fh;
                    if (n >= p.N) continue;
                    f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    conv_emit<T>(p, m, n, v, b4[g], a4[g], r4[g], dseed);
                }
            }
        }
    }
    }   // !SPLITK
#endif
}


// split-K finish: y[m, n..n+3] = epilogue(sum_s ws[s][m][n..n+3]) - the run-time epilogue form (conv_value), one thread per
// pixel and 4 channels; MODE 3 (parity-class launches) is never split, so tile pixel == y row
template <typename T>
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const ConvP p) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n4 = p.N >> 2;
    if (idx >= (int64_t)p.M * n4) return;
    const int m = (int)(idx / n4), n = (int)(idx - (int64_t)m * n4) << 2;
    const int64_t slab = (int64_t)p.M * p.N;
    const float* src = p.ws + (int64_t)m * p.N + n;
    f32x4 v = *reinterpret_cast<const f32x4*>(src);
    for (int s = 1; s < p.splits; ++s) v += *reinterpret_cast<const f32x4*>(src + s * slab);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const T* resg = reinterpret_cast<const T*>(p.dact_u ? p.dact_u : p.residual);
    const int64_t ldaux = p.dact_u ? p.lddact : p.ldres;
    const T* rag = reinterpret_cast<const T*>(p.rowadd);
    const int b = rag ? m / (p.Ho * p.Wo) : 0;
    const f32x4 b4 = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : zero4;
    const f32x4 r4 = resg ? load4<T>(resg + (int64_t)m * ldaux + n) : zero4;
    const f32x4 a4 = rag ? load4<T>(rag + (int64_t)b * p.ldra + n) : zero4;
    const uint64_t dseed = p.drop_seed + ((p.drop_thresh && p.seed_dev) ? *p.seed_dev : 0ull);
    conv_emit<T>(p, m, n, v, b4, a4, r4, dseed);
}

template <typename T, int BM, int BN>
int launch_conv(const ConvP& p, hipStream_t stream) {
    const size_t lds = (size_t)2 * (BM + BN) * 128;
    ConvP q = p;
    q.mtiles = (p.M + BM - 1) / BM;
    q.ntiles = (p.N + BN - 1) / BN;
    const int grid = q.mtiles * q.ntiles * (q.splits > 1 ? q.splits : 1);
    // algorithmic bytes: every input pixel, weight and output element once (+ the epilogue's residual / saved tensors)
    const double esz = (double)sizeof(T);
    const double abytes = ((double)p.B * p.Hi * p.Wi * p.Cin + (double)p.N * p.taps * p.Cin + (double)p.M * p.N * (1.0 + (p.residual || p.dact_u ? 1.0 : 0.0) + (p.preact ? 1.0 : 0.0))) * esz;
    ProfScope prof(p.transposed ? PROF_CONV_DGRAD : PROF_CONV_FWD, 2.0 * (double)p.M * (double)p.N * (double)p.taps * (double)p.Cin, stream, abytes);
    const ConvP fin = q;                               // (the finishing kernel runs the epilogue: it keeps the operands)
    if (q.splits > 1) { q.bias = nullptr; q.rowadd = nullptr; q.residual = nullptr; q.preact = nullptr; q.dact_u = nullptr; }
    const int mode = p.ntap > 0 ? 3 : (!p.fast ? 2 : (!p.transposed ? 0 : (p.stride == 1 ? 1 : 2)));
    if (q.splits > 1) {
        if (mode == 0) hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN, 0, true>), dim3(grid), dim3(256), lds, stream, q);
        else if (mode == 1) hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN, 1, true>), dim3(grid), dim3(256), lds, stream, q);
        else if (mode == 2) hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN, 2, true>), dim3(grid), dim3(256), lds, stream, q);
        else return set_error(PSG_ERR_ARG, "conv_gemm: parity-class launches are never split");
    } else if (mode == 3) {
        // (the 160-wide tiles are not built for the parity-class mode: its tap tables push the kernel past the scalar
        //  register file - one layer, the first downsample's data gradient, runs 128x128 instead)
        if constexpr (BN != 160) hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN, 3>), dim3(grid), dim3(256), lds, stream, q);
        else return set_error(PSG_ERR_ARG, "conv_gemm: no 160-wide parity-class kernel");
    } else if (mode == 0) hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN, 0>), dim3(grid), dim3(256), lds, stream, q);
    else if (mode == 1) hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN, 1>), dim3(grid), dim3(256), lds, stream, q);
    else hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN, 2>), dim3(grid), dim3(256), lds, stream, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "conv_gemm launch");
    if (q.splits > 1) {
        const int64_t n = (int64_t)q.M * (q.N >> 2);
        hipLaunchKernelGGL((conv_splitk_finish_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, fin);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "conv_splitk_finish launch");
    }
    return PSG_OK;
}

template <typename T, int BM, int BN>
int set_conv_attrs() {
    const int lds = 2 * (BM + BN) * 128;
    PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<T, BM, BN, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<T, BM, BN, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<T, BM, BN, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<T, BM, BN, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<T, BM, BN, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<T, BM, BN, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if constexpr (BN != 160)
        PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<T, BM, BN, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    return PSG_OK;
}

}  // namespace psg

