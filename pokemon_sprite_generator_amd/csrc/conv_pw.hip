// Persistent pointwise GEMM (every nn.Linear and 1x1 conv of the U-Net whose output is a whole number of 128 x 128 tiles;
// reference: unet.py:96,160-187,217-251) - the short-K layers, where a tile's K loop (K = 640: 10 steps) is no longer than what
// surrounds it.  tools/conv_timeline.py (round 3) on conv_gemm_kernel: 1.4 us until the first K slice has landed, 8.5 us of K
// loop, 2.1-2.6 us of epilogue whose stores ALL resident workgroups issue at the same moment (the two workgroups of a CU run in
// phase) - 36 % of a K = 640 tile with the matrix pipe idle, and with a second output tensor (the FFN's first GEMM) the
// store burst alone is as long as the K loop.
//
// Same 128 x 128 x 64 tile, same LDS-DMA double buffer, same MFMA stream and the same epilogue ARITHMETIC (conv_value_k: the
// results are those of conv_gemm_kernel bit for bit) - what changes is what a workgroup does around its K loop:
//   * persistent: gridDim = resident slots; a workgroup walks tiles b, b + G, b + 2G, ... (the XCD-contiguous raster of
//     conv_gemm_kernel, so one XCD's L2 still sees a contiguous window of tiles);
//   * the first TWO K slices of the next tile are requested before this tile's epilogue starts - slice 0 at the top of the last
//     K step (its buffer is free by then), slice 1 right after the barrier that ends the K loop - so the next K loop starts
//     without waiting for memory;
//   * that needs both tile buffers during the epilogue, so the epilogue no longer stages through them: each wave owns a
//     16-pixel-row staging strip (2.3 KB; 73 KB per workgroup, still two per CU) and flushes its sub-tile in four strips;
//   * `vmcnt` retires in order and counts stores: a wave that stores and then requests a tile cannot wait for the tile
//     without waiting for the stores.  Here every wait is a COUNT that leaves the younger requests in flight - the epilogue's
//     operand loads (bias, residual / saved derivative) are inline asm too, issued under the last K step, so hipcc (which
//     cannot see asm memory operations) inserts no drain of its own - and the stores of tile i are first waited for at the
//     end of K step 1 of tile i + 1, a K step after they were issued: the store burst drains under the next K loop.
#include "conv_gemm_kernel.h"

namespace psg {

__device__ __forceinline__ void pw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pw_load8(u32x2& r, const u32x4& rsrc, uint32_t voff) {
    asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(r) : "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void pw_load16(f32x4& r, const u32x4& rsrc, uint32_t voff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r) : "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void pw_load16u(u32x4& r, const u32x4& rsrc, uint32_t voff) {      // (the destination is written by the
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r) : "v"(voff), "s"(rsrc) : "memory");   //  load itself: no copy may sit between it and its wait)
}
template <typename V> __device__ __forceinline__ void pw_tie(V& r) { asm volatile("" : "+v"(r)); }

// EK: epilogue kind of conv_gemm_kernel.h.  HAS_AUX: a residual (added) or, for EK_DMUL, the saved derivative (multiplied).
// HAS_PRE: second output tensor (`preact`: the pre-activation, or with PSG_CONV_SAVE_DACT the epilogue's derivative).
// TR: the launch is a data gradient (transposed) - it only NAMES the instantiation, so that kernel traces and PMC summaries
// (tools/pmc_summary.py) can put a launch into the forward or the data-gradient family; the code is the same.
template <int EK, bool HAS_AUX, bool HAS_PRE, bool TR>
__global__ __launch_bounds__(256, 2) void conv_pw_kernel(const ConvP p, const int total_tiles, const uint32_t aux_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef bf16_t T;
    constexpr int BM = 128, BN = 128, WM = 64, WN = 64, NA = 4, NB = 4;
    constexpr int SLOT = (BM + BN) * 128;              // one K slice: W tile | X tile, 128-byte rows
    constexpr int PASS_BYTES = 32 * 128;               // rows per staging pass x 128 B
    constexpr int PITCH = WN * 2 + 16;                 // staging strip row (bf16): conflict-free ds_write_b64 / ds_read_b128
    constexpr int STRIP = 16 * PITCH;
    constexpr int NS = HAS_PRE ? 16 : 8;               // 16-byte store instructions per lane and tile
    constexpr int ND = 8;                              // LDS-DMA instructions per lane and K slice
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [2][SLOT] | [4 waves][STRIP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, wm = wave >> 1;
    const int l16 = lane & 15, kq = lane >> 4;
    const int sr = tid >> 3, sc = (tid & 7) ^ ((sr >> 1) & 7);
    const u32x4 xrs = make_rsrc(p.x, p.x_bytes), wrs = make_rsrc(p.w, p.w_bytes);
    const T* auxg = reinterpret_cast<const T*>(p.dact_u ? p.dact_u : p.residual);
    const int64_t ldaux = p.dact_u ? p.lddact : p.ldres;
    const u32x4 ars = make_rsrc(auxg, aux_bytes);
    const u32x4 brs = make_rsrc(p.bias, (uint32_t)p.N * 4u);
    typedef __attribute__((address_space(3))) char* lds_ptr_t;
    const uint32_t lds_wave = (uint32_t)(size_t)(lds_ptr_t)smem + (uint32_t)__builtin_amdgcn_readfirstlane(wave) * 1024u;
    char* strip = smem + 2 * SLOT + wave * STRIP;

    // virtual block id -> tile (conv_gemm_kernel's XCD-contiguous, 8-M-tile grouped raster over total_tiles "blocks")
    auto tile_of = [&](int v, int& m0, int& n0) {
        const int nb = total_tiles;
        const int q = nb >> 3, r = nb & 7, xcd = v & 7;
        const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        constexpr int GM = 8;
        const int per_group = GM * p.ntiles;
        const int g = lid / per_group, rem = lid - g * per_group;
        const int gm = min(GM, p.mtiles - g * GM);
        const int nt = rem / gm;
        m0 = (g * GM + (rem - nt * gm)) * BM; n0 = nt * BN;
    };
    struct Off { uint32_t w[4], x[4]; };
    auto offsets = [&](int m0, int n0, Off& o) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o.w[j] = (uint32_t)(((int64_t)(n0 + sr + 32 * j) * p.ldw + sc * 8) * 2);
            o.x[j] = (uint32_t)(((int64_t)(m0 + sr + 32 * j) * p.ldx + sc * 8) * 2);
        }
    };
    auto request = [&](const Off& o, int k, int slot) {           // K slice k of a tile -> LDS buffer `slot`
        const uint32_t wdst = lds_wave + (uint32_t)slot * SLOT, xdst = wdst + BN * 128;
        const uint32_t kb = (uint32_t)__builtin_amdgcn_readfirstlane(k) * 128u;
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16s(wrs, wdst + j * PASS_BYTES, o.w[j], kb);
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16s(xrs, xdst + j * PASS_BYTES, o.x[j], kb);
    };

    f32x4 acc[NA][NB];
    const int rd_w = (wn * WN + l16) * 128, rd_x = BN * 128 + (wm * WM + l16) * 128;
    const int swz = (l16 >> 1) & 7;
    constexpr int FSTEP = 16 * 128;
    auto compute = [&](int slot) {
        const char* tb = smem + slot * SLOT;
        uint4 wf[2][NA], xf[2][NB];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int co = ((4 * s2 + kq) ^ swz) << 4;
#pragma unroll
            for (int i = 0; i < NA; ++i) wf[s2][i] = *reinterpret_cast<const uint4*>(tb + rd_w + i * FSTEP + co);
#pragma unroll
            for (int j = 0; j < NB; ++j) xf[s2][j] = *reinterpret_cast<const uint4*>(tb + rd_x + j * FSTEP + co);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = mma16(wf[s2][i], xf[s2][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
    };

    // the dropout seed word (psg_set_seed_source) is read ONCE, into scalar registers, before anything is in flight: as an
    // ordinary load inside the tile loop hipcc would drain vmcnt in front of its use - every tile's epilogue would wait for the
    // next tile's K slices
    uint64_t dseed = p.drop_seed;
    if (EK == EK_DROP || EK == EK_GELU_DROP) {
        if (p.drop_thresh && p.seed_dev) {
            const uint64_t w = *p.seed_dev;
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)w), hi = __builtin_amdgcn_readfirstlane((uint32_t)(w >> 32));
            dseed += ((uint64_t)hi << 32) | lo;
        }
    }
    const int KT = p.KT, G = gridDim.x;
    int v = blockIdx.x;
    int m0, n0;
    tile_of(v, m0, n0);
    Off cur, nxt;
    offsets(m0, n0, cur);
    request(cur, 0, 0);
    request(cur, 1, 1);
    int par = 0;                                        // LDS buffer of this tile's K slice 0
    bool first = true;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (;;) {
        const bool more = v + G < total_tiles;
        int m1 = 0, n1 = 0;
        if (more) { tile_of(v + G, m1, n1); offsets(m1, n1, nxt); }
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = zero4;
        // slices 0 and 1 were requested before the previous tile's epilogue (or above); younger than slice 0 in this wave's
        // queue: slice 1 (ND) and the previous tile's stores (NS)
        // (the first tile has no stores behind its slices)
        if (first) wait_vmcnt<ND>(); else wait_vmcnt<ND + NS>();
        pw_barrier();
        compute(par);
        if (first) wait_vmcnt<0>(); else wait_vmcnt<NS>();       // slice 1 (the stores behind it stay in flight)
        first = false;
        pw_barrier();
        for (int k = 1; k + 1 < KT; ++k) {
            request(cur, k + 1, (par + k + 1) & 1);
            compute((par + k) & 1);
            wait_vmcnt<0>();                            // (k = 1: also the previous tile's stores, a K step after their issue)
            pw_barrier();
        }
        // ---- last K step: next tile's slice 0, then this tile's epilogue operands, under the MFMAs ----
        const int endpar = (par + KT) & 1;              // buffer after the last one = next tile's slice 0
        if (more) request(nxt, 0, endpar);
        f32x4 bias4[NA];
        // aux operand (residual / saved derivative) of the wave's 64 x 64 sub-tile, ROW-major: 16 bytes per lane, 8 whole
        // 128-byte rows per instruction (as 8-byte pieces in the accumulator layout - 16 rows x 32 bytes per instruction -
        // the same bytes took twice the load instructions and four times the line requests, under the K loop's own DMA
        // stream: every layer with a residual got SLOWER than on the per-tile kernel); it is turned into the accumulator
        // layout strip by strip through the wave's staging strip
        u32x4 auxr[8];
        if (p.bias) {
#pragma unroll
            for (int i = 0; i < NA; ++i) pw_load16(bias4[i], brs, (uint32_t)(n0 + wn * WN + i * 16 + 4 * kq) * 4u);
        }
        if (HAS_AUX) {
            const uint32_t rowb = (uint32_t)(((int64_t)(m0 + wm * WM + (lane >> 3)) * ldaux + n0 + wn * WN + (lane & 7) * 8) * 2);
            const uint32_t step8 = (uint32_t)(ldaux * 16);                 // 8 rows further
#pragma unroll
            for (int it = 0; it < 8; ++it) pw_load16u(auxr[it], ars, rowb + (uint32_t)it * step8);
        }
        compute((par + KT - 1) & 1);
        pw_barrier();                                   // every wave is done with the last slice's buffer
        if (more) request(nxt, 1, endpar ^ 1);
        // the epilogue's operands are older than that request (and than nothing else that matters): leave ND in flight
        if (more) wait_vmcnt<ND>(); else wait_vmcnt<0>();
        if (p.bias) {
#pragma unroll
            for (int i = 0; i < NA; ++i) pw_tie(bias4[i]);
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) bias4[i] = zero4;
        }
        if (HAS_AUX) {
#pragma unroll
            for (int it = 0; it < 8; ++it) pw_tie(auxr[it]);
        }
        // ---- epilogue: one 16-pixel strip of the wave's 64 x 64 sub-tile at a time ----
        T* yg = reinterpret_cast<T*>(p.y);
        T* pg = reinterpret_cast<T*>(p.preact);
        // rows -> accumulator cells of strip j through the staging strip (LDS operations of a wave execute in order: the
        // writes may follow the previous strip's row reads without a wait, the cell reads follow the writes)
        u32x2 auxc[2][NA];
        auto aux_cells = [&](int j, u32x2 (&c)[NA]) {
            *reinterpret_cast<u32x4*>(strip + (lane >> 3) * PITCH + (lane & 7) * 16) = auxr[2 * j];
            *reinterpret_cast<u32x4*>(strip + (8 + (lane >> 3)) * PITCH + (lane & 7) * 16) = auxr[2 * j + 1];
#pragma unroll
            for (int i = 0; i < NA; ++i) c[i] = *reinterpret_cast<const u32x2*>(strip + l16 * PITCH + (i * 16 + 4 * kq) * 2);
        };
        if (HAS_AUX) aux_cells(0, auxc[0]);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int m = m0 + wm * WM + j * 16 + l16;
            f32x4 val[NA], pre[NA];
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int n = n0 + wn * WN + i * 16 + 4 * kq;
                f32x4 a4 = zero4;
                if (HAS_AUX) {
                    const uint32_t lo = auxc[j & 1][i].x, hi = auxc[j & 1][i].y;
                    a4[0] = __uint_as_float(lo << 16); a4[1] = __uint_as_float(lo & 0xFFFF0000u);
                    a4[2] = __uint_as_float(hi << 16); a4[3] = __uint_as_float(hi & 0xFFFF0000u);
                }
                val[i] = acc[i][j];
                conv_value_k<T, EK>(p, m, n, val[i], pre[i], bias4[i], zero4, a4, dseed);
            }
            auto flush = [&](const f32x4 (&o)[NA], T* dst, int64_t ldd) {
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    bf16x4 ob = {(bf16_t)o[i][0], (bf16_t)o[i][1], (bf16_t)o[i][2], (bf16_t)o[i][3]};
                    *reinterpret_cast<bf16x4*>(strip + l16 * PITCH + (i * 16 + 4 * kq) * 2) = ob;
                }
                T* base = dst + (int64_t)(m0 + wm * WM + j * 16) * ldd + (n0 + wn * WN);
                uint4 c0 = *reinterpret_cast<const uint4*>(strip + (lane >> 3) * PITCH + (lane & 7) * 16);
                uint4 c1 = *reinterpret_cast<const uint4*>(strip + (8 + (lane >> 3)) * PITCH + (lane & 7) * 16);
                if (HAS_AUX && j + 1 < NB) aux_cells(j + 1, auxc[(j + 1) & 1]);      // (behind the row reads, ahead of the stores)
                *reinterpret_cast<uint4*>(base + (int64_t)(lane >> 3) * ldd + (lane & 7) * 8) = c0;
                *reinterpret_cast<uint4*>(base + (int64_t)(8 + (lane >> 3)) * ldd + (lane & 7) * 8) = c1;
            };
            if (HAS_PRE) flush(pre, pg, p.ldpre);
            flush(val, yg, p.ldy);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!more) break;
        v += G; m0 = m1; n0 = n1; cur = nxt; par = endpar;
    }
#endif
}

// ---- host side ----------------------------------------------------------------------------------------------------------
static int g_pw_on = -1;                                 // PSG_CONV_PW=0 / psg_conv_set_pw(0): every pointwise layer on conv_gemm_kernel (A/B, tests)
static int64_t g_pw_launches = 0;
static int pw_enabled() {
    if (g_pw_on < 0) { const char* e = getenv("PSG_CONV_PW"); g_pw_on = (e && atoi(e) == 0) ? 0 : 1; }
    return g_pw_on;
}

static int pw_kind(const ConvP& p) {
    if (p.epi_generic) return -1;
    if (p.dact_u) return ((p.flags & PSG_CONV_DACT_MUL) && !p.drop_thresh) ? EK_DMUL : -1;
    if (p.act == PSG_ACT_NONE) return p.drop_thresh ? EK_DROP : EK_PLAIN;
    if (p.act == PSG_ACT_GELU) return p.drop_thresh ? EK_GELU_DROP : EK_GELU;
    return -1;
}

// Whole 128 x 128 tiles, K a whole number (>= 3) of 64-channel slices, bf16, the staged (16-byte row) store conditions, an
// epilogue kind with a branch-free copy, no per-sample add, and enough tiles that a resident workgroup gets at least two.
bool conv_pw_applicable(const ConvP& p, int dtype) {
    if (!pw_enabled() || dtype != PSG_BF16 || p.taps != 1 || p.stride != 1 || !p.fast || p.ntap > 0 || p.splits > 1) return false;
    if (!p.epi_lds || p.rowadd || p.M % 128 || p.N % 128 || p.KT < 3 || p.Cin % 64) return false;
    if (pw_kind(p) < 0) return false;
    if (p.dact_u && p.residual) return false;
    const int64_t tiles = (int64_t)(p.M / 128) * (p.N / 128);
    if (tiles < 3 * (int64_t)avail_cus_for((double)tiles / 512.0)) return false;   // (1.5 tiles per resident slot: most workgroups get a second tile)
    const void* aux = p.dact_u ? p.dact_u : p.residual;
    if (aux) {
        const int64_t ld = p.dact_u ? p.lddact : p.ldres;
        if (((int64_t)p.M - 1) * ld * 2 + (int64_t)p.N * 2 >= 0x7FFFFFF0ll || !aligned16(aux)) return false;
    }
    if (p.preact && (p.ldpre % 8 || !aligned16(p.preact))) return false;
    if (p.bias && !aligned16(p.bias)) return false;
    return true;
}

static constexpr int PW_LDS = 2 * 256 * 128 + 4 * 16 * (64 * 2 + 16);

#define PSG_PW_FOR_ALL(X)                                                                                  \
    X(EK_PLAIN, false, false) X(EK_PLAIN, true, false) X(EK_DROP, false, false) X(EK_DROP, true, false)   \
    X(EK_GELU, false, false) X(EK_GELU, false, true) X(EK_GELU_DROP, false, false) X(EK_GELU_DROP, false, true) \
    X(EK_DMUL, true, false)

int conv_pw_set_attrs() {
#define X(EK, AUX, PRE)                                                                                                                                   \
    PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pw_kernel<EK, AUX, PRE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS)); \
    PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pw_kernel<EK, AUX, PRE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS));
    PSG_PW_FOR_ALL(X)
#undef X
    return PSG_OK;
}

int launch_conv_pw(const ConvP& p0, hipStream_t stream) {
    ConvP p = p0;
    p.mtiles = p.M / 128; p.ntiles = p.N / 128;
    const int total = p.mtiles * p.ntiles;
    int grid = 2 * avail_cus_for((double)total / 512.0);
    if (grid > total) grid = total;
    grid &= ~7;                                          // (a virtual block keeps its XCD: v and v + G agree mod 8)
    const int ek = pw_kind(p);
    const bool aux = (p.dact_u || p.residual), pre = p.preact != nullptr;
    const int64_t ldaux = p.dact_u ? p.lddact : p.ldres;
    const uint32_t aux_bytes = aux ? (uint32_t)(((int64_t)p.M - 1) * ldaux * 2 + (int64_t)p.N * 2) : 0u;
    const double abytes = ((double)p.M * p.Cin + (double)p.N * p.Cin + (double)p.M * p.N * (1.0 + (aux ? 1.0 : 0.0) + (pre ? 1.0 : 0.0))) * 2.0;
    ProfScope prof(p.transposed ? PROF_CONV_DGRAD : PROF_CONV_FWD, 2.0 * (double)p.M * (double)p.N * (double)p.Cin, stream, abytes);
    bool done = false;
#define X(EK, AUX, PRE)                                                                                                    \
    if (!done && ek == EK && aux == AUX && pre == PRE) {                                                                   \
        if (p.transposed) hipLaunchKernelGGL((conv_pw_kernel<EK, AUX, PRE, true>), dim3(grid), dim3(256), PW_LDS, stream, p, total, aux_bytes); \
        else hipLaunchKernelGGL((conv_pw_kernel<EK, AUX, PRE, false>), dim3(grid), dim3(256), PW_LDS, stream, p, total, aux_bytes);            \
        done = true;                                                                                                       \
    }
    PSG_PW_FOR_ALL(X)
#undef X
    if (!done) return -1;                                // (combination without an instantiation: the caller falls back)
    ++g_pw_launches;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "conv_pw launch");
    return PSG_OK;
}

}  // namespace psg

extern "C" {
int psg_conv_set_pw(int on) { psg::g_pw_on = on ? 1 : 0; return PSG_OK; }
int64_t psg_conv_pw_launches(void) { return psg::g_pw_launches; }
}
