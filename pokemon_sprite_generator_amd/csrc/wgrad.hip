// Weight gradient of conv / linear layers on the gfx950 matrix cores.
//
//   dW[co, q] = sum_m dY[m, co] * X[pix(m, tap(q)), ci(q)],   q = tap*Cin + ci
//
// Both operands are stored with the REDUCTION index (pixel m) as the slow
// dimension (channels-last), i.e. this is a "TN" GEMM.  Tiles [64 pixels][128 ch]
// of dY and of the gathered X go global -> LDS by LDS-DMA (inline-asm
// buffer_load ... lds, hand-placed waits) exactly as they lie in memory; the MFMA
// fragments (which need 8 consecutive reduction indices per lane) are produced by
// the hardware transposing LDS read ds_read_b64_tr_b16 (bf16: 256-byte rows, chunk
// XOR ((row&3)<<2)|((row>>2)&3) applied on the DMA source side makes the reads of
// the 16x16x32 operand conflict-free) or by plain ds_read_b32 (fp32
// v_mfma_f32_32x32x2_f32 takes one value per lane).  Output tile 128(co) x 128(q)
// per 256-thread workgroup, columns tiled over the flat q axis (so Cin=320 wastes
// nothing), split-K over pixels chosen by a makespan model, fp32 partial slabs and
// a deterministic fixed-order sum that writes the gradient in the parameter's own
// memory order (OHWI: no permute; one split: tiles written directly); the bias
// gradient rides along as one extra MFMA per dY fragment against a ones operand.
#include "psg_common.h"
#include <type_traits>

namespace psg {

struct WgP {
    const void* x; const void* dy; float* ws;
    float* bws;                                    // bias-gradient partials [splits][Cout] (or dbias itself), NULL = none
    float scale;                                   // constant factor folded into every partial
    int64_t ldx, lddy;
    int B, Hi, Wi, Cin, Ho, Wo, Cout, ks, stride, pad;
    int M, Q, taps;
    int rtiles, qtiles, splits, steps_per_split;   // steps of BKP pixels
    int BR;                                        // output rows (co) per tile: 128, or 160 = exact tiling of Cout = 320
    int wide;                                      // 1: wgrad_wide_kernel (128 x 256 tile, 32-pixel K steps)
    float inv_HoWo, inv_Wo;
    uint32_t x_bytes, dy_bytes;
};

template <typename T> struct WgCfg;
template <> struct WgCfg<bf16_t> { static constexpr int BKP = 64; static constexpr int ROWB = 256; };  // bytes per LDS row
template <> struct WgCfg<float> { static constexpr int BKP = 32; static constexpr int ROWB = 512; };

// GEOM selects how the gathered-X row addresses are produced:
//   0 general (stride 2, any padding): per-row (b, ho, wo) odometer
//   1 "same" 3x3 (stride 1, Hi==Ho, Wi==Wo): the source pixel of tap (kh,kw) is m + const, so ONE address register
//     steps linearly and the four staging passes differ only by a scalar offset; the image-border test needs just
//     (m mod HoWo, m mod Wo) trackers per pass
//   2 pointwise (1x1 conv / Linear): as 1 with no border at all
// The per-lane address VALU work is what bounds this kernel (it competes with the MFMAs for issue slots).
// BR = 160 (bf16, GEOM 1/2, Cout % 160 == 0): the dY tile is [64 pixels][160 co] with 320-byte rows - 80 dwords, so four
// consecutive rows already start 16 banks apart and the chunk key shrinks to ((row >> 3) & 1) << 1; each wave owns
// 80 co x 64 q = 5 x 4 MFMA tiles.  Three 128-row tiles over Cout = 320 leave the third half empty (18 % of the launch).
template <typename T, int GEOM, int BR = 128>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgP p) {
#if defined(__HIP_DEVICE_COMPILE__)      // the LDS-DMA builtin exists only in the device pass
    constexpr int CH = Elem<T>::CH;
    constexpr int BKP = WgCfg<T>::BKP;
    constexpr int ROWB = WgCfg<T>::ROWB;
    constexpr int CPR = 128 / CH;            // 16-byte chunks per 128-channel row
    constexpr int RPP = 256 / CPR;           // rows per staging pass
    constexpr int NPASS = BKP / RPP;         // = 4
    constexpr int TILE_BYTES = BKP * ROWB;                         // the gathered-X tile (128 q)
    constexpr int AROWB = BR * (int)sizeof(T);                     // dY tile row
    constexpr int ATILE_BYTES = BKP * AROWB;
    constexpr int STAGE_BYTES = ATILE_BYTES + TILE_BYTES;
    constexpr int ACH = BR / CH;                                   // 16-byte chunks per dY tile row
    constexpr int NPA = BKP * ACH / 256;                           // dY staging passes (4; 5 for BR = 160)
    static_assert(BR == 128 || (BR == 160 && sizeof(T) == 2 && GEOM != 0), "160-row tiles: bf16, stride-1 geometries");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [buf][A tile | B tile]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave & 1, wc = wave >> 1;

    // block -> (split, col tile, row tile), split SLOWEST: workgroups resident together then sweep the same pixel
    // range (every output tile needs the same dY / X rows of that range), so operands are served from L2 instead
    // of each tile streaming its own pixels from HBM.  XCD-aware: each XCD gets a contiguous range of logical ids.
    int lid;
    {
        const int nb = gridDim.x, b0 = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = b0 & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b0 >> 3);
    }
    const int tiles = p.rtiles * p.qtiles;
    const int split = lid / tiles;
    const int tix = lid - split * tiles;
    const int rt = tix % p.rtiles, qt = tix / p.rtiles;
    const int co0 = rt * BR, q0 = qt * 128;

    // Tiles go global -> LDS by LDS-DMA (wave-uniform base + lane*16: whole 256/512-byte rows, unpadded).  For bf16
    // the transposed fragment reads (4 consecutive pixel rows x 64-byte column windows per half-wave) are made
    // conflict-free by XOR-ing the 16-byte chunk index with ((row & 3) << 2) | ((row >> 2) & 3) - applied on the SOURCE
    // side: the lane at physical chunk (tid % CPR) of row sr fetches logical chunk sc.  (Passes step 16 rows: same key.)
    const int sr = tid / CPR;
    const int sc = sizeof(T) == 2 ? ((tid % CPR) ^ (((sr & 3) << 2) | ((sr >> 2) & 3))) : (tid % CPR);

    // A operand (dY): column chunk -> co
    const int a_co = co0 + sc * CH;
    const bool a_ok = a_co < p.Cout;
    // B operand (X gathered): column chunk -> (tap, ci)
    const int b_q = q0 + sc * CH;
    const bool b_ok = b_q < p.Q;
    const int b_tap = b_ok ? b_q / p.Cin : 0;
    const int b_ci = b_q - b_tap * p.Cin;
    const int b_kh = b_tap / p.ks, b_kw = b_tap - b_kh * p.ks;
    const int HoWo = p.Ho * p.Wo;

    const int step0 = split * p.steps_per_split;
    const int total_steps = (p.M + BKP - 1) / BKP;
    int nsteps = total_steps - step0;
    if (nsteps > p.steps_per_split) nsteps = p.steps_per_split;

    constexpr uint32_t OOB = 0x80000000u;          // extents are < 2 GiB (checked on the host)
    constexpr int ESZ = (int)sizeof(T);
    const u32x4 xrs = make_rsrc(p.x, p.x_bytes);
    const u32x4 yrs = make_rsrc(p.dy, p.dy_bytes);
    // Per-row "odometer": (m, ho, wo, byte offsets) of each staged pixel row advance by BKP pixels per K step
    // with adds / compares / selects only (no integer multiply or divide in the loop).
    const int ldxB = (int)p.ldx * ESZ;
    const int od_dB = BKP / HoWo, od_r1 = BKP - od_dB * HoWo, od_dH = od_r1 / p.Wo, od_dW = od_r1 - od_dH * p.Wo;   // uniform
    const int SW = p.stride * ldxB, SH = p.stride * p.Wi * ldxB, SB = p.Hi * p.Wi * ldxB;
    const int od_A0 = od_dW * SW + od_dH * SH + od_dB * SB, od_K1 = SH - p.Wo * SW, od_K2 = SB - p.Ho * SH;
    const int a_step = BKP * (int)p.lddy * ESZ;
    const int b_const = ((b_kh - p.pad) * p.Wi + (b_kw - p.pad)) * ldxB + b_ci * ESZ;
    // this lane's filter tap is fixed, so "source pixel inside the image" is a fixed range of output rows / columns:
    // ho*stride - pad + kh in [0, Hi)  <=>  ho in [ho_lo, ho_lo + h_rng]  (one subtract + one unsigned compare)
    const int ho_lo = max(0, (p.pad - b_kh + p.stride - 1) / p.stride), wo_lo = max(0, (p.pad - b_kw + p.stride - 1) / p.stride);
    const int ho_hi = min(p.Ho - 1, (p.Hi - 1 + p.pad - b_kh) / p.stride), wo_hi = min(p.Wo - 1, (p.Wi - 1 + p.pad - b_kw) / p.stride);
    const bool b_in = b_ok && ho_hi >= ho_lo && wo_hi >= wo_lo;
    const unsigned h_rng = (unsigned)(ho_hi - ho_lo), w_rng = (unsigned)(wo_hi - wo_lo);
    int r_m[NPASS], r_ho[NPASS], r_wo[NPASS], r_pix[NPASS], r_a[NPASS];
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
        const int m = step0 * BKP + sr + RPP * j;
        const int b = m / HoWo, rm = m - b * HoWo;
        r_m[j] = m; r_ho[j] = rm / p.Wo; r_wo[j] = rm - r_ho[j] * p.Wo;
        r_pix[j] = b * SB + r_ho[j] * SH + r_wo[j] * SW;
        r_a[j] = (m * (int)p.lddy + a_co) * ESZ;
    }
    typedef __attribute__((address_space(3))) char* lds_ptr_t;
    // this wave's 1 KiB slot inside a pass (LDS byte address, wave-uniform).  The DMA is issued from inline asm
    // (psg_common.h: lds_dma16) so hipcc does not drain it before the MFMA phase; waits are placed by hand.
    const uint32_t lds_wave = (uint32_t)(size_t)(lds_ptr_t)smem + (uint32_t)__builtin_amdgcn_readfirstlane(wave) * 1024u;
    constexpr int PASS_BYTES = RPP * ROWB;                                  // = 4096
    // GEOM 1/2 state: pass 0's pixel and its two linear byte offsets; pass j adds the uniform j*RPP rows
    int g_m = step0 * BKP + sr;
    int g_a = (g_m * (int)p.lddy + a_co) * ESZ;
    // hardware offsets are unsigned: the (possibly negative) tap shift is folded into a descriptor whose base sits
    // pad rows + pad pixels BEFORE x, so the per-lane offset is >= 0; taps that would read below x are never valid
    const int g_shift = (p.pad * p.Wi + p.pad) * ldxB;
    const u32x4 xrs_g = make_rsrc(reinterpret_cast<const char*>(p.x) - g_shift, p.x_bytes + (uint32_t)g_shift);
    int g_b = g_m * ldxB + (b_kh * p.Wi + b_kw) * ldxB + b_ci * ESZ;
    // BR = 160: lane e = 256 j + tid of pass j sits at (row e / 20, physical chunk e % 20) of the dY image and fetches the
    // logical chunk (physical ^ key(row)); rows >= M lie beyond the descriptor's extent and read as zeros (no per-row test)
    uint32_t a_voff[BR == 160 ? NPA : 1];
    if constexpr (BR == 160) {
#pragma unroll
        for (int j = 0; j < NPA; ++j) {
            const int e = j * 256 + tid, row = e / ACH, pc = e - row * ACH;
            const int lc = pc ^ (((row >> 3) & 1) << 1);
            a_voff[j] = (uint32_t)(((step0 * BKP + row) * (int)p.lddy + co0 + lc * CH) * ESZ);
        }
    }
    const int a_pass = RPP * (int)p.lddy * ESZ, b_pass = RPP * ldxB;       // uniform
    const int st_rm = BKP % HoWo, st_wo = BKP % p.Wo;
    const int rm_lo = ho_lo * p.Wo;
    const unsigned rm_rng = (unsigned)((ho_hi + 1) * p.Wo - 1 - rm_lo);
    int g_rm[NPASS], g_wo[NPASS];
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
        g_rm[j] = (g_m + RPP * j) % HoWo;
        g_wo[j] = g_rm[j] % p.Wo;
    }
    auto load_tiles = [&](int buf) {   // DMA the rows at the current odometer state into LDS buffer buf, then advance
        const uint32_t adst = lds_wave + (uint32_t)buf * STAGE_BYTES;
        const uint32_t bdst = adst + ATILE_BYTES;
        if constexpr (GEOM == 0) {
#pragma unroll
            for (int j = 0; j < NPASS; ++j) {
                const bool m_ok = r_m[j] < p.M;
                const uint32_t aoff = (m_ok && a_ok) ? (uint32_t)r_a[j] : OOB;
                const bool ok = m_ok && b_in && (unsigned)(r_ho[j] - ho_lo) <= h_rng && (unsigned)(r_wo[j] - wo_lo) <= w_rng;
                const uint32_t boff = ok ? (uint32_t)(r_pix[j] + b_const) : OOB;
                lds_dma16(yrs, adst + j * PASS_BYTES, aoff);
                lds_dma16(xrs, bdst + j * PASS_BYTES, boff);
                // advance by BKP pixels
                r_m[j] += BKP; r_a[j] += a_step;
                int wo = r_wo[j] + od_dW;
                const bool c1 = wo >= p.Wo;
                wo -= c1 ? p.Wo : 0;
                int ho = r_ho[j] + od_dH + (c1 ? 1 : 0);
                const bool c2 = ho >= p.Ho;
                ho -= c2 ? p.Ho : 0;
                r_wo[j] = wo; r_ho[j] = ho;
                r_pix[j] += od_A0 + (c1 ? od_K1 : 0) + (c2 ? od_K2 : 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NPASS; ++j) {
                // (validity without compares: sign bits of the differences OR-ed into bit 31 of the offset - wgrad_pipe_kernel)
                const int m_bad = (p.M - RPP * j - 1) - g_m;
                const uint32_t aoff = (uint32_t)g_a | ((uint32_t)(m_bad | (a_ok ? 0 : -1)) & 0x80000000u);
                int bad = m_bad | (b_in ? 0 : -1);
                if constexpr (GEOM == 1) {
                    bad |= (g_rm[j] - rm_lo) | (rm_lo + (int)rm_rng - g_rm[j]) | (g_wo[j] - wo_lo) | (wo_lo + (int)w_rng - g_wo[j]);
                    const int t = g_rm[j] + (st_rm - HoWo), u = g_wo[j] + (st_wo - p.Wo);
                    g_rm[j] = (int)min((unsigned)t, (unsigned)(t + HoWo));
                    g_wo[j] = (int)min((unsigned)u, (unsigned)(u + p.Wo));
                }
                const uint32_t boff = (uint32_t)g_b | ((uint32_t)bad & 0x80000000u);
                if constexpr (BR == 128) lds_dma16s(yrs, adst + j * PASS_BYTES, aoff, (uint32_t)(j * a_pass));
                lds_dma16s(xrs_g, bdst + j * PASS_BYTES, boff, (uint32_t)(j * b_pass));
            }
            if constexpr (BR == 160) {
#pragma unroll
                for (int j = 0; j < NPA; ++j) {
                    lds_dma16(yrs, adst + j * PASS_BYTES, a_voff[j]);
                    a_voff[j] += (uint32_t)a_step;
                }
            }
            g_m += BKP; g_a += a_step; g_b += BKP * ldxB;
        }
    };

    // bf16: 4x4 tiles of v_mfma_f32_16x16x32_bf16 per wave (higher sustained clock than 32x32x16 at the same LDS
    // traffic); fp32: 2x2 tiles of v_mfma_f32_32x32x2_f32
    constexpr bool FT16 = sizeof(T) == 2;
    constexpr int NA = FT16 ? 4 : 2, AE = FT16 ? 4 : 16;
    constexpr int NR = FT16 ? BR / 32 : 2;            // row (co) tiles per wave: 4, or 5 for BR = 160
    typedef float AccT __attribute__((ext_vector_type(AE)));
    AccT acc[NR][NA];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int j = 0; j < NA; ++j)
#pragma unroll
            for (int e = 0; e < AE; ++e) acc[i][j][e] = 0.f;
    // bias gradient = column sums of the dY tile: the waves of tile column 0 multiply their dY fragments with a
    // ones operand (every accumulator column then holds sum_k dY[k][co]); wave-uniform condition
    const bool do_bias = p.bws != nullptr && qt == 0 && wc == 0;
    AccT accb[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int e = 0; e < AE; ++e) accb[i][e] = 0.f;

    if (nsteps > 0) load_tiles(0);
    wait_vmcnt<0>();
    __syncthreads();                       // tile 0 has landed for every wave

    for (int st = 0; st < nsteps; ++st) {
        const int buf = st & 1;
        if (st + 1 < nsteps) load_tiles(buf ^ 1);      // buffer buf^1 was last read in step st-1 (barrier below)
        const char* ab = smem + buf * STAGE_BYTES;
        const char* bb = ab + ATILE_BYTES;
        if constexpr (FT16) {
            // Transposed fragment reads.  Lane -> (16-lane group g, q4 = row of the 4x16 block, p4 = 4-column piece):
            // group g of k-step s2 reads pixel rows 32*s2 + 8g + {0..3} and + {4..7}, columns tile + 4*p4..+3, and
            // lane i16 receives column i16 of those rows: element e <-> pixel 32*s2 + 8g + e, the 16x16x32 operand order.
            // A half-wave's two blocks sit 8 rows apart in the same columns: conflict-free on this swizzled image.
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
            const int key0 = (q4 << 2) | ((2 * g) & 3), key1 = (q4 << 2) | ((2 * g + 1) & 3);   // swizzle keys of the 2 rows
            const int rowb = (8 * g + q4) * ROWB + 8 * (p4 & 1);
            const int rowa = (8 * g + q4) * AROWB + 8 * (p4 & 1);
            const int chA = wr * (BR / 16) + (p4 >> 1), chB = wc * 8 + (p4 >> 1);      // 16-byte chunk of tile 0 (+2 per tile)
            // chunk keys of the dY image: the 256-byte-row swizzle, or for 320-byte rows ((row >> 3) & 1) << 1 = (g & 1) << 1
            const int ka0 = BR == 160 ? ((g & 1) << 1) : key0, ka1 = BR == 160 ? ((g & 1) << 1) : key1;
            bf16x8 af[2][NR], bf[2][4];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const char* ra = ab + s2 * 32 * AROWB + rowa;
                const char* rb = bb + s2 * 32 * ROWB + rowb;
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ra + (((chA + 2 * i) ^ ka0) << 4)));
                    s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ra + 4 * AROWB + (((chA + 2 * i) ^ ka1) << 4)));
                    s16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    af[s2][i] = *reinterpret_cast<bf16x8*>(&av);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(rb + (((chB + 2 * i) ^ key0) << 4)));
                    s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(rb + 4 * ROWB + (((chB + 2 * i) ^ key1) << 4)));
                    s16x8 bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                    bf[s2][i] = *reinterpret_cast<bf16x8*>(&bv);
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // reads above, MFMA stream below
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < NR; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s2][i], bf[s2][j], acc[i][j], 0, 0, 0);
            if (do_bias) {
                const bf16_t one = (bf16_t)1.0f;
                const bf16x8 ones = {one, one, one, one, one, one, one, one};
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int i = 0; i < NR; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s2][i], ones, accb[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
            const int fr = lane & 31, fh = lane >> 5;
#pragma unroll 4
            for (int kk = 0; kk < BKP / 2; ++kk) {
                float af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = *reinterpret_cast<const float*>(ab + (kk * 2 + fh) * ROWB + (wr * 64 + i * 32 + fr) * 4);
                    bf[i] = *reinterpret_cast<const float*>(bb + (kk * 2 + fh) * ROWB + (wc * 64 + i * 32 + fr) * 4);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
                if (do_bias) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], 1.0f, accb[i], 0, 0, 0);
                }
            }
        }
        wait_vmcnt<0>();                   // AFTER the MFMA phase: the DMA of the next tile overlapped it
        __syncthreads();                   // next tile landed, this one no longer needed
    }

    // partial slab: ws[split][co][q]
    float* wsb = p.ws + (int64_t)split * p.Cout * p.Q;
    if (do_bias) {                                 // column 0 of the ones-product holds the sums
        float* bw = p.bws + (int64_t)split * p.Cout;
        if constexpr (FT16) {
            if ((lane & 15) == 0) {
#pragma unroll
                for (int i = 0; i < NR; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = co0 + wr * (BR / 2) + i * 16 + 4 * (lane >> 4) + r;
                        if (co < p.Cout) bw[co] = accb[i][r] * p.scale;
                    }
            }
        } else {
            if ((lane & 31) == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = co0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        if (co < p.Cout) bw[co] = accb[i][r] * p.scale;
                    }
            }
        }
    }
    if constexpr (FT16) {
        const int l16 = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + wc * 64 + j * 16 + l16;
            if (q >= p.Q) continue;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + wr * (BR / 2) + i * 16 + 4 * kq + r;
                    if (co < p.Cout) wsb[(int64_t)co * p.Q + q] = acc[i][j][r] * p.scale;
                }
            }
        }
    } else {
        const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = q0 + wc * 64 + j * 32 + fr;
            if (q >= p.Q) continue;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (co < p.Cout) wsb[(int64_t)co * p.Q + q] = acc[i][j][r] * p.scale;
                }
            }
        }
    }
#endif
}


// ---------------------------------------------------------------------------------------------------------------------
// Wide tile (round 3): 128 co x 256 q per 256-thread workgroup, K step of 32 pixels, each wave 64 co x 128 q = 4 x 8 MFMA
// tiles.  What bounds the 128 x 128 x 64 kernel above was measured by ablation: with
// the MFMAs removed the operand staging alone (LDS-DMA from L2 + fragment reads) takes as long as the MFMAs alone, the two
// overlap only partly, and the staging runs at the ~15 TB/s (58 GB/s per CU) the L2 -> LDS path delivers - issuing the DMA
// instructions costs nothing (out-of-range pieces: no change).  The lever is bytes per FLOP: this tile moves 24 KB per
// 2.1 MFLOP instead of 32 KB, with 25 % fewer DMA pieces and transposing fragment reads per MFMA: +9..18 % on the layers it
// takes (same-box A/B).  bf16, stride-1 geometries (GEOM 1 / 2) only; still two workgroups per CU (48 KB of LDS, 234 registers).
// X tile [32 px][256 q]: 512-byte rows = two bank periods, so all rows of a transposed fragment read would hit the same
// banks: chunk index XOR ((px & 3) << 2) | ((px >> 3) & 1) << 1 spreads the 8 rows of a half-wave (px = q4 + 8 g') over
// the eight 32-byte bank windows; the LDS row of pixel px is (b4, b2, b3, b1, b0) of its index so that the key is the
// same in all four staging passes (one (tap, ci) decode per lane, not four).
// dY tile [32 px][128 co]: 256-byte rows with the 128 x 128 kernel's key.
// Measured and dropped on this path (same box): a 160 x 256 variant for Cout = 320 (80 x 128 per wave, 160 accumulator
// registers, four X fragments live and re-filled behind their MFMAs: -3 % against the 160 x 128 x 64 tiles it would replace),
// a ring of three stages (72 KB, still two workgroups per CU: +-0 - the
// kernel is not waiting on latency), a 320 x 256 tile on 512 threads (one workgroup per CU, ring of three 40 KB stages,
// 2.3x fewer bytes per FLOP than 128 x 128: +10 % on 14x14 1280->640, -5 % elsewhere - its eight waves run in lockstep
// between workgroup barriers, and its MFMA-only ablation is already slower than this kernel's), and that tile with the two
// halves of the workgroup staggered by one phase (SIMD partners alternate fragment reads and MFMAs: slower still).
template <int GEOM>
__global__ __launch_bounds__(256, 2) void wgrad_wide_kernel(const WgP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int CH = 8, BKP = 32, BR = 128, BQ = 256, ESZ = 2;
    constexpr int NR = BR / 32;                                     // dY fragments per wave
    constexpr int AROWB = BR * ESZ, BROWB = BQ * ESZ;              // 256, 512
    constexpr int NPA = 2, NPB = 4, PASS_BYTES = 4096;
    constexpr int AREG = NPA * PASS_BYTES, BTILE = BKP * BROWB;    // 8 KB, 16 KB
    constexpr int STAGE = AREG + BTILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [buf][dY tile | X tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave & 1, wc = wave >> 1;
    int lid;
    {
        const int nb = gridDim.x, b0 = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = b0 & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b0 >> 3);
    }
    const int tiles = p.rtiles * p.qtiles;
    const int split = lid / tiles;
    const int tix = lid - split * tiles;
    const int rt = tix % p.rtiles, qt = tix / p.rtiles;
    const int co0 = rt * BR, q0 = qt * BQ;

    const int step0 = split * p.steps_per_split;
    const int total_steps = (p.M + BKP - 1) / BKP;
    int nsteps = total_steps - step0;
    if (nsteps > p.steps_per_split) nsteps = p.steps_per_split;

    constexpr uint32_t OOB = 0x80000000u;
    const u32x4 yrs = make_rsrc(p.dy, p.dy_bytes);
    const int ldxB = (int)p.ldx * ESZ, lddyB = (int)p.lddy * ESZ;
    // dY staging: pass j = pixels 16 j + srA, one linear offset + a uniform pass stride
    const int srA = tid >> 4, pcA = tid & 15;
    const int lcA = pcA ^ (((srA & 3) << 2) | ((srA >> 2) & 3));
    const int a_co = co0 + lcA * CH;
    const bool a_ok = a_co < p.Cout;
    int g_mA = step0 * BKP + srA;
    int g_a = (g_mA * (int)p.lddy + a_co) * ESZ;
    // X staging: pass j = pixels pxB + 16 (j >> 1) + 4 (j & 1) (LDS rows 8 j + srB)
    const int srB = tid >> 5, pcB = tid & 31;
    const int lcB = pcB ^ (((srB & 3) << 2) | (((srB >> 2) & 1) << 1));
    const int pxB = ((srB >> 2) << 3) | (srB & 3);
    const int b_q = q0 + lcB * CH;
    const bool b_ok = b_q < p.Q;
    const int b_tap = b_ok ? b_q / p.Cin : 0;
    const int b_ci = b_q - b_tap * p.Cin;
    const int b_kh = b_tap / p.ks, b_kw = b_tap - b_kh * p.ks;
    const int HoWo = p.Ho * p.Wo;
    const int ho_lo = max(0, p.pad - b_kh), wo_lo = max(0, p.pad - b_kw);
    const int ho_hi = min(p.Ho - 1, p.Hi - 1 + p.pad - b_kh), wo_hi = min(p.Wo - 1, p.Wi - 1 + p.pad - b_kw);
    const bool b_in = b_ok && ho_hi >= ho_lo && wo_hi >= wo_lo;
    const unsigned w_rng = (unsigned)(wo_hi - wo_lo);
    const int rm_lo = ho_lo * p.Wo;
    const unsigned rm_rng = (unsigned)((ho_hi + 1) * p.Wo - 1 - rm_lo);
    typedef __attribute__((address_space(3))) char* lds_ptr_t;
    const uint32_t lds_wave = (uint32_t)(size_t)(lds_ptr_t)smem + (uint32_t)__builtin_amdgcn_readfirstlane(wave) * 1024u;
    // the tap shift is folded into a descriptor whose base sits pad rows + pad pixels BEFORE x, so the per-lane offset stays
    // >= 0 (taps that would read below x are never valid)
    int g_mB = step0 * BKP + pxB;
    const int g_shift = (p.pad * p.Wi + p.pad) * ldxB;
    const u32x4 xrs_g = make_rsrc(reinterpret_cast<const char*>(p.x) - g_shift, p.x_bytes + (uint32_t)g_shift);
    int g_b = g_mB * ldxB + (b_kh * p.Wi + b_kw) * ldxB + b_ci * ESZ;
    const int st_rm = BKP % HoWo, st_wo = BKP % p.Wo;
    int g_rm[NPB], g_wo[NPB];
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
        g_rm[j] = (g_mB + 16 * (j >> 1) + 4 * (j & 1)) % HoWo;
        g_wo[j] = g_rm[j] % p.Wo;
    }
    const int a_step = BKP * lddyB, b_step = BKP * ldxB;
    // (validity without compares: see wgrad_pipe_kernel's x_track - sign bits of the differences OR-ed into bit 31 of the offset)
    const int v_rm_lo = b_in ? rm_lo : 0x3FFFFFFF;
    const int v_rm_hi = rm_lo + (int)rm_rng, v_wo_hi = wo_lo + (int)w_rng;
    const int a_bad = a_ok ? 0 : -1;
    auto load_tiles = [&](int buf) {
        const uint32_t adst = lds_wave + (uint32_t)buf * STAGE;
        const uint32_t bdst = adst + AREG;
#pragma unroll
        for (int j = 0; j < NPA; ++j) {
            const int bad = ((p.M - 16 * j - 1) - g_mA) | a_bad;
            const uint32_t aoff = (uint32_t)g_a | ((uint32_t)bad & 0x80000000u);
            lds_dma16s(yrs, adst + j * PASS_BYTES, aoff, (uint32_t)(16 * j * lddyB));
        }
#pragma unroll
        for (int j = 0; j < NPB; ++j) {
            const int po = 16 * (j >> 1) + 4 * (j & 1);
            int bad = (p.M - po - 1) - g_mB;
            if constexpr (GEOM == 1) {
                bad |= (g_rm[j] - v_rm_lo) | (v_rm_hi - g_rm[j]) | (g_wo[j] - wo_lo) | (v_wo_hi - g_wo[j]);
                const int t = g_rm[j] + (st_rm - HoWo), u = g_wo[j] + (st_wo - p.Wo);
                g_rm[j] = (int)min((unsigned)t, (unsigned)(t + HoWo));
                g_wo[j] = (int)min((unsigned)u, (unsigned)(u + p.Wo));
            } else {
                bad |= b_in ? 0 : -1;
            }
            const uint32_t boff = (uint32_t)g_b | ((uint32_t)bad & 0x80000000u);
            lds_dma16s(xrs_g, bdst + j * PASS_BYTES, boff, (uint32_t)(po * ldxB));
        }
        g_mA += BKP; g_mB += BKP; g_a += a_step; g_b += b_step;
    };

    typedef float AccT __attribute__((ext_vector_type(4)));
    AccT acc[NR][8];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    // bias gradient = column sums of the dY tile, from the waves of tile column 0: one MFMA per dY fragment against a ones
    // operand, as in the 128 x 128 kernel.  (The tiles that carry it must not run longer than the others - a launch is often
    // ONE round of workgroups: per-lane VALU sums of the fragments, 80 operations per K step, cost 7x7 1280->1280 11 %.)
    const bool do_bias = p.bws != nullptr && qt == 0 && wc == 0;
    AccT accb[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) accb[i][e] = 0.f;

    // fragment read addresses (byte offsets inside a stage): lane -> (group g, row q4 of the 4 x 16 block, 4-column piece p4)
    const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
    // dY image keys of rows 8g+q4 and 8g+4+q4
    const int keyA0 = (q4 << 2) | ((2 * g) & 3), keyA1 = (q4 << 2) | ((2 * g + 1) & 3);
    const int rowa = (8 * g + q4) * AROWB + 8 * (p4 & 1);
    const int chA = wr * (BR / 16) + (p4 >> 1);
    const int keyB = (q4 << 2) | ((g & 1) << 1);                                                    // both rows of the X image
    const int rowb = AREG + (((g >> 1) << 4) | ((g & 1) << 2) | q4) * BROWB + 8 * (p4 & 1);   // LDS row (b4, 0, b3, b1 b0); second read + 8 rows
    const int chB = wc * 16 + (p4 >> 1);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    auto rd_b = [&](const char* sb, int j) {
        const char* b = sb + rowb + (((chB + 2 * j) ^ keyB) << 4);
        s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b));
        s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b + 8 * BROWB));
        s16x8 bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        return *reinterpret_cast<bf16x8*>(&bv);
    };

    if (nsteps > 0) load_tiles(0);
    wait_vmcnt<0>();
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int buf = st & 1;
        if (st + 1 < nsteps) load_tiles(buf ^ 1);
        const char* sb = smem + buf * STAGE;
        bf16x8 af[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sb + rowa + (((chA + 2 * i) ^ keyA0) << 4)));
            s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sb + rowa + 4 * AROWB + (((chA + 2 * i) ^ keyA1) << 4)));
            s16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            af[i] = *reinterpret_cast<bf16x8*>(&av);
        }
        bf16x8 bf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bf[j] = rd_b(sb, j);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        if (do_bias) {
            const bf16_t one = (bf16_t)1.0f;
            const bf16x8 ones = {one, one, one, one, one, one, one, one};
#pragma unroll
            for (int i = 0; i < NR; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, accb[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt<0>();                   // AFTER the MFMA phase: the DMA of the next tile overlapped it
        __syncthreads();
    }

    float* wsb = p.ws + (int64_t)split * p.Cout * p.Q;
    const int l16 = lane & 15, kq = lane >> 4;
    if (do_bias && l16 == 0) {             // column 0 of the ones-product holds the sums
        float* bw = p.bws + (int64_t)split * p.Cout;
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wr * (BR / 2) + i * 16 + 4 * kq + r;
                if (co < p.Cout) bw[co] = accb[i][r] * p.scale;
            }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int q = q0 + wc * 128 + j * 16 + l16;
        if (q >= p.Q) continue;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wr * (BR / 2) + i * 16 + 4 * kq + r;
                if (co < p.Cout) wsb[(int64_t)co * p.Q + q] = acc[i][j][r] * p.scale;
            }
        }
    }
#endif
}


// ---------------------------------------------------------------------------------------------------------------------
// Pipelined 320 x 192 tile (round 3): ONE wave per SIMD with the whole register file.  4 waves = 2 (co) x 2 (q), each
// 160 co x 96 q = 10 x 6 MFMA tiles (240 accumulator registers: they fit the 256 AGPRs; 320 x 256 = 320 accumulators did not,
// hipcc spilled 417 registers).  K step = 32 pixels, a ring of four 32 KB stages: the tile of step k+3 is requested during
// step k, the tiles of steps k and k+1 have landed at the barrier that opens step k.  There is no second workgroup to hide a
// wave's LDS latency behind, so the wave software-pipelines itself: during the ten MFMAs that use X fragment j it fetches X
// fragment j+1, a share of the NEXT step's dY fragments and a share of the ring's DMA pieces; fragment registers are
// double-buffered (dY: 2 x 10, X: 2 x 1).  32 KB staged per 3.9 MFLOP (1.9x fewer bytes per FLOP than 128 x 128 x 64), 0.53
// transposing reads per MFMA instead of 1.0.  320 divides every Cout of the network, 192 every 3x3 layer's 9 Cin.
// Both images have rows of 1.5 / 2.5 bank periods (384 / 640 bytes: odd pixels start 128 bytes into the period): chunk bits
// 1-2 XOR ((px & 3) >> 1 | ((px >> 3) & 1) << 1) spread the 8 rows of a half-wave's transposed read over the eight 32-byte
// bank windows.  The X image keeps wgrad_wide_kernel's row order (b4, b2, b3, b1 b0) so that a lane's (tap, ci) decode is
// the same in all four staging passes; 24 chunks x 8 rows = 192 lanes per pass: waves 0-2 stage X, all four stage dY.
// bf16, stride-1 geometries.
// Bias gradient (BIAS): every workgroup multiplies ONE dY fragment per K step with a ones operand - fragment qt % 10 of its
// wave's half, so the first ten q tiles of a row tile cover all ten fragments between them (the launcher takes this kernel
// only for >= 10 q tiles).  Spread like this it costs every workgroup one MFMA and two reads in sixty and no branch; as ten
// MFMAs in the workgroups of q tile 0 it would stretch exactly those workgroups by 17 % - and a launch is one round of them.
template <int GEOM, bool BIAS>
__global__ __launch_bounds__(256, 1) void wgrad_pipe_kernel(const WgP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int CH = 8, BKP = 32, BR = 320, BQ = 192, ESZ = 2;
    constexpr int AROWB = BR * ESZ, BROWB = BQ * ESZ;              // 640, 384
    constexpr int ACH = BR / CH, BCH = BQ / CH;                    // 40 / 24 chunks per row
    constexpr int NPA = 5, NPB = 4, APASS = 4096, BPASS = 192 * 16;   // dY: 1280 chunks = 5 x 256 lanes; X: 768 = 4 x 192 lanes
    constexpr int AREG = NPA * APASS, BTILE = BKP * BROWB;         // 20 KB, 12 KB
    constexpr int STAGE = AREG + BTILE;                            // 32 KB
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave & 1, wc = wave >> 1;
    int lid;
    {
        const int nb = gridDim.x, b0 = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = b0 & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b0 >> 3);
    }
    const int tiles = p.rtiles * p.qtiles;
    const int split = lid / tiles;
    const int tix = lid - split * tiles;
    const int rt = tix % p.rtiles, qt = tix / p.rtiles;
    const int co0 = rt * BR, q0 = qt * BQ;
    const int step0 = split * p.steps_per_split;
    const int total_steps = (p.M + BKP - 1) / BKP;
    int nsteps = total_steps - step0;
    if (nsteps > p.steps_per_split) nsteps = p.steps_per_split;

    constexpr uint32_t OOB = 0x80000000u;
    const u32x4 yrs = make_rsrc(p.dy, p.dy_bytes);
    const int ldxB = (int)p.ldx * ESZ, lddyB = (int)p.lddy * ESZ;
    // dY staging: element e = 256 j + tid of pass j sits at (row e / 40, physical chunk e % 40) and fetches the logical chunk
    // physical ^ key(row); rows >= M lie beyond the descriptor's extent and read as zeros (Cout % 320 == 0: no column tail)
    uint32_t a_voff[NPA];
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
        const int e = j * 256 + tid, row = e / ACH, pc = e - row * ACH;
        const int lc = pc ^ ((((row & 3) >> 1) | (((row >> 3) & 1) << 1)) << 1);
        a_voff[j] = (uint32_t)(((step0 * BKP + row) * (int)p.lddy + co0 + lc * CH) * ESZ);
    }
    // X staging (lanes 0..191): pass j = pixels pxB + 16 (j >> 1) + 4 (j & 1), LDS rows 8 j + r8
    const bool xlane = tid < 192;                                   // (waves 0-2: wave-uniform)
    const int r8 = xlane ? tid / BCH : 0, pcB = xlane ? tid - r8 * BCH : 0;
    const int pxB = ((r8 >> 2) << 3) | (r8 & 3);
    const int lcB = pcB ^ ((((pxB & 3) >> 1) | (((pxB >> 3) & 1) << 1)) << 1);
    const int b_q = q0 + lcB * CH;
    const bool b_ok = xlane && b_q < p.Q;
    const int b_tap = b_ok ? b_q / p.Cin : 0;
    const int b_ci = b_q - b_tap * p.Cin;
    const int b_kh = b_tap / p.ks, b_kw = b_tap - b_kh * p.ks;
    const int HoWo = p.Ho * p.Wo;
    const int ho_lo = max(0, p.pad - b_kh), wo_lo = max(0, p.pad - b_kw);
    const int ho_hi = min(p.Ho - 1, p.Hi - 1 + p.pad - b_kh), wo_hi = min(p.Wo - 1, p.Wi - 1 + p.pad - b_kw);
    const bool b_in = b_ok && ho_hi >= ho_lo && wo_hi >= wo_lo;
    const unsigned w_rng = (unsigned)(wo_hi - wo_lo);
    const int rm_lo = ho_lo * p.Wo;
    const unsigned rm_rng = (unsigned)((ho_hi + 1) * p.Wo - 1 - rm_lo);
    typedef __attribute__((address_space(3))) char* lds_ptr_t;
    const uint32_t lds_wave = (uint32_t)(size_t)(lds_ptr_t)smem + (uint32_t)__builtin_amdgcn_readfirstlane(wave) * 1024u;
    int g_mB = step0 * BKP + pxB;
    const int g_shift = (p.pad * p.Wi + p.pad) * ldxB;
    const u32x4 xrs_g = make_rsrc(reinterpret_cast<const char*>(p.x) - g_shift, p.x_bytes + (uint32_t)g_shift);
    int g_b = g_mB * ldxB + (b_kh * p.Wi + b_kw) * ldxB + b_ci * ESZ;
    const int st_rm = BKP % HoWo, st_wo = BKP % p.Wo;
    int g_rm[NPB], g_wo[NPB];
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
        g_rm[j] = (g_mB + 16 * (j >> 1) + 4 * (j & 1)) % HoWo;
        g_wo[j] = g_rm[j] % p.Wo;
    }
    const uint32_t a_step = (uint32_t)(BKP * lddyB);
    const int b_step = BKP * ldxB;
    const bool xwave = __builtin_amdgcn_readfirstlane(wave) < 3;   // this wave's lanes stage X pieces; wave 3's X pieces are
    // out-of-range dummies into a scratch KB behind the ring, so that EVERY wave issues 9 pieces per tile: one vmcnt count
    // for all waves and no branch in the steady-state loop
    const uint32_t lds_dummy = (uint32_t)(size_t)(lds_ptr_t)smem + 4u * STAGE;
    // the DMA pieces of the tile being requested: 5 dY passes, 4 X passes.  An X piece is two halves for the instruction
    // interleave below: x_track computes the piece's offset (and advances the border trackers), x_issue sends it
    auto a_piece = [&](int pc, uint32_t stage_base) {
        lds_dma16(yrs, stage_base + pc * APASS, a_voff[pc]);
        a_voff[pc] += a_step;
    };
    // Validity of a staged X piece WITHOUT compares: every condition is a difference whose sign bit says "outside", the sign
    // bits are OR-ed and moved into bit 31 of the offset (>= 2^31 = out of range for the buffer load).  The compare / s_or /
    // v_cndmask form made a serial VALU -> VCC -> SALU chain of ~16 instructions per piece; this is ~8 independent VALU ops.
    // The (m mod HoWo, m mod Wo) trackers wrap with an unsigned min instead of compare + select.
    const int v_rm_lo = b_in ? rm_lo : 0x3FFFFFFF;                 // (a lane with nothing to fetch: always "outside")
    const int v_rm_hi = rm_lo + (int)rm_rng, v_wo_hi = wo_lo + (int)w_rng;
    auto x_track = [&](int j) -> uint32_t {
        const int po = 16 * (j >> 1) + 4 * (j & 1);
        int bad = (p.M - po - 1) - g_mB;                            // pixel beyond M
        if constexpr (GEOM == 1) {
            bad |= (g_rm[j] - v_rm_lo) | (v_rm_hi - g_rm[j]) | (g_wo[j] - wo_lo) | (v_wo_hi - g_wo[j]);
            const int t = g_rm[j] + (st_rm - HoWo), u = g_wo[j] + (st_wo - p.Wo);
            g_rm[j] = (int)min((unsigned)t, (unsigned)(t + HoWo));
            g_wo[j] = (int)min((unsigned)u, (unsigned)(u + p.Wo));
        } else {
            bad |= b_in ? 0 : -1;
        }
        return (uint32_t)g_b | ((uint32_t)bad & 0x80000000u);
    };
    auto x_issue = [&](int j, uint32_t boff, uint32_t stage_base) {
        const int po = 16 * (j >> 1) + 4 * (j & 1);
        const uint32_t dst = xwave ? stage_base + AREG + j * BPASS : lds_dummy;      // (scalar select)
        lds_dma16s(xrs_g, dst, boff, (uint32_t)(po * ldxB));
        if (j == NPB - 1) { g_mB += BKP; g_b += b_step; }                             // (the last X pass advances the pixel odometer)
    };
    auto dma_piece = [&](int pc, uint32_t stage_base) {
        if (pc < NPA) a_piece(pc, stage_base);
        else x_issue(pc - NPA, x_track(pc - NPA), stage_base);
    };
    // all but the newest `tiles_in_flight` tiles' pieces of this wave have landed (9 pieces per tile and wave)
    auto wait_tiles = [&](int tiles_in_flight) {
        if (tiles_in_flight == 0) wait_vmcnt<0>();
        else if (tiles_in_flight == 1) wait_vmcnt<9>();
        else wait_vmcnt<18>();
    };

    typedef float AccT __attribute__((ext_vector_type(4)));
    AccT acc[10][6];
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
    const int key = ((q4 >> 1) | ((g & 1) << 1)) << 1;             // both images (pixels 8g + q4 and 8g + 4 + q4)
    const int rowa = (8 * g + q4) * AROWB + 8 * (p4 & 1);
    const int chA = wr * 20 + (p4 >> 1);
    // bias gradient: this workgroup's fragment of the dY tile (wave-uniform index), its accumulator and the ones operand
    const int ib = qt % 10;
    const int biasoff = rowa + (((chA + 2 * ib) ^ key) << 4);
    AccT accb = {0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
    {
        const bf16_t one = (bf16_t)1.0f;
        const bf16x8 o = {one, one, one, one, one, one, one, one};
        ones = o;
    }
    const int rowb = AREG + (((g >> 1) << 4) | ((g & 1) << 2) | q4) * BROWB + 8 * (p4 & 1);
    const int chB = wc * 12 + (p4 >> 1);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    auto rd_a = [&](const char* sb, int i) {
        const char* a = sb + rowa + (((chA + 2 * i) ^ key) << 4);
        s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
        s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 4 * AROWB));
        s16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        return *reinterpret_cast<bf16x8*>(&av);
    };
    auto rd_b = [&](const char* sb, int j) {
        const char* b = sb + rowb + (((chB + 2 * j) ^ key) << 4);
        s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b));
        s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b + 8 * BROWB));
        s16x8 bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        return *reinterpret_cast<bf16x8*>(&bv);
    };

    // ---- prologue: tiles 0, 1, 2 requested; tiles 0 and 1 landed; tile 0's dY fragments and X fragment 0 in registers
    const uint32_t lds0 = lds_wave;
#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (t < nsteps) {
#pragma unroll
            for (int pc = 0; pc < 9; ++pc) dma_piece(pc, lds0 + (uint32_t)t * STAGE);
        }
    wait_tiles(nsteps > 2 ? 1 : 0);                    // (step 0 prefetches from tile 1)
    __syncthreads();
    bf16x8 afA[10], afB[10], bf0, bf1;
#pragma unroll
    for (int i = 0; i < 10; ++i) afA[i] = rd_a(smem, i);
    bf0 = rd_b(smem, 0);

    // one K step: the MFMAs of tile k (dY fragments `cur`, staged in buffer `b`), the prefetch of tile k+1's dY fragments into
    // `nxt` from buffer `nb`, the request of tile k+3 into buffer `ib`.  FULL: both are known to exist (steady state: no branch)
    auto kstep = [&](auto full_c, bf16x8 (&cur)[10], bf16x8 (&nxt)[10], int b, int nb, int ib, bool have_next_rt, bool issue_rt) {
        constexpr bool FULL = decltype(full_c)::value;
        const bool have_next = FULL || have_next_rt, issue = FULL || issue_rt;
        const char* sb = smem + b * STAGE;
        const char* sn = smem + nb * STAGE;
        const uint32_t ibase = lds0 + (uint32_t)ib * STAGE;
        // One wave per SIMD: nothing else hides what this wave does between two MFMAs, and an MFMA occupies the pipe for 16
        // cycles = four issue slots.  So the non-MFMA work of a group (6 fragment reads, 1-2 DMA pieces, their address VALU) is
        // dealt out by hand, a slice behind each of the group's ten MFMAs, and pinned there with sched_barrier (as one block in
        // front of the ten MFMAs - the first form of this kernel - the pipe sat idle for ~250 of every ~410 cycles).
        auto trA = [&](const char* sb_, int i, int half) {
            return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sb_ + rowa + (((chA + 2 * i) ^ key) << 4) + half * 4 * AROWB));
        };
        auto trB = [&](const char* sb_, int j_, int half) {
            return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sb_ + rowb + (((chB + 2 * j_) ^ key) << 4) + half * 8 * BROWB));
        };
        auto cat = [&](s16x4 a, s16x4 c) { s16x8 v = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]}; return *reinterpret_cast<bf16x8*>(&v); };
        s16x4 w0 = {0, 0, 0, 0}, w1 = {0, 0, 0, 0};
        // the ones operand is written by hand, two MFMAs ahead of its use: left to the compiler it is a constant that may get
        // rematerialised (v_mov) directly in front of the asm MFMA reading it - a VALU-write -> MFMA-read hazard the hazard
        // recogniser cannot see through the asm (seen with a second bias fragment: garbage sums; tools/mfma_hazard_check.py
        // looks for this pattern in the -S dump)
        u32x4 onesw = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            bf16x8& bc = (j & 1) ? bf1 : bf0;
            bf16x8& bn = (j & 1) ? bf0 : bf1;
            const bool rdb = j < 5 || have_next;                       // the next X fragment: this tile's j+1, or the next tile's 0
            const char* sbn = j < 5 ? sb : sn;
            const int jn = j < 5 ? j + 1 : 0;
            const bool rda = have_next && j < 5;                       // two of the next tile's dY fragments
            const int xj = j < 3 ? j : (j == 5 ? 3 : -1);             // X pass requested in this group (pass 3 last: it advances the odometer)
            s16x4 t0, t1, u0, u1, v0, v1;
            uint32_t xoff = 0;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(cur[i]), "v"(bc));
                if (i == 0 && rdb) t0 = trB(sbn, jn, 0);
                if (i == 1 && rdb) t1 = trB(sbn, jn, 1);
                if (i == 2 && rda) u0 = trA(sn, 2 * j, 0);
                if (i == 3 && rda) u1 = trA(sn, 2 * j, 1);
                if (i == 4 && rda) v0 = trA(sn, 2 * j + 1, 0);
                if (i == 5 && rda) v1 = trA(sn, 2 * j + 1, 1);
                if (i == 6 && issue && j < 5) a_piece(j, ibase);
                if (i == 7 && issue && xj >= 0) xoff = x_track(xj);
                if (i == 8 && issue && xj >= 0) x_issue(xj, xoff, ibase);
                if constexpr (BIAS) {                  // (groups 3 / 4 have the lightest slices)
                    if (j == 4 && i == 7) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) asm volatile("v_mov_b32 %0, 0x3f803f80" : "=v"(onesw[e]));
                    }
                    const bf16x8 ones = *reinterpret_cast<const bf16x8*>(&onesw);
                    if (j == 3 && i == 8) w0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sb + biasoff));
                    if (j == 3 && i == 9) w1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sb + biasoff + 4 * AROWB));
                    if (j == 4 && i == 9) { const bf16x8 wf = cat(w0, w1); asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(accb) : "v"(wf), "v"(ones)); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (rdb) bn = cat(t0, t1);
            if (rda) { nxt[2 * j] = cat(u0, u1); nxt[2 * j + 1] = cat(v0, v1); }
        }
    };
    // (a step has 6 X fragments, so fragment 0 of every tile sits in bf0)
    int b = 0, st = 0;
    for (; st + 4 < nsteps; st += 2) {                 // steady state: tiles st+1 .. st+4 all exist
        kstep(std::true_type{}, afA, afB, b, (b + 1) & 3, (b + 3) & 3, true, true);
        wait_vmcnt<9>();                               // tile st+2 landed; tile st+3 (just requested) may fly
        __syncthreads();
        kstep(std::true_type{}, afB, afA, (b + 1) & 3, (b + 2) & 3, b, true, true);
        wait_vmcnt<9>();
        __syncthreads();
        b = (b + 2) & 3;
    }
    for (; st < nsteps; st += 2) {                     // the last (up to four) steps
        {
            const bool have_next = st + 1 < nsteps, issue = st + 3 < nsteps;
            kstep(std::false_type{}, afA, afB, b, (b + 1) & 3, (b + 3) & 3, have_next, issue);
            if (st + 2 < nsteps) wait_tiles(issue ? 1 : 0);
            __syncthreads();
            b = (b + 1) & 3;
        }
        if (st + 1 < nsteps) {
            const bool have_next = st + 2 < nsteps, issue = st + 4 < nsteps;
            kstep(std::false_type{}, afB, afA, b, (b + 1) & 3, (b + 3) & 3, have_next, issue);
            if (st + 3 < nsteps) wait_tiles(issue ? 1 : 0);
            __syncthreads();
            b = (b + 1) & 3;
        }
    }

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");    // the asm MFMAs are invisible to the hazard recogniser: drain before the accumulators are read
    float* wsb = p.ws + (int64_t)split * p.Cout * p.Q;
    const int l16 = lane & 15, kq = lane >> 4;
    if constexpr (BIAS) {
        if (wc == 0 && qt < 10 && l16 == 0) {          // column 0 of the ones-product holds the sums of fragment ib
            float* bw = p.bws + (int64_t)split * p.Cout;
#pragma unroll
            for (int r = 0; r < 4; ++r) bw[co0 + wr * 160 + ib * 16 + 4 * kq + r] = accb[r] * p.scale;
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int q = q0 + wc * 96 + j * 16 + l16;
        if (q >= p.Q) continue;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wr * 160 + i * 16 + 4 * kq + r;
                wsb[(int64_t)co * p.Q + q] = acc[i][j][r] * p.scale;
            }
        }
    }
#endif
}

// dw[co][ci][tap] (+)= sum_s ws[s][co][tap*Cin + ci]; one block per (co, 256-ci chunk), LDS transpose
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Cout, int Cin, int taps,
                                    int splits, int accumulate) {
    __shared__ float tile[9 * 256];
    const int co = blockIdx.x;
    const int ci0 = blockIdx.y * 256;
    const int nci = min(256, Cin - ci0);
    const int64_t Q = (int64_t)taps * Cin;
    const int64_t slab = (int64_t)Cout * Q;
    for (int e = threadIdx.x; e < taps * nci; e += blockDim.x) {
        const int tap = e / nci, ci = e - tap * nci;
        const float* src = ws + (int64_t)co * Q + (int64_t)tap * Cin + ci0 + ci;
        float a = 0.f;
        for (int s = 0; s < splits; ++s) a += src[(int64_t)s * slab];
        tile[ci * taps + tap] = a;
    }
    __syncthreads();
    float* dst = dw + ((int64_t)co * Cin + ci0) * taps;
    for (int e = threadIdx.x; e < taps * nci; e += blockDim.x) {
        float v = tile[e];
        if (accumulate) v += dst[e];
        dst[e] = v;
    }
}

// dbias[co] (+)= sum_s bws[s][co], fixed order
__device__ __forceinline__ void bias_sum(const float* __restrict__ bws, float* __restrict__ dbias, int Cout, int splits, int accumulate) {
    const int co = blockIdx.x * blockDim.x + threadIdx.x;
    if (co < Cout) {
        float a = 0.f;
        for (int s = 0; s < splits; ++s) a += bws[(int64_t)s * Cout + co];
        if (accumulate) a += dbias[co];
        dbias[co] = a;
    }
}
__global__ void wgrad_bias_sum_kernel(const float* __restrict__ bws, float* __restrict__ dbias, int Cout, int splits, int accumulate) {
    bias_sum(bws, dbias, Cout, splits, accumulate);
}

// native order (dw[co][q], q = tap*Cin + ci): dw (+)= sum_s ws[s], float4, fixed order; the first workgroups also
// finish the bias gradient when bws != NULL
__global__ void wgrad_sum_kernel(const float* __restrict__ ws, float* __restrict__ dw, int64_t n4, int splits, int accumulate,
                                 const float* __restrict__ bws, float* __restrict__ dbias, int Cout, int accumulate_bias) {
    if (bws) bias_sum(bws, dbias, Cout, splits, accumulate_bias);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 a = reinterpret_cast<const f32x4*>(ws)[i];
        for (int s = 1; s < splits; ++s) a += reinterpret_cast<const f32x4*>(ws)[i + (int64_t)s * n4];
        if (accumulate) a += reinterpret_cast<const f32x4*>(dw)[i];
        reinterpret_cast<f32x4*>(dw)[i] = a;
    }
}

static void wgrad_plan(const psg_wgrad_desc* d, WgP& p) {
    int BKP = d->dtype == PSG_BF16 ? 64 : 32;
    p.M = d->B * d->Ho * d->Wo;
    p.taps = d->ksize * d->ksize;
    p.Q = p.taps * d->Cin;
    p.wide = 0;
    {
        // 160-row tiles where they tile Cout exactly and 128 does not (Cout = 320: 2 tiles instead of 2.5 -> 3)
        static int off = -1;
        if (off < 0) { const char* e = getenv("PSG_WGRAD_BR160"); off = (e && atoi(e) == 0) ? 1 : 0; }
        const bool same = d->stride == 1 && d->Hi == d->Ho && d->Wi == d->Wo;
        p.BR = (!off && d->dtype == PSG_BF16 && same && d->Cout % 160 == 0 && d->Cout % 128 != 0) ? 160 : 128;
    }
    {
        // wide (128 x 256) tiles where the 256-column grid wastes < 5 % of the q axis (PSG_WGRAD_WIDE=0: off, A/B runs)
        static int off = -1;
        if (off < 0) { const char* e = getenv("PSG_WGRAD_WIDE"); off = (e && atoi(e) == 0) ? 1 : 0; }
        const bool same = d->stride == 1 && d->Hi == d->Ho && d->Wi == d->Wo;
        const int64_t q256 = ((int64_t)p.Q + 255) / 256 * 256;
        // ... and only where the narrow grid has >= 200 tiles: below that the wide grid leaves the chip to split-K alone (7x7 1280->1280 1x1: 100 narrow tiles, 53 -> 61 us wide; 1280->2560: 200 tiles, 99 -> 90 us)
        const int64_t tiles128 = (int64_t)((d->Cout + 127) / 128) * ((p.Q + 127) / 128);
        if (!off && d->dtype == PSG_BF16 && same && p.BR == 128 && q256 * 100 <= (int64_t)p.Q * 105 && tiles128 >= 200) { p.wide = 1; BKP = 32; }
    }
    // (a weight gradient's tile count is small and its split count is chosen to fill the chip in ONE or two rounds: it always
    //  plans around the reserve)
    int kSlots = 2 * avail_cus();
    {
        // pipelined 320 x 192 tiles (one 256-thread workgroup per CU, wgrad_pipe_kernel): 320 divides Cout, the 192-column grid
        // wastes < 4 %, enough tiles, >= 10 q tiles (they share out the bias gradient's fragments) (PSG_WGRAD_PIPE=0: off)
        static int off = -1;
        if (off < 0) { const char* e = getenv("PSG_WGRAD_PIPE"); off = (e && atoi(e) == 0) ? 1 : 0; }
        static int mint = -1;
        if (mint < 0) { const char* e = getenv("PSG_WGRAD_PIPE_MIN_TILES"); mint = e ? atoi(e) : 15; }
        const bool same = d->stride == 1 && d->Hi == d->Ho && d->Wi == d->Wo;
        const int64_t q192 = ((int64_t)p.Q + 191) / 192 * 192;
        const int64_t tiles320 = (int64_t)(d->Cout / 320) * (q192 / 192);
        const int64_t ext = ((int64_t)p.M + 64) * d->lddy * 2;
        if (!off && d->dtype == PSG_BF16 && same && d->Cout % 320 == 0 && q192 * 100 <= (int64_t)p.Q * 104 && tiles320 >= mint && q192 / 192 >= 10 &&
            ext < 0x7FFFFFF0ll) { p.wide = 3; p.BR = 320; BKP = 32; kSlots = avail_cus(); }
    }
    p.rtiles = (d->Cout + p.BR - 1) / p.BR;
    p.qtiles = p.wide == 3 ? (p.Q + 191) / 192 : (p.wide ? (p.Q + 255) / 256 : (p.Q + 127) / 128);
    const int total_steps = (p.M + BKP - 1) / BKP;
    const int tiles = p.rtiles * p.qtiles;
    // Split-K choice by a makespan model: 2 workgroups (64 KB LDS each) per CU x 256 CUs = 512 slots; a launch runs
    // in ceil(blocks/512) rounds of (steps_per_split + fixed prologue/epilogue) K steps, and every extra split adds
    // one slab of traffic to the deterministic sum pass.  Picking the split count that merely "fills the chip"
    // leaves e.g. 1035 blocks = 2.02 rounds (a third of the time on 11 stragglers); this picks 966 or 483 instead.
    const double step_us = (d->dtype == PSG_BF16 ? 1.05 : 8.0) * (p.BR == 160 ? 1.25 : 1.0), fixed_steps = 6.0;
    const double slab_us = (double)d->Cout * p.Q * 4.0 / 4.0e6;          // one fp32 slab through HBM at ~4 TB/s
    const bool native = d->dw_layout == PSG_W_OHWI || p.taps == 1;
    int max_splits = (total_steps + 7) / 8;                               // at least 8 K steps per split
    if (max_splits > 256) max_splits = 256;
    if (max_splits < 1) max_splits = 1;
    double best = 1e300;
    int best_sps = total_steps;
    for (int want = 1; want <= max_splits; ++want) {
        const int sps = (total_steps + want - 1) / want;
        const int sp = (total_steps + sps - 1) / sps;
        const int64_t blocks = (int64_t)tiles * sp;
        const double rounds = (double)((blocks + kSlots - 1) / kSlots);
        const bool direct = native && sp == 1 && !d->accumulate;
        const double cost = rounds * (sps + fixed_steps) * step_us + (direct ? 0.0 : (sp + 1.0) * slab_us + 3.0);
        if (cost < best - 1e-9) { best = cost; best_sps = sps; }
    }
    {
        static int force = -2;                             // PSG_WGRAD_SPLITS=n pins the split count (kernel experiments)
        if (force == -2) { const char* e = getenv("PSG_WGRAD_SPLITS"); force = e ? atoi(e) : -1; }
        if (force > 0) best_sps = (total_steps + force - 1) / force;
    }
    p.steps_per_split = best_sps;
    p.splits = (total_steps + p.steps_per_split - 1) / p.steps_per_split;
}
// tiles land directly in dw: native order, one split, no accumulation
static bool wgrad_direct(const psg_wgrad_desc* d, const WgP& p) {
    return (d->dw_layout == PSG_W_OHWI || p.taps == 1) && p.splits == 1 && !d->accumulate;
}

}  // namespace psg
using namespace psg;

extern "C" {

int psg_wgrad_init_attrs(void) {
#define PSG_WG_ATTR(G)                                                                                                                   \
    PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<bf16_t, G>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 256)); \
    PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<float, G>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32 * 512));
    PSG_WG_ATTR(0) PSG_WG_ATTR(1) PSG_WG_ATTR(2)
#undef PSG_WG_ATTR
    PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<bf16_t, 1, 160>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * (320 + 256)));
    PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<bf16_t, 2, 160>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * (320 + 256)));
    PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_pipe_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768 + 1024));
    PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_pipe_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768 + 1024));
    PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_pipe_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768 + 1024));
    PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_pipe_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768 + 1024));
    PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_wide_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (8192 + 16384)));
    PSG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_wide_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (8192 + 16384)));
    return PSG_OK;
}

static int wgrad_check(const psg_wgrad_desc* d) {
    PSG_REQUIRE(d, PSG_ERR_ARG, "wgrad: null descriptor");
    PSG_REQUIRE(d->dtype == PSG_F32 || d->dtype == PSG_BF16, PSG_ERR_DTYPE, "wgrad: dtype %d", d->dtype);
    const int CH = d->dtype == PSG_BF16 ? 8 : 4;
    PSG_REQUIRE(d->B > 0 && d->Hi > 0 && d->Wi > 0 && d->Cin > 0 && d->Ho > 0 && d->Wo > 0 && d->Cout > 0, PSG_ERR_SHAPE, "wgrad: non-positive dimension");
    PSG_REQUIRE((d->ksize == 1 && d->pad == 0) || (d->ksize == 3 && d->pad == 1), PSG_ERR_SHAPE, "wgrad: ksize/pad");
    PSG_REQUIRE(d->stride == 1 || d->stride == 2, PSG_ERR_SHAPE, "wgrad: stride");
    PSG_REQUIRE(d->Ho == (d->Hi + 2 * d->pad - d->ksize) / d->stride + 1 && d->Wo == (d->Wi + 2 * d->pad - d->ksize) / d->stride + 1, PSG_ERR_SHAPE, "wgrad: geometry");
    PSG_REQUIRE(d->Cin % CH == 0 && d->Cout % CH == 0, PSG_ERR_SHAPE, "wgrad: Cin=%d Cout=%d must be multiples of %d", d->Cin, d->Cout, CH);
    PSG_REQUIRE(d->ldx >= d->Cin && d->ldx % CH == 0 && d->lddy >= d->Cout && d->lddy % CH == 0, PSG_ERR_SHAPE, "wgrad: row strides");
    PSG_REQUIRE((int64_t)d->B * d->Ho * d->Wo < (1 << 24) && (int64_t)d->B * d->Hi * d->Wi < (1ll << 30), PSG_ERR_SHAPE, "wgrad: too many pixels");
    PSG_REQUIRE(d->dw_layout == PSG_W_OIHW || d->dw_layout == PSG_W_OHWI, PSG_ERR_ARG, "wgrad: dw_layout %d", d->dw_layout);
    return PSG_OK;
}

int64_t psg_conv_wgrad_workspace_bytes(const psg_wgrad_desc* d) {
    if (wgrad_check(d) != PSG_OK) return -1;
    WgP p;
    wgrad_plan(d, p);
    const int64_t slabs = wgrad_direct(d, p) ? 0 : (int64_t)p.splits * d->Cout * p.Q;
    const int64_t bias = (d->dbias && !(p.splits == 1 && !d->accumulate_bias)) ? (int64_t)p.splits * d->Cout : 0;
    return (slabs + bias) * (int64_t)sizeof(float);
}

int psg_conv_wgrad(const psg_wgrad_desc* d, psg_stream_t stream) {
    int rc = wgrad_check(d);
    if (rc) return rc;
    PSG_REQUIRE(d->x && d->dy && d->dw, PSG_ERR_ARG, "wgrad: null pointer");
    PSG_REQUIRE(aligned16(d->x) && aligned16(d->dy) && aligned16(d->dw), PSG_ERR_ALIGN, "wgrad: x/dy/dw must be 16-byte aligned");
    WgP p;
    wgrad_plan(d, p);
    const bool direct = wgrad_direct(d, p);
    const bool native = d->dw_layout == PSG_W_OHWI || p.taps == 1;
    const bool bias_direct = d->dbias && p.splits == 1 && !d->accumulate_bias;
    const int64_t slab_floats = direct ? 0 : (int64_t)p.splits * d->Cout * p.Q;
    const int64_t bias_floats = (d->dbias && !bias_direct) ? (int64_t)p.splits * d->Cout : 0;
    if (slab_floats + bias_floats > 0) {
        PSG_REQUIRE(d->ws && aligned16(d->ws), PSG_ERR_ARG, "wgrad: workspace missing or not 16-byte aligned");
        PSG_REQUIRE(d->ws_bytes >= (slab_floats + bias_floats) * (int64_t)sizeof(float), PSG_ERR_WORKSPACE, "wgrad: workspace too small");
    }
    p.x = d->x; p.dy = d->dy; p.ws = direct ? d->dw : (float*)d->ws; p.ldx = d->ldx; p.lddy = d->lddy;
    p.scale = d->scale == 0.f ? 1.0f : d->scale;
    p.bws = !d->dbias ? nullptr : (bias_direct ? d->dbias : (float*)d->ws + slab_floats);
    p.B = d->B; p.Hi = d->Hi; p.Wi = d->Wi; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    p.ks = d->ksize; p.stride = d->stride; p.pad = d->pad;
    p.inv_HoWo = 1.0f / (float)(d->Ho * d->Wo); p.inv_Wo = 1.0f / (float)d->Wo;
    {
        const int64_t esz = d->dtype == PSG_BF16 ? 2 : 4;
        const int64_t xb = (((int64_t)d->B * d->Hi * d->Wi - 1) * d->ldx + d->Cin) * esz
                           + (int64_t)(d->pad * d->Wi + d->pad) * d->ldx * esz;      // (+ the shifted-base slack of GEOM 1/2)
        const int64_t yb = (((int64_t)p.M - 1) * d->lddy + d->Cout) * esz;
        PSG_REQUIRE(xb < 0x7FFFFFF0ll && yb < 0x7FFFFFF0ll, PSG_ERR_SHAPE, "wgrad: operand extent >= 2 GiB");
        p.x_bytes = (uint32_t)(xb - (int64_t)(d->pad * d->Wi + d->pad) * d->ldx * esz); p.dy_bytes = (uint32_t)yb;
    }
    const int grid = p.rtiles * p.qtiles * p.splits;
    hipStream_t s = (hipStream_t)stream;
    {
        static int dbg = -1;                               // PSG_WGRAD_DEBUG=1: print the plan of every launch
        if (dbg < 0) { const char* e = getenv("PSG_WGRAD_DEBUG"); dbg = e ? atoi(e) : 0; }
        if (dbg) fprintf(stderr, "psg wgrad: M=%d Cout=%d Q=%d wide=%d BR=%d tiles=%dx%d splits=%d steps/split=%d grid=%d direct=%d\n", p.M, d->Cout,
                         p.Q, p.wide, p.BR, p.rtiles, p.qtiles, p.splits, p.steps_per_split, grid, (int)direct);
    }
    {
        const double esz = d->dtype == PSG_BF16 ? 2.0 : 4.0;
        ProfScope prof(PROF_WGRAD, 2.0 * (double)p.M * (double)p.Cout * (double)p.Q, s,
                       ((double)d->B * d->Hi * d->Wi * d->Cin + (double)p.M * d->Cout) * esz + (double)d->Cout * p.Q * 4.0);
        const int geom = d->stride != 1 || d->Hi != d->Ho || d->Wi != d->Wo ? 0 : (d->ksize == 1 ? 2 : 1);
#define PSG_WG_LAUNCH(G)                                                                                               \
        if (d->dtype == PSG_BF16) hipLaunchKernelGGL((wgrad_kernel<bf16_t, G>), dim3(grid), dim3(256), 4 * 64 * 256, s, p); \
        else hipLaunchKernelGGL((wgrad_kernel<float, G>), dim3(grid), dim3(256), 4 * 32 * 512, s, p);
        if (p.wide == 3) {
            constexpr int PL = 4 * 32768 + 1024;
            if (p.bws) {
                if (geom == 1) hipLaunchKernelGGL((wgrad_pipe_kernel<1, true>), dim3(grid), dim3(256), PL, s, p);
                else hipLaunchKernelGGL((wgrad_pipe_kernel<2, true>), dim3(grid), dim3(256), PL, s, p);
            } else {
                if (geom == 1) hipLaunchKernelGGL((wgrad_pipe_kernel<1, false>), dim3(grid), dim3(256), PL, s, p);
                else hipLaunchKernelGGL((wgrad_pipe_kernel<2, false>), dim3(grid), dim3(256), PL, s, p);
            }
        } else if (p.wide) {
            if (geom == 1) hipLaunchKernelGGL((wgrad_wide_kernel<1>), dim3(grid), dim3(256), 2 * (8192 + 16384), s, p);
            else hipLaunchKernelGGL((wgrad_wide_kernel<2>), dim3(grid), dim3(256), 2 * (8192 + 16384), s, p);
        } else if (p.BR == 160) {
            if (geom == 1) hipLaunchKernelGGL((wgrad_kernel<bf16_t, 1, 160>), dim3(grid), dim3(256), 2 * 64 * (320 + 256), s, p);
            else hipLaunchKernelGGL((wgrad_kernel<bf16_t, 2, 160>), dim3(grid), dim3(256), 2 * 64 * (320 + 256), s, p);
        } else if (geom == 0) { PSG_WG_LAUNCH(0) } else if (geom == 1) { PSG_WG_LAUNCH(1) } else { PSG_WG_LAUNCH(2) }
#undef PSG_WG_LAUNCH
    }
    PSG_LAUNCH_CHECK("wgrad");
    const bool bias_pending = d->dbias && !bias_direct;
    if (bias_pending && (direct || !native)) {
        hipLaunchKernelGGL(wgrad_bias_sum_kernel, dim3((d->Cout + 255) / 256), dim3(256), 0, s, p.bws, d->dbias, d->Cout, p.splits, d->accumulate_bias);
        PSG_LAUNCH_CHECK("wgrad_bias_sum");
    }
    if (direct) return PSG_OK;
    if (native) {                                  // Cout*Q is a multiple of 16 (Cin, Cout multiples of 4)
        const int64_t n4 = (int64_t)d->Cout * p.Q / 4;
        int g = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
        if (g < (d->Cout + 255) / 256) g = (d->Cout + 255) / 256;       // enough workgroups to cover the bias channels
        hipLaunchKernelGGL(wgrad_sum_kernel, dim3(g), dim3(256), 0, s, (const float*)d->ws, d->dw, n4, p.splits, d->accumulate,
                           bias_pending ? p.bws : nullptr, d->dbias, d->Cout, d->accumulate_bias);
    } else {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(d->Cout, (d->Cin + 255) / 256), dim3(256), 0, s, (const float*)d->ws, d->dw,
                           d->Cout, d->Cin, p.taps, p.splits, d->accumulate);
    }
    PSG_LAUNCH_CHECK("wgrad_reduce");
    return PSG_OK;
}

}  // extern "C"
