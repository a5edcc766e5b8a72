// Implicit-GEMM convolution / linear: C ABI, descriptor checks and tile choice.  The kernel template lives in
// conv_gemm_kernel.h; each (dtype, tile) instantiation is its own translation unit (conv_tile.hip compiled once per tile,
// see the Makefile) so that the eight of them build in parallel - one file with all of them took five minutes.
#include "conv_gemm_kernel.h"

namespace psg {

#define PSG_TILE_DECL(T, BM, BN)                                                  \
    extern template int launch_conv<T, BM, BN>(const ConvP&, hipStream_t);        \
    extern template int set_conv_attrs<T, BM, BN>();
PSG_TILE_DECL(float, 128, 128) PSG_TILE_DECL(float, 128, 64) PSG_TILE_DECL(float, 64, 64)
PSG_TILE_DECL(bf16_t, 128, 128) PSG_TILE_DECL(bf16_t, 128, 64) PSG_TILE_DECL(bf16_t, 64, 64)
PSG_TILE_DECL(bf16_t, 128, 160) PSG_TILE_DECL(bf16_t, 64, 160)
#undef PSG_TILE_DECL


// persistent pointwise kernel (conv_pw.hip)
bool conv_pw_applicable(const ConvP& p, int dtype);
int launch_conv_pw(const ConvP& p, hipStream_t stream);
int conv_pw_set_attrs();

// tile choice: maximise (useful fraction of the tile grid) x (chip fill of the last wave) x (tile efficiency); then, for
// grids that leave most of the chip idle (small M: sampling, small batches), split-K on top
struct ConvPlan { int BM, BN, splits, kt_per_split; };
static ConvPlan conv_plan(const ConvP& p, int dtype, bool may_split) {
    const int64_t M = p.M;
    ConvPlan pl = {128, 128, 1, p.KT};
    static int force = -2;                             // PSG_CONV_TILE=0..4 pins a candidate (kernel A/B runs)
    if (force == -2) { const char* e = getenv("PSG_CONV_TILE"); force = e ? atoi(e) : -1; }
    const int cand[5][2] = {{128, 128}, {128, 64}, {64, 64}, {128, 160}, {64, 160}};
    double eff[5] = {1.0, 0.78, 0.55, 1.0, 0.70};      // measured relative MFMA efficiency of the tile shapes
    // (a K <= 640 boost for 128x64 - three resident workgroups hiding the short loop ends - paid before the epilogue was
    //  specialised per kind; since then 128x128 wins those layers by 10 %: gemm_direct.py)
    if (dtype != PSG_BF16 || p.ntap > 0) eff[3] = eff[4] = 0.0;   // 160 = 2 x 5 x 16: only the 16x16x32 bf16 tiles divide it
    // resident workgroups on the chip (2 per CU; psg_set_available_cus / psg_set_reserve_rounds)
    const double slots = 2.0 * avail_cus_for((double)((M + 127) / 128) * (double)((p.N + 127) / 128) / 512.0);
    double best = -1.0;
    for (int c = 0; c < 5; ++c) {
        const double tiles = (double)((M + cand[c][0] - 1) / cand[c][0]) * (double)((p.N + cand[c][1] - 1) / cand[c][1]);
        const double useful = (double)M * p.N / (tiles * cand[c][0] * cand[c][1]);
        const double waves = ceil(tiles / slots);
        double score = useful * (tiles / (waves * slots)) * eff[c];
        if (force >= 0) score = (c == force) ? 1.0 : 0.0;
        if (score > best) { best = score; pl.BM = cand[c][0]; pl.BN = cand[c][1]; }
    }
    // Split-K: when the unsplit grid fills less than ~60 % of the chip and the K loop is long enough to share.  Estimated time
    // of (tile shape c, s splits) in K steps of a 128 x 128 tile: rounds x (steps per split + 6 of prologue / epilogue) x
    // (tile area / efficiency) + the partial tiles' trip through memory ((s + 1) x M x N x 4 bytes at ~3 TB/s, 0.9 us per step).
    static int nosplit = -1;
    if (nosplit < 0) { const char* e = getenv("PSG_CONV_SPLITK"); nosplit = (e && atoi(e) == 0) ? 1 : 0; }
    if (may_split && !nosplit && p.ntap == 0 && force < 0 && p.KT >= 16) {
        const double tiles0 = (double)((M + pl.BM - 1) / pl.BM) * (double)((p.N + pl.BN - 1) / pl.BN);
        if (tiles0 < 0.6 * slots || best < 0.6) {
            auto est = [&](int c, int s) {
                const double tiles = (double)((M + cand[c][0] - 1) / cand[c][0]) * (double)((p.N + cand[c][1] - 1) / cand[c][1]);
                const int per = (p.KT + s - 1) / s;
                const double rounds = ceil(tiles * s / slots);
                const double step = (double)cand[c][0] * cand[c][1] / (128.0 * 128.0) / eff[c];
                const double fin = s > 1 ? ((s + 1.0) * (double)M * p.N * 4.0 / 3.0e6 + 4.0) / 0.9 : 0.0;
                return rounds * (per + 6.0) * step + fin;
            };
            int bc = 0;
            for (int c = 0; c < 5; ++c) if (cand[c][0] == pl.BM && cand[c][1] == pl.BN) bc = c;
            double tbest = est(bc, 1);
            const int svals[9] = {2, 3, 4, 5, 6, 8, 10, 12, 16};
            for (int c = 0; c < 5; ++c) {
                if (eff[c] <= 0.0) continue;
                for (int si = 0; si < 9; ++si) {
                    const int s = svals[si];
                    if (p.KT / s < 8) break;                       // at least 8 K steps per split
                    if ((double)s * M * p.N * 4.0 > 256.0e6) break;   // partial tiles: <= 256 MB
                    const double t = est(c, s);
                    if (t < tbest * 0.9) { tbest = t; pl.BM = cand[c][0]; pl.BN = cand[c][1]; pl.splits = s; }
                }
            }
            if (pl.splits > 1) { pl.kt_per_split = (p.KT + pl.splits - 1) / pl.splits; pl.splits = (p.KT + pl.kt_per_split - 1) / pl.kt_per_split; }
        }
    }
    return pl;
}

static int choose_and_launch(const ConvP& p0, int dtype, hipStream_t s) {
    ConvP p = p0;
    ConvPlan pl = conv_plan(p, dtype, p.ws != nullptr);
    p.splits = 1; p.kt_per_split = p.KT;
    if (pl.splits > 1) {
        if (p.ws != nullptr && (int64_t)pl.splits * p.M * p.N * 4 <= p.ws_bytes) { p.splits = pl.splits; p.kt_per_split = pl.kt_per_split; }
        else pl = conv_plan(p, dtype, false);          // workspace too small: the best UNSPLIT tile, not the split plan's tile
    }
    const int BM = pl.BM, BN = pl.BN;
#ifdef PSG_ABL
    if (PSG_ABL & 8) { const char* e = getenv("PSG_DBG_PTR"); p.ws = (e && p.splits <= 1) ? reinterpret_cast<float*>(strtoull(e, nullptr, 0)) : (p.splits <= 1 ? nullptr : p.ws); }
#endif
    {
        static int dbg = -1;                               // PSG_CONV_DEBUG=1: print the tile chosen for every launch
        if (dbg < 0) { const char* e = getenv("PSG_CONV_DEBUG"); dbg = e ? atoi(e) : 0; }
        if (dbg) fprintf(stderr, "psg conv: M=%d N=%d Cin=%d ks=%d tr=%d fast=%d epi_lds=%d -> tile %dx%d splits=%d\n", p.M, p.N, p.Cin, p.ks, p.transposed,
                         p.fast, p.epi_lds, BM, BN, p.splits);
    }
    if (BM == 128 && BN == 128 && p.splits == 1 && conv_pw_applicable(p, dtype)) {
        const int rc = launch_conv_pw(p, s);
        if (rc != -1) return rc;
    }
    if (BN == 160 && BM == 64) return launch_conv<bf16_t, 64, 160>(p, s);
    if (BN == 160) return launch_conv<bf16_t, 128, 160>(p, s);
    if (dtype == PSG_F32) {
        if (BM == 128 && BN == 128) return launch_conv<float, 128, 128>(p, s);
        if (BM == 128 && BN == 64) return launch_conv<float, 128, 64>(p, s);
        return launch_conv<float, 64, 64>(p, s);
    } else {
        if (BM == 128 && BN == 128) return launch_conv<bf16_t, 128, 128>(p, s);
        if (BM == 128 && BN == 64) return launch_conv<bf16_t, 128, 64>(p, s);
        return launch_conv<bf16_t, 64, 64>(p, s);
    }
}

}  // namespace psg
using namespace psg;

extern "C" {

int psg_conv_init_attrs(void) {
    int rc;
    if ((rc = set_conv_attrs<float, 128, 128>())) return rc;
    if ((rc = set_conv_attrs<bf16_t, 128, 128>())) return rc;
    if ((rc = set_conv_attrs<float, 128, 64>())) return rc;
    if ((rc = set_conv_attrs<bf16_t, 128, 64>())) return rc;
    if ((rc = set_conv_attrs<float, 64, 64>())) return rc;
    if ((rc = set_conv_attrs<bf16_t, 64, 64>())) return rc;
    if ((rc = set_conv_attrs<bf16_t, 128, 160>())) return rc;
    if ((rc = set_conv_attrs<bf16_t, 64, 160>())) return rc;
    if ((rc = conv_pw_set_attrs())) return rc;
    return PSG_OK;
}

static int conv_setup(const psg_conv_desc* d, ConvP& p) {
    PSG_REQUIRE(d && d->x && d->w && d->y, PSG_ERR_ARG, "conv_fwd: null pointer");
    PSG_REQUIRE(d->dtype == PSG_F32 || d->dtype == PSG_BF16, PSG_ERR_DTYPE, "conv_fwd: dtype %d", d->dtype);
    const int CH = d->dtype == PSG_BF16 ? 8 : 4;
    PSG_REQUIRE(d->B > 0 && d->Hi > 0 && d->Wi > 0 && d->Cin > 0 && d->Ho > 0 && d->Wo > 0 && d->Cout > 0, PSG_ERR_SHAPE,
                "conv_fwd: non-positive dimension");
    PSG_REQUIRE((d->ksize == 1 && d->pad == 0) || (d->ksize == 3 && d->pad == 1) ||
                (d->ksize == 4 && (d->pad == 1 || d->pad == 2) && d->stride == 2 && !d->transposed),
                PSG_ERR_SHAPE, "conv_fwd: ksize/pad/stride %d/%d/%d", d->ksize, d->pad, d->stride);
    PSG_REQUIRE(d->stride == 1 || d->stride == 2, PSG_ERR_SHAPE, "conv_fwd: stride %d", d->stride);
    PSG_REQUIRE(d->Cin % CH == 0, PSG_ERR_SHAPE, "conv_fwd: Cin=%d must be a multiple of %d", d->Cin, CH);
    PSG_REQUIRE(d->Cout % 4 == 0, PSG_ERR_SHAPE, "conv_fwd: Cout=%d must be a multiple of 4", d->Cout);
    PSG_REQUIRE(d->ldx >= d->Cin && d->ldx % CH == 0, PSG_ERR_SHAPE, "conv_fwd: ldx=%ld (Cin=%d)", (long)d->ldx, d->Cin);
    PSG_REQUIRE(d->ldy >= d->Cout && d->ldy % 4 == 0, PSG_ERR_SHAPE, "conv_fwd: ldy=%ld", (long)d->ldy);
    PSG_REQUIRE(aligned16(d->x) && aligned16(d->w) && aligned16(d->y), PSG_ERR_ALIGN, "conv_fwd: x/w/y must be 16-byte aligned");
    PSG_REQUIRE(!d->bias || aligned16(d->bias), PSG_ERR_ALIGN, "conv_fwd: bias alignment");
    PSG_REQUIRE(!d->rowadd || (aligned16(d->rowadd) && d->ld_rowadd % 4 == 0), PSG_ERR_ALIGN, "conv_fwd: rowadd alignment");
    PSG_REQUIRE(!d->residual || (aligned16(d->residual) && d->ld_residual % 4 == 0), PSG_ERR_ALIGN, "conv_fwd: residual alignment");
    PSG_REQUIRE(!d->preact || (aligned16(d->preact) && d->ld_preact % 4 == 0), PSG_ERR_ALIGN, "conv_fwd: preact alignment");
    PSG_REQUIRE(!d->dact_u || (aligned16(d->dact_u) && d->ld_dact % 4 == 0), PSG_ERR_ALIGN, "conv_fwd: dact_u alignment");
    PSG_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, PSG_ERR_ARG, "conv_fwd: drop_p %f", d->drop_p);
    PSG_REQUIRE(!(d->residual && d->dact_u), PSG_ERR_ARG, "conv_fwd: residual and dact_u are mutually exclusive");
    PSG_REQUIRE((d->flags & ~(PSG_CONV_SAVE_DACT | PSG_CONV_DACT_MUL | PSG_CONV_GENERIC_EPILOGUE)) == 0, PSG_ERR_ARG, "conv_fwd: unknown flags 0x%x", d->flags);
    PSG_REQUIRE(!(d->flags & PSG_CONV_SAVE_DACT) || (d->preact && !d->dact_u), PSG_ERR_ARG, "conv_fwd: SAVE_DACT needs preact (forward form)");
    PSG_REQUIRE(!(d->flags & PSG_CONV_DACT_MUL) || (d->dact_u && d->drop_p == 0.f), PSG_ERR_ARG,
                "conv_fwd: DACT_MUL needs dact_u and drop_p = 0 (the saved derivative already carries the mask)");
    // geometry consistency
    if (!d->transposed) {
        PSG_REQUIRE(d->Ho == (d->Hi + 2 * d->pad - d->ksize) / d->stride + 1 && d->Wo == (d->Wi + 2 * d->pad - d->ksize) / d->stride + 1,
                    PSG_ERR_SHAPE, "conv_fwd: output %dx%d inconsistent with input %dx%d k%d s%d p%d", d->Ho, d->Wo, d->Hi, d->Wi, d->ksize, d->stride, d->pad);
    } else {
        PSG_REQUIRE(d->Hi == (d->Ho + 2 * d->pad - d->ksize) / d->stride + 1 && d->Wi == (d->Wo + 2 * d->pad - d->ksize) / d->stride + 1,
                    PSG_ERR_SHAPE, "conv_fwd(transposed): grad grid %dx%d inconsistent with input grid %dx%d", d->Hi, d->Wi, d->Ho, d->Wo);
    }
    const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
    PSG_REQUIRE(M < (1ll << 30) && (int64_t)d->B * d->Hi * d->Wi < (1ll << 30), PSG_ERR_SHAPE, "conv_fwd: too many pixels");
    const int taps = d->ksize * d->ksize;
    const int64_t Kpad = psg_kpad((int64_t)taps * d->Cin, d->dtype);
    PSG_REQUIRE(d->ldw == 0 || (d->ldw >= Kpad && d->ldw % CH == 0), PSG_ERR_SHAPE, "conv_fwd: ldw=%ld < Kpad=%ld", (long)d->ldw, (long)Kpad);

    p.x = d->x; p.w = d->w; p.y = d->y; p.bias = d->bias; p.rowadd = d->rowadd; p.residual = d->residual;
    p.preact = d->preact; p.dact_u = d->dact_u;
    p.ldw = d->ldw > 0 ? d->ldw : Kpad;
    p.ldx = d->ldx; p.ldy = d->ldy; p.ldra = d->ld_rowadd; p.ldres = d->ld_residual; p.ldpre = d->ld_preact; p.lddact = d->ld_dact;
    p.B = d->B; p.Hi = d->Hi; p.Wi = d->Wi; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.N = d->Cout;
    p.ks = d->ksize; p.stride = d->stride; p.pad = d->pad; p.transposed = d->transposed;
    p.M = (int)M; p.taps = taps; p.cpt = d->Cin / CH; p.Kpad = (int)Kpad; p.KT = (int)(Kpad / (8 * CH));
    p.fast = (p.cpt % 8 == 0) ? 1 : 0; p.tpt = p.fast ? p.cpt / 8 : 1;
    p.act = d->act; p.alpha = d->alpha; p.flags = d->flags;
    {
        // coalesced (LDS-staged) stores need 16-byte row chunks: bf16, channel counts and row strides multiples of 8
        static int off = -1;
        if (off < 0) { const char* e = getenv("PSG_EPI_LDS"); off = (e && atoi(e) == 0) ? 1 : 0; }
        p.epi_lds = (!off && d->dtype == PSG_BF16 && d->Cout % 8 == 0 && d->ldy % 8 == 0 && (!d->preact || d->ld_preact % 8 == 0) &&
                     (!d->residual || d->ld_residual % 8 == 0) && (!d->dact_u || d->ld_dact % 8 == 0)) ? 1 : 0;
    }
    { static int gen = -1; if (gen < 0) { const char* e = getenv("PSG_EPI_KINDS"); gen = (e && atoi(e) == 0) ? 1 : 0; } p.epi_generic = (gen || (d->flags & PSG_CONV_GENERIC_EPILOGUE)) ? 1 : 0; }
    p.drop_thresh = d->drop_p > 0.f ? drop_thresh(d->drop_p) : 0u;
    p.drop_scale = d->drop_p > 0.f ? 1.0f / (1.0f - d->drop_p) : 1.0f;
    p.drop_seed = d->drop_seed;
    p.seed_dev = seed_source();
    p.mtiles = p.ntiles = 0;
    {
        const int64_t esz = d->dtype == PSG_BF16 ? 2 : 4;
        const int64_t xb = (((int64_t)d->B * d->Hi * d->Wi - 1) * d->ldx + d->Cin) * esz;
        const int64_t wb = (((int64_t)d->Cout - 1) * p.ldw + Kpad) * esz;
        PSG_REQUIRE(xb < 0x7FFFFFF0ll && wb < 0x7FFFFFF0ll, PSG_ERR_SHAPE, "conv_fwd: operand extent >= 2 GiB (x %ld B, w %ld B)", (long)xb, (long)wb);
        p.x_bytes = (uint32_t)xb; p.w_bytes = (uint32_t)wb;
    }
    p.ws = (d->ws && aligned16(d->ws)) ? reinterpret_cast<float*>(d->ws) : nullptr;
    p.ws_bytes = p.ws ? d->ws_bytes : 0;
    p.splits = 1; p.kt_per_split = p.KT;
    p.ntap = 0; p.sub_h0 = p.sub_w0 = p.sub_nH = p.sub_nW = 0;
    return PSG_OK;
}

// Bytes of workspace with which psg_conv_fwd would split the K axis of this launch over several workgroups (0: it would not -
// large grids, stride-2 data gradients, short K).  The workspace is optional: without it (ws NULL / too small) the launch
// runs unsplit.  Meant for small-M launches (sampling, small batches); the call costs a plan evaluation on the host.
int64_t psg_conv_fwd_workspace_bytes(const psg_conv_desc* d) {
    ConvP p;
    if (conv_setup(d, p) != PSG_OK) return -1;
    if (p.transposed && p.stride == 2 && p.fast && p.ks == 3) return 0;
    const ConvPlan pl = conv_plan(p, d->dtype, true);
    return pl.splits > 1 ? (int64_t)pl.splits * p.M * p.N * 4 : 0;
}

int psg_conv_fwd(const psg_conv_desc* d, psg_stream_t stream) {
    ConvP p;
    const int rc0 = conv_setup(d, p);
    if (rc0) return rc0;
    hipStream_t s = (hipStream_t)stream;
    if (p.transposed && p.stride == 2 && p.fast && p.ks == 3) {
        // Data gradient of a stride-2 conv: a result pixel (ho, wo) is reached only by the taps with
        // kh = (ho + pad) mod 2 (mod 2), same for kw - 1, 2, 2 or 4 of the 9.  One launch per parity class, each a
        // dense stride-1-like gather on the class's own grid: 2.25 taps per pixel on average instead of 9.
        for (int ah = 0; ah < 2; ++ah)
            for (int aw = 0; aw < 2; ++aw) {
                ConvP q = p;
                q.sub_h0 = (ah - p.pad) & 1; q.sub_w0 = (aw - p.pad) & 1;
                q.sub_nH = (p.Ho - q.sub_h0 + 1) >> 1; q.sub_nW = (p.Wo - q.sub_w0 + 1) >> 1;
                if (q.sub_nH <= 0 || q.sub_nW <= 0) continue;
                q.ntap = 0;
                for (int kh = ah; kh < 3; kh += 2)
                    for (int kw = aw; kw < 3; kw += 2) {
                        q.tap_dh[q.ntap] = (q.sub_h0 + p.pad - kh) / 2;
                        q.tap_dw[q.ntap] = (q.sub_w0 + p.pad - kw) / 2;
                        q.tap_wi[q.ntap] = kh * 3 + kw;
                        ++q.ntap;
                    }
                q.M = p.B * q.sub_nH * q.sub_nW;
                q.taps = q.ntap;                           // (profiling: useful FLOPs of this class)
                q.KT = q.ntap * p.tpt;
                const int rc = choose_and_launch(q, d->dtype, s);
                if (rc) return rc;
            }
        return PSG_OK;
    }
    return choose_and_launch(p, d->dtype, s);
}

}  // extern "C"
