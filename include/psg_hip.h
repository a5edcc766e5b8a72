/*
 * psg_hip.h — C ABI of libpsg_hip.so: the MI355X (gfx950) kernels behind the
 * U-Net denoising train step of GabrieleConte/pokemon-sprite-generator.
 *
 * The reference has NO native code, plugin registry or FFI: its hot path is
 * PyTorch ATen ops called from src/models/unet.py and
 * src/training/improved_diffusion_trainer.py.  Each entry point below replaces
 * the ATen call(s) made at the cited reference lines (paths relative to the
 * reference root).  The host side (the .py files of pokemon_sprite_generator_amd) binds
 * these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *  - Plain pointers and sizes only; no torch types.  All pointers are DEVICE
 *    pointers borrowed for the duration of the call.  Nothing is allocated or
 *    freed on behalf of the caller: scratch is a caller-provided workspace
 *    (query *_workspace_bytes first).
 *  - Every launch is asynchronous on `stream` (a hipStream_t); no hidden
 *    synchronisation, no host reads: graph-capture safe.
 *  - Return 0 on success, <0 on error (enum below); psg_last_error() returns a
 *    thread-local message.  NaN/Inf in data is reported as data (flags), never
 *    as an error.
 *  - Activations are channels-last: a [B, H, W, C] (or [B, L, C] token) tensor
 *    is a row-major matrix of B*H*W rows with an explicit row stride `ld`
 *    (in elements), so slices of wider buffers are addressable.  dtype selects
 *    the element type of activations and prepared weights (PSG_F32 / PSG_BF16);
 *    statistics, biases, accumulators, master weights and gradients of
 *    parameters are always fp32.
 */
#ifndef PSG_HIP_H
#define PSG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* psg_stream_t; /* hipStream_t */

enum psg_status {
    PSG_OK = 0,
    PSG_ERR_SHAPE = -1,
    PSG_ERR_DTYPE = -2,
    PSG_ERR_ALIGN = -3,
    PSG_ERR_WORKSPACE = -4,
    PSG_ERR_HIP = -5,
    PSG_ERR_ARG = -6
};
enum psg_dtype { PSG_F32 = 0, PSG_BF16 = 1 };
enum psg_act { PSG_ACT_NONE = 0, PSG_ACT_SILU = 1, PSG_ACT_GELU = 2, PSG_ACT_RELU = 3, PSG_ACT_TANH = 4 };   /* RELU / TANH: the frozen VAE (vae_decoder.py:82-91,185) */
/* Bits of the device-side NaN/Inf flag word of one train step (the reference's five host-side
 * check_for_nans scans, improved_diffusion_trainer.py:353-393).  A step whose word has any bit of
 * PSG_FLAG_SKIP_MASK set is skipped like the reference's `continue`: no optimizer step, no scheduler
 * step.  PSG_FLAG_FALLBACK alone does NOT skip: add_noise replaced a non-finite result by
 * x0 + 0.1*noise (:61-63) and the batch trains on that. */
enum psg_flag {
    PSG_FLAG_NOISY_BAD = 1,   /* noisy latent still non-finite after the fallback (:376) */
    PSG_FLAG_T_RANGE = 2,     /* a timestep outside [0, num_t) (the reference raises an IndexError -> batch skipped) */
    PSG_FLAG_PRED_BAD = 4,    /* predicted noise non-finite (:383) */
    PSG_FLAG_LOSS_BAD = 8,    /* loss non-finite (:390) */
    PSG_FLAG_FALLBACK = 16,   /* add_noise took the fallback (informational) */
    PSG_FLAG_INPUT_BAD = 32,  /* text embedding / clean latent non-finite (:353,359; set by the host-side caller) */
    PSG_FLAG_SKIP_MASK = 47
};

const char* psg_last_error(void);
int psg_version(void);
/* Select device, raise LDS limits of the big kernels.  Idempotent. */
int psg_init(int device);

/* ---------------------------------------------------------------------------
 * Noise schedule / sampler elementwise ops (bit-exact fp32, contraction off)
 * ------------------------------------------------------------------------- */

/* NoiseScheduler.add_noise — improved_diffusion_trainer.py:50-65 (+ clamp :363).
 * out[b,i] = tabA[t[b]] * clamp?(x0[b,i]) + tabB[t[b]] * noise[b,i]; two rounded
 * multiplies and one rounded add, bit-identical to the CPU path.  *flag (int32,
 * device) is OR-ed with PSG_FLAG_FALLBACK if any output is NaN/Inf, with
 * PSG_FLAG_T_RANGE if any t[b] is outside [0,num_t).  Zero *flag before the call. */
int psg_noise_add_f32(const float* x0, const float* noise, const int64_t* t, const float* tabA,
                      const float* tabB, float* out, int32_t* flag, int64_t B, int64_t chw,
                      int num_t, int do_clamp, psg_stream_t stream);
/* The reference's fallback (:61-63): if (*flag & PSG_FLAG_FALLBACK) out = clamp?(x0) + 0.1*noise, and
 * *flag |= PSG_FLAG_NOISY_BAD if that is still non-finite (the trainer's re-check, :376).  Device-side
 * test of the flag; no host sync. */
int psg_noise_fallback_f32(const float* x0, const float* noise, float* out, int32_t* flag,
                           int64_t n, int do_clamp, psg_stream_t stream);
/* ddpm_sample update — improved_diffusion_trainer.py:543-567.  step tables hold the per-timestep
 * scalars c1=1/sqrt(alpha_t), c2=beta_t/sqrt(1-alphabar_t), sigma=sqrt(beta_t) (fp32, computed on
 * the host with the reference's torch ops); t_index selects the entry on the device so the launch
 * is graph-replayable: x <- c1*(x - c2*eps) [+ sigma*z if t>0].  t_dev: int32 device scalar. */
int psg_ddpm_update_f32(float* x, const float* eps, const float* z, const float* c1, const float* c2,
                        const float* sigma, const int32_t* t_dev, int64_t n, psg_stream_t stream);

/* Update steps of the two other samplers that consume the trained U-Net, in place on x (fp32, the reference's operation
 * order, bit-exact):
 *   mode 1 — NoiseScheduler.sample_previous_timestep, src/training/final_trainer.py:52-71:
 *            x <- c0 * (x - (c1 * eps) / c2) [+ c3 * z]   with c0 = sqrt(1/alpha_t), c1 = beta_t, c2 = sqrt(1 - abar_t),
 *            c3 = sqrt(posterior_variance_t); z == NULL for t == 0
 *   mode 2 — the t == 0 step of FinalPokemonGenerator.forward, final_trainer.py:203:  x <- x - eps
 *   mode 3 — PokemonGradioGenerator.ddpm_sample, gradio_app.py:343-358:
 *            x <- (x - c0 * eps) / c1  with c0 = (1 - alpha_t)/sqrt(1 - abar_t), c1 = sqrt(alpha_t); then, when z != NULL
 *            (not the last step and next_t > 0), x <- c2 * x + c3 * z with c2 = sqrt(alpha_next), c3 = sqrt(1 - alpha_next)
 * The scalars are computed by the host with the reference's own torch expressions. */
int psg_sampler_update_f32(float* x, const float* eps, const float* z, int mode, float c0, float c1, float c2,
                           float c3, int64_t n, psg_stream_t stream);

/* VAE reparameterisation, src/models/vae_decoder.py:120-123: out = mu + eps * exp(0.5 * logvar) (fp32, separately rounded). */
int psg_reparam_f32(const float* mu, const float* logvar, const float* eps, float* out, int64_t n, psg_stream_t stream);

/* SmoothL1Loss(beta) mean + its gradient — improved_diffusion_trainer.py:300,388,396.
 * loss_out: fp32 device scalar; grad (may be NULL) = dL/dpred * grad_scale.  Deterministic
 * two-stage reduction.  ws: >= psg_reduce_workspace_bytes() bytes. */
int psg_smooth_l1_f32(const float* pred, const float* target, float* grad, float* loss_out,
                      int32_t* nan_flag, float beta, float grad_scale, int64_t n, void* ws,
                      psg_stream_t stream);
int64_t psg_reduce_workspace_bytes(void);

/* ---------------------------------------------------------------------------
 * Layout / small ops at the boundary (NCHW fp32 <-> channels-last dtype)
 * ------------------------------------------------------------------------- */
/* [B,C,HW] fp32 -> [B,HW,C] dtype (UNet.forward input, unet.py:448) and back (:507-509). */
int psg_nchw_to_nhwc(const float* src, void* dst, int64_t ld_dst, int B, int C, int HW, int dtype,
                     psg_stream_t stream);
int psg_nhwc_to_nchw(const void* src, int64_t ld_src, float* dst, int B, int C, int HW, int dtype,
                     psg_stream_t stream);
/* AdaptiveAvgPool1d(1) over tokens — unet.py:322,445: text [B,S,D] fp32 -> out[B, :D] (dtype),
 * and the dtype copy of the token embeddings themselves (text_cast, may be NULL). */
int psg_text_pool(const float* text, void* pooled, int64_t ld_pooled, void* text_cast, int B, int S,
                  int D, int dtype, psg_stream_t stream);
/* Sinusoidal features — unet.py:47-50: out[b, :half]=sin(t*coeff), out[b, half:]=cos(t*coeff),
 * accurate sinf/cosf (arguments reach 999 rad). */
int psg_timestep_sinusoid(const int64_t* t, const float* coeff, void* out, int64_t ld_out, int B,
                          int half, int dtype, psg_stream_t stream);
/* nn.Upsample(size, bilinear, align_corners=False) — unet.py:365,375,385, channels-last. */
int psg_upsample_bilinear_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, int B, int Hi, int Wi,
                              int Ho, int Wo, int C, int dtype, psg_stream_t stream);
int psg_upsample_bilinear_bwd(const void* dy, int64_t lddy, void* dx, int64_t lddx, int B, int Hi,
                              int Wi, int Ho, int Wo, int C, int dtype, psg_stream_t stream);
/* y = a + b (elementwise, strided rows); used for gradient fan-in of skip tensors. */
int psg_add(const void* a, int64_t lda, const void* b, int64_t ldb, void* y, int64_t ldy, int64_t rows,
            int cols, int dtype, psg_stream_t stream);
/* Dropout under hipGraph replay.  Every mask in this library is a stateless hash of (seed, element index); the seeds are
 * launch arguments, so a captured train step would replay the SAME masks forever.  After psg_set_seed_source(p) every launch
 * that draws a mask adds the 64-bit word at device address p to its seed when it RUNS (conv / Linear epilogues, attention,
 * psg_dropout_apply, psg_epilogue_bwd - forward and backward of a step read the same value, so they still agree); the caller
 * advances the word on the device between steps.  NULL (default) restores the plain seeds.  Process-wide (one process per
 * GPU).  Reference: nn.Dropout / MultiheadAttention(dropout=) draw fresh masks every step (unet.py:160-187). */
int psg_set_seed_source(const uint64_t* seed_dev);

/* y = a (+ b (+ c)), b / c may be NULL (c needs b): row-strided [rows][cols] operands, 16-byte chunks (cols and row strides
 * multiples of 8 bf16 / 4 fp32), the fp32 sum rounded once.  One source: the strided copy of a skip tensor into its half of
 * the decoder's concat buffer (reference unet.py:480-504 `torch.cat([x, skip], dim=1)`: the x half is written there by
 * its producer); three: the gradient fan-in of a skip tensor (its three consumers) in one pass. */
int psg_sum_rows(const void* a, int64_t lda, const void* b, int64_t ldb, const void* c, int64_t ldc, void* y, int64_t ldy,
                 int64_t rows, int cols, int dtype, psg_stream_t stream);

/* ---------------------------------------------------------------------------
 * GroupNorm (+ fused SiLU) — unet.py:79,89,115,127,397 (eps 1e-5) and
 * :156-157,214,231 (eps 1e-6 on [B,C,L]); channels-last [B, HW, C].
 * mean/rstd: fp32 [B*G].  Biased variance, eps inside the sqrt.
 * ------------------------------------------------------------------------- */
int psg_groupnorm_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma,
                      const float* beta, float* mean, float* rstd, int B, int HW, int C, int G,
                      float eps, int silu, int dtype, void* ws, psg_stream_t stream);
int64_t psg_groupnorm_fwd_workspace_bytes(int B, int G);
/* dx (may alias dy), dgamma/dbeta fp32 [C] (overwritten, or accumulated if accumulate!=0).
 * ws: >= psg_groupnorm_bwd_workspace_bytes(B,C) bytes. */
int psg_groupnorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma,
                      const float* beta, const float* mean, const float* rstd, void* dx, int64_t lddx,
                      float* dgamma, float* dbeta, int B, int HW, int C, int G, int silu, int accumulate,
                      int dtype, void* ws, psg_stream_t stream);
/* Same, with the gradient of the normalised tensor's OTHER consumer added in the same pass:
 * dx = groupnorm_bwd(dy) + dres.  Every GroupNorm input of the U-Net also feeds a path that bypasses the norm
 * (ResBlock skip, unet.py:132; the attention residuals :220,238), so autograd's separate accumulation pass over
 * the activation gradient disappears.  dres may be NULL (then identical to psg_groupnorm_bwd) and may alias dx. */
int psg_groupnorm_bwd_res(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma,
                          const float* beta, const float* mean, const float* rstd, const void* dres,
                          int64_t lddres, void* dx, int64_t lddx, float* dgamma, float* dbeta, int B, int HW,
                          int C, int G, int silu, int accumulate, int dtype, void* ws, psg_stream_t stream);
int64_t psg_groupnorm_bwd_workspace_bytes(int B, int C);

/* ---------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear on MFMA — nn.Conv2d 3x3 s1/s2 p1 and 1x1
 * (unet.py:80,90,96,325,335,342,349,366,376,386,399) and every nn.Linear
 * (unet.py:28-34,83,86,176,181-187 and the MHA in/out projections :160-173).
 *
 *   y[m, n] = residual[m, n] + alpha * drop(act(acc[m, n] + bias[n] + rowadd[b(m), n])) * dact'(...)
 *   acc[m, n] = sum_{kh,kw,ci} x[pix(m,kh,kw), ci] * w[n, (kh,kw,ci)]
 *
 * m indexes output pixels (b, ho, wo) row-major; a Linear is ksize=1, H=W=1.
 * `w` is a PREPARED weight [N][Kpad] (psg_prep_weight), Kpad = K rounded up to
 * the kernel's K tile.  transposed=1 gathers as the data-gradient of a strided
 * conv (x is then dY on the [Hi,Wi] grid of the forward OUTPUT, the result lives
 * on the forward INPUT grid [Ho,Wo]); with a dgrad-prepared weight this is
 * conv2d's dgrad for any stride.
 * ------------------------------------------------------------------------- */
/* PSG_CONV_SAVE_DACT (forward form): `preact` receives the epilogue's DERIVATIVE d = act'(u) * mask/(1-p) instead of
 *   the pre-activation u, so that backward multiplies by a loaded value instead of re-evaluating erf/exp and the mask.
 * PSG_CONV_DACT_MUL (backward form): `dact_u` holds that saved derivative: value *= dact_u (no act', drop_p must be 0).
 * PSG_CONV_GENERIC_EPILOGUE: run the launch through the run-time (one-step) epilogue instead of its specialised copy -
 * same results bit for bit; exists so that tests can hold the two against each other in one process. */
enum psg_conv_flags { PSG_CONV_SAVE_DACT = 1, PSG_CONV_DACT_MUL = 2, PSG_CONV_GENERIC_EPILOGUE = 4 };
typedef struct psg_conv_desc {
    int32_t dtype;               /* psg_dtype of x, w, y, rowadd, residual, preact, dact_u */
    int32_t B, Hi, Wi, Cin;      /* gather source x: [B, Hi, Wi, Cin] */
    int32_t Ho, Wo, Cout;        /* result y: [B, Ho, Wo, Cout] */
    int32_t ksize, stride, pad;  /* 1 (pad 0), 3 (pad 1), or the VAE encoder's 4 (stride 2, pad 1 or 2: forward gather only); stride 1 or 2 */
    int32_t transposed;          /* 0 forward gather, 1 data-gradient gather */
    int32_t act;                 /* psg_act applied to (acc + bias + rowadd) */
    float alpha;                 /* scale of the activated value (gates 0.7/0.8/0.6: unet.py:220,238,250) */
    float drop_p;                /* dropout probability on the activated value (0 = off) */
    int32_t flags;               /* psg_conv_flags (0 = none) */
    uint64_t drop_seed;          /* mask = hash(seed, m*Cout+n) — regenerated in backward */
    int64_t ldx, ldy, ld_rowadd, ld_residual, ld_preact, ld_dact;
    int64_t ldw;                 /* row stride of w in elements; 0 = Kpad (a column slice of a wider prepared
                                    weight is addressed with w + offset and ldw = its Kpad) */
    const void* x;
    const void* w;
    void* y;
    const float* bias;           /* [Cout] fp32 or NULL */
    const void* rowadd;          /* [B, Cout] per-sample add (time/text proj, unet.py:119-124) or NULL */
    const void* residual;        /* [M, Cout] or NULL; may alias y (accumulate) */
    void* preact;                /* optional store of (acc+bias+rowadd) for backward, or NULL */
    const void* dact_u;          /* backward form: when non-NULL, `act` is NOT applied; the value is multiplied by
                                    act'(dact_u[m,n]) (saved pre-activation) and by the same dropout mask.  Used by
                                    the FFN backward: the data gradient of its second Linear comes out already
                                    multiplied by GELU'(u) * mask of the first.  Mutually exclusive with residual. */
    void* ws;                    /* optional split-K workspace (16-byte aligned) of ws_bytes bytes, or NULL: with at least */
    int64_t ws_bytes;            /* psg_conv_fwd_workspace_bytes(d) bytes a launch whose tile grid would leave most of the
                                    chip idle (small M: the sampler of improved_diffusion_trainer.py:534-567 at 64 samples,
                                    training batches of 2-4) shares each tile's K axis among several workgroups - fp32
                                    partial tiles, summed in fixed order by a second kernel that applies the epilogue */
} psg_conv_desc;
int psg_conv_fwd(const psg_conv_desc* d, psg_stream_t stream);
int64_t psg_conv_fwd_workspace_bytes(const psg_conv_desc* d);   /* 0: the launch would not be split; < 0: bad descriptor */
/* Diagnostics: psg_conv_fwd runs pointwise layers (1x1 / Linear, bf16, whole 128 x 128 tiles, >= 1.5 tiles per resident
 * workgroup slot) on the persistent kernel of csrc/conv_pw.hip - same arithmetic, same bits.  psg_conv_set_pw(0) sends them
 * to the per-tile kernel instead (A/B runs, the bitwise test); psg_conv_pw_launches counts launches the persistent kernel took. */
int psg_conv_set_pw(int on);
int64_t psg_conv_pw_launches(void);

/* Weight gradient — convolution_backward's wgrad for the same layers.
 * dw (fp32) = sum_m dy[m,co] * x[pix(m,kh,kw), ci], stored in the parameter's own memory order:
 * PSG_W_OIHW dw[co][ci][kh][kw] (torch contiguous) or PSG_W_OHWI dw[co][kh][kw][ci]
 * (torch channels_last — the kernel's native order: no permute pass, and when the plan has a
 * single K split the tiles are written straight into dw).  Overwritten, or accumulated if
 * accumulate != 0.  Split-K over pixels with a deterministic fixed-order second pass.
 * ws: >= psg_conv_wgrad_workspace_bytes(...) (0 bytes / NULL allowed when that returns 0). */
enum psg_w_layout { PSG_W_OIHW = 0, PSG_W_OHWI = 1 };
typedef struct psg_wgrad_desc {
    int32_t dtype;
    int32_t B, Hi, Wi, Cin, Ho, Wo, Cout, ksize, stride, pad;
    int32_t accumulate;
    int32_t dw_layout;  /* psg_w_layout of dw */
    int32_t accumulate_bias;   /* like accumulate, for dbias */
    float scale;               /* dw, dbias = scale * (...) — 0 is read as 1 (folds a constant output gate, e.g. the
                                  0.7 / 0.8 attention gates of unet.py:220,238, into the gradient instead of a pass over dy) */
    int32_t reserved1;
    int64_t ldx, lddy;
    const void* x;    /* forward input  [B,Hi,Wi,Cin] */
    const void* dy;   /* output grad    [B,Ho,Wo,Cout] */
    float* dw;        /* Cout*Cin*k*k fp32 in dw_layout order */
    float* dbias;     /* optional [Cout] fp32: the layer's bias gradient sum_m dy[m,co], produced by the SAME launch
                         (the dY tiles are already in LDS: one extra MFMA against a ones operand per tile row) */
    void* ws;
    int64_t ws_bytes;
} psg_wgrad_desc;
int psg_conv_wgrad(const psg_wgrad_desc* d, psg_stream_t stream);
int64_t psg_conv_wgrad_workspace_bytes(const psg_wgrad_desc* d);

/* Master weight (w_dtype PSG_F32, or PSG_BF16 for the AdamW shadow copy; w_layout PSG_W_OIHW or PSG_W_OHWI memory
 * order; a bf16 source must be OHWI or 1x1) -> prepared forward weight wf [O][Kpad] with k=(kh,kw,ci) and prepared
 * data-gradient weight wd [I][Kpad'] with k=(kh,kw,co) (either may be NULL); zero K padding.  Kpad: psg_kpad(). */
int psg_prep_weight(const void* w, int w_dtype, int w_layout, void* wf, void* wd, int O, int I, int ksize, int dtype,
                    psg_stream_t stream);
int64_t psg_kpad(int64_t K, int dtype);

/* Column sums: out[g, c] = sum_{r<R} a[(g*R + r), c] for g<groups.  groups=1 gives a bias
 * gradient (fp32 out); groups=B, R=HW gives the gradient of the per-sample rowadd.  out_dtype
 * selects fp32 or bf16 output; accumulate adds into out (fp32 only).  Deterministic. */
int psg_colsum(const void* a, int64_t lda, void* out, int64_t ld_out, int64_t R, int groups, int cols,
               int dtype, int out_dtype, int accumulate, void* ws, int64_t ws_bytes, psg_stream_t stream);
int64_t psg_colsum_workspace_bytes(int64_t R, int groups, int cols);

/* Backward of the psg_conv_fwd epilogue: g[m,n] = dy[m,n] * alpha * mask(seed, m*cols+n)/(1-p) * act'(u[m,n])
 * (u = saved pre-activation, may be NULL when act == PSG_ACT_NONE). */
int psg_epilogue_bwd(const void* dy, int64_t lddy, const void* u, int64_t ldu, void* g, int64_t ldg, int64_t rows,
                     int cols, int act, float alpha, float drop_p, uint64_t seed, int dtype, psg_stream_t stream);
/* dropout backward helper: y = x * mask(seed, idx) * scale (same hash as psg_conv_fwd). */
int psg_dropout_apply(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int cols,
                      float p, uint64_t seed, float scale, int dtype, psg_stream_t stream);

/* ---------------------------------------------------------------------------
 * Multi-head attention core — the scaled-dot-product inside
 * nn.MultiheadAttention (unet.py:160-173,217,235; torch/nn/functional.py
 * multi_head_attention_forward): P = softmax((q/sqrt(d)) k^T), o = drop(P) v.
 * q rows: token (b,l) at q + (b*L+l)*ldq + h*d; k, v: (b,s) at + (b*S+s)*ldk.
 * lse: fp32 [B, heads, L] (log-sum-exp of the scaled scores) saved for backward.
 * ------------------------------------------------------------------------- */
int psg_attn_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                 void* o, int64_t ldo, float* lse, int B, int heads, int L, int S, int d, float scale,
                 float drop_p, uint64_t seed, int dtype, psg_stream_t stream);
/* Diagnostic: launches of psg_attn_fwd / psg_attn_bwd served so far by the bf16 MFMA kernels (head_dim 16 / 32 / 64 / 80 /
 * 160 / 320, 16-byte aligned rows), by the VALU kernels (other shapes) and by the exact-fp32 MFMA kernels (fp32, head_dim
 * 16 / 32 / 64 / 80 / 160 while K/V - and Q/dO for backward - fit LDS).  All paths draw the same dropout mask. */
int psg_attn_path_counts(int64_t* mfma, int64_t* valu, int64_t* mfma_f32);
/* Diagnostic: which kernel families later psg_attn_fwd / psg_attn_bwd calls may take: bit 0 the bf16 MFMA kernels, bit 1
 * the exact-fp32 MFMA kernels; a cleared bit sends those launches to the VALU kernels (default 3).  The parity tests pin
 * each family against the same reference this way (a process-wide switch: not for concurrent use). */
int psg_attn_set_paths(int allow_mask);
/* delta: fp32 [B, heads, L] scratch.  dq/dk/dv have the strides of q/k/v. */
int psg_attn_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                 const void* o, int64_t ldo, const void* dout, int64_t lddo, const float* lse,
                 float* delta, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                 int B, int heads, int L, int S, int d, float scale, float drop_p, uint64_t seed,
                 int dtype, psg_stream_t stream);

/* ---------------------------------------------------------------------------
 * Optimizer side — improved_diffusion_trainer.py:399-413
 * ------------------------------------------------------------------------- */
/* out[0] (+)= sum(g^2) over n fp32 values; deterministic.  Replaces the 478 .item() syncs (:399-404). */
int psg_sumsq_f32(const float* g, int64_t n, float* out, int accumulate, void* ws, psg_stream_t stream);
/* clip_grad_norm_ (:410) fused with AdamW (:277-283,412): coef = min(1, max_norm/(sqrt(*normsq)+1e-6))
 * read on the device (normsq may be NULL = no clipping); decoupled weight decay; step is 1-based.
 * skip_flag (may be NULL): when (*skip_flag & PSG_FLAG_SKIP_MASK) != 0 the update is skipped (NaN batch, :353-393).
 * shadow_bf16 (may be NULL): bf16 copy of the updated parameters written in the same pass, element i at
 * shadow_bf16[i] — with OHWI master weights this IS the prepared forward weight of the next step. */
int psg_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, const float* normsq,
                  float max_norm, const int32_t* skip_flag, void* shadow_bf16, psg_stream_t stream);
/* The same update with the step count and the learning-rate schedule ON THE DEVICE, so that a skipped batch
 * advances neither (the reference `continue`s before optimizer.step() / scheduler.step() / global_step += 1,
 * :353-393,412-418) without a host read of the flag: step = *step_dev + 1 (1-based, bias correction computed in
 * the kernel), lr = lr_table[k], beta1 = beta1_table[k] with k = min(step - 1, sched_len - 1) (entry k = the values
 * the reference's scheduler holds after k scheduler.step() calls; OneCycleLR (:313-319) cycles Adam's beta1 along
 * with the lr; beta1_table may be NULL = constant beta1), and *step_dev += 1 after the update unless skipped. */
int psg_adamw_dev_f32(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_table,
                      const float* beta1_table, int sched_len, float beta1, float beta2, float eps,
                      float weight_decay, int32_t* step_dev, const float* normsq, float max_norm,
                      const int32_t* skip_flag, void* shadow_bf16, psg_stream_t stream);
/* g *= min(1, max_norm/(sqrt(*normsq)+1e-6)) — plain clip for callers that keep torch.optim. */
int psg_clip_scale_f32(float* g, int64_t n, const float* normsq, float max_norm, psg_stream_t stream);

/* ---------------------------------------------------------------------------
 * Measurement hooks (bench.py): between begin and end every launch of the matrix / attention /
 * GroupNorm kernel families is bracketed by hipEvents on its stream.  end() drains once and
 * returns, per family k < nkinds (0 conv fwd gather, 1 conv data-gradient gather, 2 wgrad,
 * 3 attention, 4 GroupNorm): summed kernel milliseconds, summed algorithmic work (FLOPs for
 * 0-3, bytes for 4) and launch count.  Off by default: no cost on the product path.
 * ------------------------------------------------------------------------- */
/* Planning input: the number of CUs the tile / split-K choosers of psg_conv_fwd and psg_conv_wgrad may count on (default
 * 256 = the whole MI355X; 0 restores it).  A data-parallel step that overlaps the RCCL all-reduce with backward sets it to
 * 256 minus the CUs the collective's channels occupy, so a launch planned as ONE round of workgroups does not become two. */
int psg_set_available_cus(int n);
/* ... r > 0: only launches that would take at most r rounds of workgroups on the whole chip plan around the reserve (a
 * one-round grid doubles when CUs are taken; a many-round grid loses the CU share either way); 0: every launch (default). */
int psg_set_reserve_rounds(int r);
/* Measurement helpers (tools/contention.py): a HIP stream restricted to n_cus CUs (hipExtStreamCreateWithCUMask; the CUs
 * taken out are spread evenly over the 8 XCDs), to rehearse the step with part of the chip occupied by a collective. */
int psg_stream_create_cu_mask(int n_cus, psg_stream_t* stream);
int psg_stream_destroy(psg_stream_t stream);
int psg_profile_begin(void);
int psg_profile_end(double* ms, double* work, int64_t* launches, int nkinds);
/* Algorithmic operand bytes (each input / output / weight element once) summed per family over the launches the last
 * psg_profile_end() accounted for: printed beside the PMC traffic so the re-read ratio needs no derivation. */
int psg_profile_bytes(double* bytes, int nkinds);

#ifdef __cplusplus
}
#endif
#endif /* PSG_HIP_H */
