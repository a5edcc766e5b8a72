/* TEST INFRASTRUCTURE ONLY — plain-C restatement of the bit-exact elementwise
 * pieces of the hot path.  Built by oracle/Makefile with -ffp-contract=off so
 * that every multiply and add rounds separately, exactly as the reference's
 * PyTorch CPU elementwise kernels do.  Never linked or loaded by the product.
 *
 * Reference lines (relative to the reference root):
 *   add_noise        src/training/improved_diffusion_trainer.py:55-65 (+ clamp :363)
 *   ddpm update      src/training/improved_diffusion_trainer.py:554-567
 *   SmoothL1(beta)   src/training/improved_diffusion_trainer.py:300,388
 */
#include <math.h>
#include <stdint.h>

/* noisy[b,i] = tabA[t[b]] * clamp(x0[b,i]) + tabB[t[b]] * noise[b,i]; returns 1 if
 * any output is NaN/Inf, in which case the caller (like :61-63) must take the
 * fallback x0 + 0.1*noise (written to `fallback` when non-NULL). */
int oracle_noise_add_f32(const float* x0, const float* noise, const int64_t* t,
                         const float* tabA, const float* tabB, float* out, float* fallback,
                         int64_t B, int64_t chw, int do_clamp)
{
    int bad = 0;
    for (int64_t b = 0; b < B; ++b) {
        const float a = tabA[t[b]], c = tabB[t[b]];
        for (int64_t i = 0; i < chw; ++i) {
            float x = x0[b * chw + i];
            if (do_clamp && x == x) x = fminf(fmaxf(x, -3.0f), 3.0f);   /* torch.clamp(latent,-3,3) :363 - NaN stays NaN */
            const float p = a * x;
            const float q = c * noise[b * chw + i];
            const float r = p + q;
            out[b * chw + i] = r;
            if (isnan(r) || isinf(r)) bad = 1;
            if (fallback) {
                const float s = 0.1f * noise[b * chw + i];
                fallback[b * chw + i] = x + s;
            }
        }
    }
    return bad;
}

/* x <- c1 * (x - c2*eps) [+ sigma*z]   (:557, :563/:567) */
void oracle_ddpm_update_f32(float* x, const float* eps, const float* z, float c1, float c2,
                            float sigma, int add_noise, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        const float m = c2 * eps[i];
        const float d = x[i] - m;
        float r = c1 * d;
        if (add_noise) { const float s = sigma * z[i]; r = r + s; }
        x[i] = r;
    }
}

/* mean SmoothL1 with beta; grad[i] = dL/dpred[i].  Loss accumulated in double
 * (the reference's fp32 mean differs from this by reduction order only). */
double oracle_smooth_l1_f32(const float* pred, const float* target, float* grad, float beta, int64_t n)
{
    double acc = 0.0;
    const float inv_n = 1.0f / (float)n;
    for (int64_t i = 0; i < n; ++i) {
        const float d = pred[i] - target[i];
        const float ad = fabsf(d);
        if (ad < beta) {
            acc += 0.5 * (double)d * (double)d / (double)beta;
            if (grad) grad[i] = d / beta * inv_n;
        } else {
            acc += (double)ad - 0.5 * (double)beta;
            if (grad) grad[i] = (d > 0.f ? 1.f : -1.f) * inv_n;
        }
    }
    return acc / (double)n;
}
