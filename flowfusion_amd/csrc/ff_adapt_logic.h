// ff_adapt_logic.h -- the arithmetic of the device-side adaptive step controller, shared by the gfx950 kernels
// (ff_adaptive.hip) and the host entry points the CPU tests call (ff_api.cpp: ff_adapt_host_row / _transition).
//
// What is restated here, and from where:
//   * torchdiffeq (>=0.2.5,<0.3.0; not in /root/reference, parity unpinned -- DESIGN.md section 6) rk_common.py:
//     `_select_initial_step`, the accept / reject rule of `_adaptive_step`, `_optimal_step_size`, the min_step /
//     max_step clamps and the "underflow in dt" / "non-finite values" / "max_num_steps" assertions, exactly as
//     flowfusion_amd/adaptive.py (the host controller, which stays for module right-hand sides and CPU emulation)
//     states them: time and step sizes in float64, everything that touches the state in fp32;
//   * the time-dependent part of the reference's right-hand side in the reference's fp32 operation order:
//     ScoreModel.ode_drift flowfusion/diffusion.py:276-278 with VESDE :818-1003 (sigma :889, diffusion :905-930),
//     VPSDE :1006-1180 (beta :1098, drift :1131, diffusion :1150, marginal_prob_scalars :1060-1090), SUBVPSDE
//     :1183-1366 (diffusion :1316-1340), score / no_sigma :268-272; the Gaussian-Fourier features pushed through the
//     first layer, MLP.forward :109-113; the flows' raw time column, flowfusion/flow.py:112-118, 583-586.
//     (flowfusion_amd/diffusion.py `_schedule_on_host` and flow.py `_schedule` are the torch statements of the same.)
// Transcendentals come from libm on the host and from the device library on the GPU: they agree to an ulp or two, not
// bit for bit.
#pragma once
#include <math.h>
#include <stdint.h>
#include "flowfusion_amd.h"
#include "ff_layout.h"

namespace ff {
namespace adapt {

constexpr double kSafety = 0.9, kIFactor = 10.0, kDFactor = 0.2;      // torchdiffeq rk_common defaults

enum Phase : int {
    kPhaseInit0 = 0,   // write the row of f(t0, y)
    kPhaseStashD0 = 5, // d0 = max(norms): the scaled norm of the state components (first launch of them)
    kPhaseStashMore = 6, // ... folded with a further launch's components (a tuple state with more than FF_NORM_TERMS)
    kPhaseInit1 = 1,   // d1 = max(norms) -> h0, row of f(t0 + h0, y + h0 f0)
    kPhaseInit2 = 2,   // d2 -> first step, rows of the first attempt
    kPhaseStep = 3,    // error ratio -> accept / reject, next step, rows of the next attempt
    kPhaseFirst = 4,   // options["first_step"]: rows of the first attempt
};

// what the controller must write after a transition
enum Rows : int { kRowsNone = 0, kRowsDerivAtH0 = 1, kRowsAttempt = 2 };

// ---- schedule scalars (a, b) of  rhs = a y + b NET  at real time t, before the reversal sign ------------------------
FF_HD void schedule_ab(const ff_adapt_config& c, float t, float* a_out, float* b_out)
{
#pragma clang fp contract(off)
    if (c.sched == FF_SCHED_FLOW) { *a_out = 0.f; *b_out = 1.f; return; }
    float a, g, sigma;
    if (c.sched == FF_SCHED_VE) {
        const float smin = (float)c.p[0], smax = (float)c.p[1], T = (float)c.p[2];
        sigma = smin * powf(smax / smin, t / T);                                   // VESDE.sigma
        g = sigma * sqrtf(2.f * (logf(smax) - logf(smin)) / T);                    // VESDE.diffusion
        a = 0.f;                                                                   // VESDE.drift = 0
    } else {
        const float bmin = (float)c.p[0], T = (float)c.p[2];
        const float dbeta = (float)(c.p[1] - c.p[0]);                              // (beta_max - beta_min): Python floats
        const float beta = bmin + dbeta * (t / T);                                 // VPSDE.beta
        a = -0.5f * beta;                                                          // drift = -0.5 beta x
        const float lc = (float)(0.5 * (c.p[1] - c.p[0])) * (t * t) / T + bmin * t;    // _log_coeff
        if (c.sched == FF_SCHED_VP) {
            g = sqrtf(beta);
            sigma = sqrtf(1.0f - expf(-lc));
        } else {
            const float decay = expf((float)(-2.0 * c.p[0]) * t - dbeta * (t * t) / T);
            g = sqrtf(beta * (1.0f - decay));
            sigma = 1.0f - expf(-lc);
        }
    }
    float b = -(0.5f * (g * g));
    if (!c.no_sigma) b = b / sigma;
    *a_out = a;
    *b_out = b;
}

// ---- first-layer time part c1[h] at real time t ----------------------------------------------------------------------
// time feature k < n_tcols: score networks [sin(arg_e) | cos(arg_e)], arg_e = ((t W_e) 2) pi (MLP.forward :109-110);
// flows: t itself (flow.py:112-115)
FF_HD float time_feature(const ff_adapt_config& c, float t, int k)
{
#pragma clang fp contract(off)
    if (c.sched == FF_SCHED_FLOW) return t;
    const int e = k < c.n_emb ? k : k - c.n_emb;
    const float arg = t * c.emb_w[e] * 2.f * c.pi;
    return k < c.n_emb ? sinf(arg) : cosf(arg);
}

// c1[h] = sum_k feat[k] w0t[h][k] + b0[h]  (products rounded, then summed in column order, then the bias)
FF_HD float c1_from_features(const ff_adapt_config& c, const float* feat, int h)
{
#pragma clang fp contract(off)
    const float* w = c.w0t + (size_t)h * c.n_tcols;
    if (c.sched == FF_SCHED_FLOW) return feat[0] * w[0] + c.b0[h];
    float acc = 0.f;
    for (int k = 0; k < c.n_tcols; ++k) acc = acc + feat[k] * w[k];
    return acc + c.b0[h];
}

// Python's max(list) over floats (a NaN wins only from the first position)
FF_HD double py_max(const float* v, int n)
{
    double m = (double)v[0];
    for (int i = 1; i < n; ++i)
        if ((double)v[i] > m) m = (double)v[i];
    return m;
}

FF_HD double clamp_step(const ff_adapt_config& c, double dt)
{
    if (dt != dt) return dt;
    const double lo = dt > c.min_step ? dt : c.min_step;          // max(dt, min_step)
    return lo < c.max_step ? lo : c.max_step;                     // min(.., max_step)
}

FF_HD double optimal_step(const ff_adapt_config& c, double last_step, double ratio)
{
    if (ratio != ratio) return ratio;                             // NaN propagates; the next attempt underflows
    if (ratio == 0.0) return last_step * kIFactor;
    const double dfactor = ratio < 1.0 ? 1.0 : kDFactor;
    const double want = kSafety / pow(ratio, 1.0 / (double)c.order);
    const double floor_ = want > dfactor ? want : dfactor;        // max(want, dfactor)
    const double factor = kIFactor < floor_ ? kIFactor : floor_;  // min(ifactor, ..)
    return last_step * factor;
}

// what every attempt starts with (adaptive.py integrate: the head of the loop).  Returns kRowsAttempt or stops the solve.
FF_HD int begin_attempt(const ff_adapt_config& c, ff_adapt_state& s)
{
    if (s.n_steps >= c.max_num_steps) { s.error = FF_ADAPT_ERR_MAXSTEPS; s.active = 0; return kRowsNone; }
    s.dt = clamp_step(c, s.dt);
    if (!(s.t + s.dt > s.t)) { s.error = FF_ADAPT_ERR_UNDERFLOW; s.active = 0; return kRowsNone; }   // also dt = NaN
    return kRowsAttempt;
}

// One controller transition.  `norms` = what the reduction produced for this phase.  Returns the rows to write.
FF_HD int transition(const ff_adapt_config& c, ff_adapt_state& s, int phase, const float* norms, int n_norms)
{
    if (phase == kPhaseStashD0) { s.d0 = py_max(norms, n_norms); return kRowsNone; }
    if (phase == kPhaseStashMore) {
        for (int i = 0; i < n_norms; ++i)
            if ((double)norms[i] > s.d0) s.d0 = (double)norms[i];
        return kRowsNone;
    }
    if (phase == kPhaseInit1) {
        const double d0 = s.d0, d1 = py_max(norms, n_norms);
        s.d1 = d1;
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        s.h0 = (double)(float)fabs(h0);
        return kRowsDerivAtH0;
    }
    if (phase == kPhaseInit2) {
        const double d2 = fabs(py_max(norms, n_norms) / s.h0);
        double h1;
        if (s.d1 <= 1e-15 && d2 <= 1e-15) h1 = (1e-6 > s.h0 * 1e-3) ? 1e-6 : s.h0 * 1e-3;
        else h1 = pow(0.01 / (s.d1 > d2 ? s.d1 : d2), 1.0 / (double)c.order);
        const double cap = 100.0 * s.h0;
        s.dt = cap < fabs(h1) ? cap : fabs(h1);
        return begin_attempt(c, s);
    }
    if (phase == kPhaseFirst) {
        s.dt = c.first_step;
        return begin_attempt(c, s);
    }
    // kPhaseStep: norms[0] = error ratio, norms[1] != 0 = the proposal holds a non-finite value
    const double ratio = (double)norms[0];
    const bool bad = norms[1] != 0.f;
    s.last_ratio = norms[0];
    s.n_attempts += 1;
    bool accept = ratio <= 1.0;
    if (s.dt > c.max_step) accept = false;
    if (s.dt <= c.min_step) accept = true;
    s.commit = 0;
    if (accept) {
        if (bad) { s.error = FF_ADAPT_ERR_NONFINITE; s.active = 0; return kRowsNone; }
        s.n_accepted += 1;
        s.t_prev = s.t;
        s.dt_prev = s.dt;
        s.t = s.t + s.dt;
    }
    s.dt = clamp_step(c, optimal_step(c, s.dt, ratio));
    s.n_steps += 1;
    if (accept && !(s.t_end > s.t)) { s.done = 1; s.active = 0; return kRowsNone; }   // the buffers keep this step
    if (accept) s.commit = 1;
    const int rows = begin_attempt(c, s);
    if (rows == kRowsNone) s.commit = 0;          // stopped: leave the buffers as they are
    return rows;
}

// ---- the words of one evaluation row / tail row ---------------------------------------------------------------------
// stage i (1 .. n_stages-1) of an attempt from (t, dt): fp32 stage time, as adaptive.py `_attempt` computes it
FF_HD float stage_time(const ff_adapt_config& c, double t, double dt, int i)
{
#pragma clang fp contract(off)
    const float t0f = (float)t, dtf = (float)dt, t1f = (float)(t + dt);
    const float al = c.alpha[i - 1];
    return al == 1.0f ? t1f : t0f + al * dtf;
}

} // namespace adapt
} // namespace ff
