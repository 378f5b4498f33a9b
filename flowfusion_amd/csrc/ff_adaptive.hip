// ff_adaptive.hip -- adaptive embedded Runge-Kutta solves with the step control on the device (gfx950).
//
// The reference's default solver at every call site is torchdiffeq's adaptive dopri5 (flowfusion/diffusion.py:572,
// 631-639; 649, 744-752; 762; flowfusion/flow.py:299-303, 313, 371-382): one step size for the whole batch, chosen from
// a norm over the entire state.  Rounds 1-2 kept that decision on the host: one fused launch per attempted step, then a
// reduction, a read-back, a few hundred microseconds of Python (schedule, table, upload) before the next launch -- half
// of a notebook-scale call.  Here the decision lives on the device:
//
//     attempt (ff_mlp_ode_launch, gated)  ->  adapt_control_kernel  ->  adapt_commit_kernel        x n, no host in between
//
//   adapt_control_kernel   the deterministic scaled-RMS reduction of ff_norm.h over (error estimate, y0, y1[, lp]) plus the
//                          finiteness check of y1; the block that arrives last is the controller: accept / reject, next
//                          step size (ff_adapt_logic.h, float64 time), then its 256 threads write the next attempt's
//                          evaluation rows -- stage times, a_e / b_e of the SDE, c1_e = W1[:, time] feat(t_e) + b1 -- into
//                          the table the next launch reads.  Roofline: HBM (16 B per element of the state, one pass).
//   adapt_commit_kernel    an accepted step's proposal becomes the current state: (y, f0, lp, fl0) <- (y1, f1, lp1, fl1);
//                          a streaming copy, a no-op after a rejection, after the last step and after an error.
//   adapt_finish_kernel    torchdiffeq's fourth-order dense output of the last step at t_end (`_interp_fit` /
//                          `_interp_evaluate`), one pass.
//   adapt_sum_kernel       FF_MODE_EXACT with more unit tangents than one launch carries: adds the passes' divergences.
//   adapt_lp_combine_kernel  Hutch++ / XTrace models (ff_adapt_buffers.est_kind; reference flowfusion/diffusion.py:336-481):
//                          the attempt records every row's Jacobian, ff_trace_estimate (ff_trace.hip) turns them into
//                          divergence estimates, and this kernel takes the attempt's linear combinations of them.
//
// Launches enqueued behind the end of the solve (the host enqueues chunks and reads the state once per chunk) find
// ff_adapt_state.active == 0 and return at once; so does the fused kernel (ff_ode_args.gate).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "flowfusion_amd.h"
#include "ff_layout.h"
#include "ff_norm.h"
#include "ff_adapt_logic.h"

namespace ff {

constexpr int kMaxTimeCols = 64;       // time features of the first layer the controller keeps in LDS

struct ControlArgs {
    NormArgs norm;
    ff_adapt_config cfg;
    ff_adapt_state* state;
    float* etab;
    int etab_stride;       // FF_ROW_HDR + plan width
    int phase;
    double t0, t_end;      // kPhaseInit0 only
    int stage;             // 0: norms and controller in one launch; several shards of one batch (ff_adapt_buffers.exchange):
    double* gsums;         // 1 = reduce into gsums, 2 = controller from gsums after they were summed over the ranks
};

// words of one evaluation row of the table for real time `t_real`; `feat` = the row's time features (LDS)
__device__ __forceinline__ void write_eval_row(const ff_adapt_config& c, float* row, int stride, const float* ab, int slot,
                                               const float* cin8, const float* feat)
{
#pragma clang fp contract(off)
    for (int i = threadIdx.x; i < stride; i += blockDim.x) {
        float v = 0.f;
        if (i < 2) {
            v = ab[i];
        } else if (i == 4) {
            v = __builtin_bit_cast(float, slot);
        } else if (i >= 8 && i < 16) {
            v = cin8[i - 8];
        } else if (i >= FF_ROW_HDR) {
            const int h = i - FF_ROW_HDR;
            v = h < c.h_real ? adapt::c1_from_features(c, feat, h) : 0.f;
        }
        row[i] = v;
    }
}

__device__ __forceinline__ void write_tail_row(float* row, int stride, const float* cin8, const float* cout8, uint32_t use_y)
{
    for (int i = threadIdx.x; i < stride; i += blockDim.x) {
        float v = 0.f;
        if (i == 3) v = __builtin_bit_cast(float, use_y);
        else if (i >= 8 && i < 16) v = cin8[i - 8];
        else if (i >= 16 && i < 24) v = cout8[i - 16];
        row[i] = v;
    }
}

__global__ __launch_bounds__(256) void adapt_control_kernel(const ControlArgs a)
{
    __shared__ double sh[4];
    __shared__ bool last;
    __shared__ float res[FF_NORM_TERMS + 1];
    __shared__ int rows_kind;
    __shared__ double s_t, s_dt, s_h0;
    __shared__ float row_t[FF_MAX_SLOTS];                       // solver-time of each evaluation row
    __shared__ float row_cin[FF_MAX_SLOTS][8];
    __shared__ float row_ab[FF_MAX_SLOTS][2];                   // sign * a_e, sign * b_e of each evaluation row
    __shared__ float feat[FF_MAX_SLOTS][kMaxTimeCols];
    __shared__ float tail[4][8];

    // launches enqueued behind the end of the solve do nothing (uniform: nobody writes `active` before the last block)
    if (a.phase == adapt::kPhaseStep && *(const volatile int*)&a.state->active == 0) return;

    if (a.stage == 2) {
        // the sums of squares of every rank's rows have met: the norms are over the whole batch
        if (threadIdx.x == 0) {
            for (int t = 0; t < FF_NORM_TERMS; ++t) {
                const double cnt = a.gsums[FF_NORM_TERMS + 1 + t];
                res[t] = cnt > 0.0 ? (float)sqrt(a.gsums[t] / cnt) : 0.f;
            }
            res[FF_NORM_TERMS] = a.gsums[FF_NORM_TERMS] > 0.0 ? 1.f : 0.f;
        }
        __syncthreads();
    } else {
        bool is_last = true;
        if (a.norm.n_terms > 0 || a.norm.n_check > 0) is_last = scaled_rms_reduce(a.norm, sh, &last, res);
        if (!is_last || a.stage == 1) return;
    }

    const ff_adapt_config& c = a.cfg;
    if (threadIdx.x == 0) {
        ff_adapt_state s = *a.state;
        int rows = adapt::kRowsNone;
        if (a.phase == adapt::kPhaseInit0) {
            memset(&s, 0, sizeof(s));
            s.t = a.t0; s.t_prev = a.t0; s.t_end = a.t_end; s.active = 1;
            rows = -1;                                         // the row of f(t0, y)
        } else if (a.phase == adapt::kPhaseStep) {
            float nr[2] = {(float)0, res[FF_NORM_TERMS]};
            // Python's max over the terms' norms, as a float (the host controller reads them back as floats)
            float m = res[0];
            for (int i = 1; i < a.norm.n_terms; ++i)
                if (res[i] > m) m = res[i];
            nr[0] = m;
            rows = adapt::transition(c, s, a.phase, nr, 2);
        } else {
            rows = adapt::transition(c, s, a.phase, res, a.norm.n_terms);
        }
        *a.state = s;
        rows_kind = rows;
        s_t = s.t; s_dt = s.dt; s_h0 = s.h0;
    }
    __syncthreads();
    const int kind = rows_kind;
    if (kind == adapt::kRowsNone) return;

    // ---- the next launch's evaluation rows ---------------------------------------------------------------------------
    const int S = c.n_stages;
    const int n_rows = kind == adapt::kRowsAttempt ? S - 1 : 1;
    if ((int)threadIdx.x < n_rows) {
#pragma clang fp contract(off)
        const int r = threadIdx.x;
        float ts;
        for (int j = 0; j < 8; ++j) row_cin[r][j] = 0.f;
        if (kind == adapt::kRowsAttempt) {
            ts = adapt::stage_time(c, s_t, s_dt, r + 1);
            const float dtf = (float)s_dt;
            for (int j = 0; j < 8; ++j) row_cin[r][j] = c.beta[r][j] * dtf;
        } else if (kind == adapt::kRowsDerivAtH0) {
            ts = (float)((double)(float)s_t + s_h0);            // float(_f32(t0)) + h0, then to fp32
            row_cin[r][0] = (float)s_h0;
        } else {
            ts = (float)s_t;
        }
        row_t[r] = ts;
        // the schedule scalars of the row (one thread per row: the transcendentals of six rows run side by side)
        float sa, sb;
        adapt::schedule_ab(c, c.sign * ts, &sa, &sb);
        row_ab[r][0] = c.sign * sa;
        row_ab[r][1] = c.sign * sb;
    }
    if (threadIdx.x < 32) {                                     // tail coefficients: 4 x 8 words
#pragma clang fp contract(off)
        const int j = threadIdx.x >> 3, k = threadIdx.x & 7;
        float v = 0.f;
        if (kind == adapt::kRowsAttempt) {
            const float dtf = (float)s_dt;
            if (j == 0) v = dtf * c.c_sol[k];                   // y1    = y + dt k . c_sol
            else if (j == 1) v = k == S - 1 ? 1.f : 0.f;        // f1    = the last stage
            else if (j == 2) v = dtf * c.c_mid[k];              // y_mid = y + dt k . c_mid
            else v = dtf * c.c_err[k];                          // err   = dt k . c_error
        } else if (j == 0) {
            const int slot = kind == adapt::kRowsDerivAtH0 ? 1 : 0;
            v = k == slot ? 1.f : 0.f;                          // aux_0 = k[slot]
        }
        tail[j][k] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_rows * c.n_tcols; i += blockDim.x) {
        const int r = i / c.n_tcols, k = i - r * c.n_tcols;
        feat[r][k] = adapt::time_feature(c, c.sign * row_t[r], k);
    }
    __syncthreads();
    for (int r = 0; r < n_rows; ++r) {
        const int slot = kind == adapt::kRowsAttempt ? r + 1 : (kind == adapt::kRowsDerivAtH0 ? 1 : 0);
        write_eval_row(c, a.etab + (size_t)r * a.etab_stride, a.etab_stride, row_ab[r], slot, row_cin[r], feat[r]);
    }
    write_tail_row(a.etab + (size_t)n_rows * a.etab_stride, a.etab_stride, tail[0], tail[1],
                   kind == adapt::kRowsAttempt ? 0b0101u : 0u);
    write_tail_row(a.etab + (size_t)(n_rows + 1) * a.etab_stride, a.etab_stride, tail[2], tail[3], 0u);
}

struct CommitArgs {
    const ff_adapt_state* state;
    float* dst[4];
    const float* src[4];
    long long n[4];
};

__global__ __launch_bounds__(256) void adapt_commit_kernel(const CommitArgs a)
{
    if (*(const volatile int*)&a.state->commit == 0) return;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!a.dst[j]) continue;
        const bool vec = ((((uintptr_t)a.dst[j]) | ((uintptr_t)a.src[j])) & 15) == 0;
        const long long n4 = vec ? a.n[j] / 4 : 0;
        for (long long i = tid; i < n4; i += stride) ((f32x4a*)a.dst[j])[i] = ((const f32x4a*)a.src[j])[i];
        for (long long i = 4 * n4 + tid; i < a.n[j]; i += stride) a.dst[j][i] = a.src[j][i];
    }
}

struct FinishArgs {
    const ff_adapt_state* state;
    const float* y0[2]; const float* y1[2]; const float* ymid[2]; const float* f0[2]; const float* f1[2];
    float* out[2];
    long long n[2];
};

// torchdiffeq `_interp_fit` + `_interp_evaluate` (adaptive.py `_fit_eval`), one rounding per torch op
__device__ __forceinline__ float fit_eval(float y0, float y1, float ym, float f0, float f1, float dt, float x)
{
#pragma clang fp contract(off)
    const float a = 2.f * dt * (f1 - f0) - 8.f * (y1 + y0) + 16.f * ym;
    const float b = dt * (5.f * f0 - 3.f * f1) + 18.f * y0 + 14.f * y1 - 32.f * ym;
    const float c = dt * (f1 - 4.f * f0) - 11.f * y0 - 5.f * y1 + 16.f * ym;
    const float d = dt * f0;
    float total = y0 + x * d;
    float xp = x;
    xp = xp * x; total = total + xp * c;
    xp = xp * x; total = total + xp * b;
    xp = xp * x; total = total + xp * a;
    return total;
}

__global__ __launch_bounds__(256) void adapt_finish_kernel(const FinishArgs a)
{
    if (*(const volatile int*)&a.state->done == 0) return;
    const double ta = a.state->t_prev, tb = a.state->t, te = a.state->t_end;
    const float x = (float)((te - ta) / (tb - ta));
    const float dt = (float)a.state->dt_prev;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (!a.out[j]) continue;
        for (long long i = tid; i < a.n[j]; i += stride)
            a.out[j][i] = fit_eval(a.y0[j][i], a.y1[j][i], a.ymid[j][i], a.f0[j][i], a.f1[j][i], dt, x);
    }
}

struct SumArgs {
    const int* gate;
    const float* part;        // [n_passes][rows][n]
    float* out[FF_MAX_AUX];   // rows pointers
    int n_passes, rows;
    long long n;
};

__global__ __launch_bounds__(256) void adapt_sum_kernel(const SumArgs a)
{
    if (a.gate && *(const volatile int*)a.gate == 0) return;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = 0; r < a.rows; ++r) {
        if (!a.out[r]) continue;
        for (long long i = tid; i < a.n; i += stride) {
            float v = a.part[((size_t)0 * FF_MAX_AUX + r) * a.n + i];
            for (int p = 1; p < a.n_passes; ++p) v = v + a.part[((size_t)p * FF_MAX_AUX + r) * a.n + i];   // the host's `aux_lp + olp`
            a.out[r][i] = v;
        }
    }
}

// Divergence slots from ESTIMATES: kl[slot_e] = div[e] for the evaluation rows of the launch (kl[0] = kl1 when the first
// stage came in), then the same combinations the fused kernel's epilogue takes of its own slots:
//     out_j = use_y_j * lp0 + sum_s coef_j[s] * kl[s]     coefficients / use_y bits in the two rows behind the evaluation rows
struct LpCombineArgs {
    const int* gate;
    const float* etab;
    int etab_stride, n_evals, n_aux;
    const float* div;         // [n_evals][n]
    const float* kl1;         // [n] or NULL
    const float* lp0;         // [n] or NULL
    float* out[FF_MAX_AUX];
    long long n;
};

__global__ __launch_bounds__(256) void adapt_lp_combine_kernel(const LpCombineArgs a)
{
    if (a.gate && *(const volatile int*)a.gate == 0) return;
    __shared__ float coef[FF_MAX_AUX][FF_MAX_SLOTS];
    __shared__ int slot_of[FF_MAX_SLOTS];
    __shared__ uint32_t use_y;
    if (threadIdx.x < FF_MAX_AUX * FF_MAX_SLOTS) {
        const int j = threadIdx.x / FF_MAX_SLOTS, s = threadIdx.x % FF_MAX_SLOTS;
        const float* row = a.etab + (size_t)(a.n_evals + (j >> 1)) * a.etab_stride;
        coef[j][s] = row[((j & 1) ? 16 : 8) + s];
    }
    if (threadIdx.x < FF_MAX_SLOTS)
        slot_of[threadIdx.x] = (int)threadIdx.x < a.n_evals ? __builtin_bit_cast(int, a.etab[(size_t)threadIdx.x * a.etab_stride + 4]) : -1;
    if (threadIdx.x == 0) use_y = __builtin_bit_cast(uint32_t, a.etab[(size_t)a.n_evals * a.etab_stride + 3]);
    __syncthreads();
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        float kl[FF_MAX_SLOTS];
#pragma unroll
        for (int s = 0; s < FF_MAX_SLOTS; ++s) kl[s] = 0.f;
        if (a.kl1) kl[0] = a.kl1[i];
        for (int e = 0; e < a.n_evals; ++e) {
            const float d = a.div[(size_t)e * a.n + i];
#pragma unroll
            for (int s = 0; s < FF_MAX_SLOTS; ++s) kl[s] = slot_of[e] == s ? d : kl[s];
        }
        const float l0 = a.lp0 ? a.lp0[i] : 0.f;
        for (int j = 0; j < a.n_aux; ++j) {
            if (!a.out[j]) continue;
            float part = 0.f;
#pragma unroll
            for (int s = 0; s < FF_MAX_SLOTS; ++s) part = __builtin_fmaf(coef[j][s], kl[s], part);
            a.out[j][i] = (((use_y >> j) & 1u) ? 1.f : 0.f) * l0 + part;
        }
    }
}

static unsigned copy_grid(long long n)
{
    const long long want = (n / 4 + 255) / 256;
    return (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
}

} // namespace ff

extern "C" int ff_mlp_ode_launch(const ff_mlp_plan_t* plan, const ff_ode_args* a, void* hip_stream);

namespace {

struct Driver {
    const ff_mlp_plan_t* plan;
    const ff_ode_args* base;
    const ff_adapt_config* cfg;
    const ff_adapt_buffers* b;
    hipStream_t stream;
    long long B, D;
    bool has_lp;

    int control(int phase, const ff_norm_term* terms, int n_terms, const float* check, long long n_check, double t0, double t_end) const
    {
        ff::ControlArgs k;
        memset(&k, 0, sizeof(k));
        unsigned grid = 1;
        if (n_terms > 0 || n_check > 0) {
            grid = ff::norm_args_from_terms(k.norm, terms, n_terms, cfg->atol, cfg->rtol, check, n_check, nullptr, b->norm_workspace);
            if (grid == 0) return FF_ERR_BADARG;
        }
        k.cfg = *cfg; k.state = b->state; k.etab = b->etab; k.etab_stride = FF_ROW_HDR + plan->width;
        k.phase = phase; k.t0 = t0; k.t_end = t_end;
        if (b->exchange && (n_terms > 0 || n_check > 0)) {
            // this rank's sums of squares -> summed over the ranks (enqueued by the caller's hook) -> the controller
            k.stage = 1; k.gsums = b->exchange_sums; k.norm.sums = b->exchange_sums;
            hipLaunchKernelGGL(ff::adapt_control_kernel, dim3(grid), dim3(256), 0, stream, k);
            if (hipGetLastError() != hipSuccess) return FF_ERR_HIP;
            if (b->exchange(b->exchange_user, (void*)stream) != 0) return FF_ERR_EXCHANGE;
            k.stage = 2; k.norm.sums = nullptr; grid = 1;
        }
        hipLaunchKernelGGL(ff::adapt_control_kernel, dim3(grid), dim3(256), 0, stream, k);
        return hipGetLastError() == hipSuccess ? FF_OK : FF_ERR_HIP;
    }

    // one fused launch over the controller's table: n_evals rows, n_aux outputs into out / out_lp; unit-tangent passes
    // write their divergences to the partial buffers and are summed
    int fused(int n_evals, int n_aux, bool with_k1, float* const* out, float* const* out_lp, bool gated) const
    {
        const int P = b->n_passes > 1 ? b->n_passes : 1;
        for (int p = 0; p < P; ++p) {
            ff_ode_args a = *base;
            a.x_in = b->y; a.x_out = b->scratch_x; a.dlogp_out = has_lp ? b->scratch_lp : nullptr;
            a.noise = nullptr; a.etab = b->etab; a.in_shift = a.in_scale = a.out_scale = a.out_shift = nullptr;
            a.status = nullptr; a.n_evals = n_evals; a.noise_stride = 0;
            a.k1_in = with_k1 ? b->f0 : nullptr;
            a.kl1_in = (with_k1 && has_lp && p == 0) ? b->fl0 : nullptr;
            a.dlogp_in = (has_lp && p == 0) ? b->lp : nullptr;
            a.n_aux = n_aux;
            a.jac_out = nullptr; a.jac_all = 0;
            if (b->est_kind) {                 // the rows' Jacobians leave instead of their traces
                a.jac_out = b->est_jac; a.jac_all = 1;
                a.kl1_in = nullptr; a.dlogp_in = nullptr;
            }
            a.stage_slots = n_evals > 1 ? cfg->n_stages : (with_k1 ? 2 : 1);
            a.gate = gated ? &b->state->active : nullptr;
            a.tangent_first = b->n_passes >= 1 && base->mode == FF_MODE_EXACT ? b->pass_first[p] : 0;
            a.tangent_count = b->n_passes >= 1 && base->mode == FF_MODE_EXACT ? b->pass_count[p] : 0;
            for (int j = 0; j < FF_MAX_AUX; ++j) {
                a.aux_out[j] = (j < n_aux && p == 0) ? out[j] : nullptr;
                a.aux_lp_out[j] = nullptr;
                if (j < n_aux && has_lp && !b->est_kind)
                    a.aux_lp_out[j] = P > 1 ? b->aux_lp_pass + ((size_t)p * FF_MAX_AUX + j) * B : out_lp[j];
            }
            const int rc = ff_mlp_ode_launch(plan, &a, stream);
            if (rc != FF_OK) return rc;
        }
        if (b->est_kind) {
            const int32_t* gate = gated ? &b->state->active : nullptr;
            ff_trace_args t;
            memset(&t, 0, sizeof(t));
            t.kind = b->est_kind; t.dim = (int32_t)D; t.n_rows = n_evals; t.r = b->est_r; t.m = b->est_m; t.batch = B;
            t.jac = b->est_jac; t.probes0 = b->est_probes0; t.probes1 = b->est_probes1;
            t.out = b->est_div; t.workspace = b->est_workspace; t.gate = gate;
            const int rc = ff_trace_estimate(&t, stream);
            if (rc != FF_OK) return rc;
            ff::LpCombineArgs c;
            memset(&c, 0, sizeof(c));
            c.gate = gate; c.etab = b->etab; c.etab_stride = FF_ROW_HDR + plan->width; c.n_evals = n_evals; c.n_aux = n_aux;
            c.div = b->est_div; c.kl1 = with_k1 ? b->fl0 : nullptr; c.lp0 = b->lp; c.n = B;
            for (int j = 0; j < n_aux; ++j) c.out[j] = out_lp[j];
            hipLaunchKernelGGL(ff::adapt_lp_combine_kernel, dim3(ff::copy_grid(4 * B)), dim3(256), 0, stream, c);
            if (hipGetLastError() != hipSuccess) return FF_ERR_HIP;
        } else if (P > 1 && has_lp) {
            ff::SumArgs s;
            memset(&s, 0, sizeof(s));
            s.gate = gated ? &b->state->active : nullptr;
            s.part = b->aux_lp_pass; s.n_passes = P; s.rows = n_aux; s.n = B;
            for (int j = 0; j < n_aux; ++j) s.out[j] = out_lp[j];
            hipLaunchKernelGGL(ff::adapt_sum_kernel, dim3(ff::copy_grid(B)), dim3(256), 0, stream, s);
            if (hipGetLastError() != hipSuccess) return FF_ERR_HIP;
        }
        return FF_OK;
    }
};

} // namespace

extern "C" int ff_mlp_ode_adaptive(const ff_mlp_plan_t* plan, const ff_ode_args* base, const ff_adapt_config* cfg,
                                   const ff_adapt_buffers* b, double t0, double t_end, int32_t what, int32_t n_attempts,
                                   void* hip_stream)
{
    if (!plan || !base || !cfg || !b || n_attempts < 0) return FF_ERR_BADARG;
    if (cfg->n_stages < 2 || cfg->n_stages > FF_MAX_SLOTS || cfg->order < 1) return FF_ERR_BADARG;
    if (cfg->sched < FF_SCHED_FLOW || cfg->sched > FF_SCHED_SUBVP || !cfg->w0t || !cfg->b0) return FF_ERR_BADARG;
    if (cfg->h_real < 1 || cfg->h_real > plan->width || cfg->n_tcols < 1 || cfg->n_tcols > ff::kMaxTimeCols) return FF_ERR_UNSUPPORTED;
    if (cfg->sched != FF_SCHED_FLOW && (!cfg->emb_w || cfg->n_tcols != 2 * cfg->n_emb)) return FF_ERR_BADARG;
    if (cfg->sched == FF_SCHED_FLOW && cfg->n_tcols != 1) return FF_ERR_BADARG;
    const bool has_lp = base->mode != FF_MODE_STATE;
    if (!b->y || !b->f0 || !b->scratch_x || !b->etab || !b->out_y || !b->state || !b->norm_workspace) return FF_ERR_BADARG;
    for (int j = 0; j < FF_MAX_AUX; ++j)
        if (!b->aux[j] || (has_lp && !b->aux_lp[j])) return FF_ERR_BADARG;
    if (has_lp && (!b->lp || !b->fl0 || !b->scratch_lp || !b->out_lp)) return FF_ERR_BADARG;
    if (b->n_passes < 1 || b->n_passes > FF_ADAPT_MAX_PASSES || (b->n_passes > 1 && (!has_lp || (!b->aux_lp_pass && !b->est_kind)))) return FF_ERR_BADARG;
    if (base->batch < 0) return FF_ERR_BADARG;
    if (b->exchange && !b->exchange_sums) return FF_ERR_BADARG;
    if (b->est_kind) {
        if (base->mode != FF_MODE_EXACT || (b->est_kind != FF_TRACE_HUTCHPP && b->est_kind != FF_TRACE_XTRACE)) return FF_ERR_BADARG;
        if (!b->est_probes0 || !b->est_jac || !b->est_div || !b->est_workspace) return FF_ERR_BADARG;
        if (b->est_r < 1 || b->est_r > plan->dim || (b->est_kind == FF_TRACE_HUTCHPP && (b->est_m < 1 || !b->est_probes1))) return FF_ERR_BADARG;
    }
    if (base->batch == 0) return b->exchange ? FF_ERR_UNSUPPORTED : FF_OK;     // an empty shard cannot take part in the exchange

    Driver d{plan, base, cfg, b, (hipStream_t)hip_stream, (long long)base->batch, (long long)plan->dim, has_lp};
    const long long nBD = d.B * d.D, nB = d.B;
    int rc;
    if (what & FF_ADAPT_START) {
        // f0 = f(t0, y): the controller writes the row, the fused kernel evaluates it into f0 / fl0
        if ((rc = d.control(ff::adapt::kPhaseInit0, nullptr, 0, nullptr, 0, t0, t_end)) != FF_OK) return rc;
        float* o0[1] = {b->f0};
        float* l0[1] = {b->fl0};
        if ((rc = d.fused(1, 1, false, o0, l0, false)) != FF_OK) return rc;
        if (cfg->first_step == cfg->first_step) {
            if ((rc = d.control(ff::adapt::kPhaseFirst, nullptr, 0, nullptr, 0, 0, 0)) != FF_OK) return rc;
        } else {
            // torchdiffeq `_select_initial_step`: d0 = || y / scale ||, d1 = || f0 / scale || over the tuple state
            ff_norm_term st[6];
            int ns = 0;
            st[ns++] = ff_norm_term{b->y, nullptr, b->y, nullptr, nBD};
            if (has_lp) st[ns++] = ff_norm_term{b->lp, nullptr, b->lp, nullptr, nB};
            for (int j = 0; j < 2; ++j)
                if (b->norm_only[j] && b->norm_only_n[j] > 0)
                    st[ns++] = ff_norm_term{b->norm_only[j], nullptr, b->norm_only[j], nullptr, b->norm_only_n[j]};
            for (int i = 0; i < ns; i += FF_NORM_TERMS) {
                const int n = ns - i < FF_NORM_TERMS ? ns - i : FF_NORM_TERMS;
                if ((rc = d.control(i == 0 ? ff::adapt::kPhaseStashD0 : ff::adapt::kPhaseStashMore, st + i, n, nullptr, 0, 0, 0)) != FF_OK) return rc;
            }
            ff_norm_term d1[2] = {ff_norm_term{b->f0, nullptr, b->y, nullptr, nBD}, ff_norm_term{b->fl0, nullptr, b->lp, nullptr, nB}};
            if ((rc = d.control(ff::adapt::kPhaseInit1, d1, has_lp ? 2 : 1, nullptr, 0, 0, 0)) != FF_OK) return rc;
            // f1 = f(t0 + h0, y + h0 f0) into the proposal's derivative slot
            float* o1[1] = {b->aux[1]};
            float* l1[1] = {b->aux_lp[1]};
            if ((rc = d.fused(1, 1, true, o1, l1, false)) != FF_OK) return rc;
            ff_norm_term d2[2] = {ff_norm_term{b->aux[1], b->f0, b->y, nullptr, nBD},
                                  ff_norm_term{b->aux_lp[1], b->fl0, b->lp, nullptr, nB}};
            if ((rc = d.control(ff::adapt::kPhaseInit2, d2, has_lp ? 2 : 1, nullptr, 0, 0, 0)) != FF_OK) return rc;
        }
    }
    for (int i = 0; i < n_attempts; ++i) {
        float* ao[FF_MAX_AUX];
        float* al[FF_MAX_AUX];
        for (int j = 0; j < FF_MAX_AUX; ++j) { ao[j] = b->aux[j]; al[j] = b->aux_lp[j]; }
        if ((rc = d.fused(cfg->n_stages - 1, 4, true, ao, al, true)) != FF_OK) return rc;
        // error ratio: || err / (atol + rtol max(|y0|, |y1|)) || over the tuple state, and the finiteness of y1
        ff_norm_term er[2] = {ff_norm_term{b->aux[3], nullptr, b->y, b->aux[0], nBD},
                              ff_norm_term{b->aux_lp[3], nullptr, b->lp, b->aux_lp[0], nB}};
        if ((rc = d.control(ff::adapt::kPhaseStep, er, has_lp ? 2 : 1, b->aux[0], nBD, 0, 0)) != FF_OK) return rc;
        ff::CommitArgs c;
        memset(&c, 0, sizeof(c));
        c.state = b->state;
        c.dst[0] = b->y; c.src[0] = b->aux[0]; c.n[0] = nBD;
        c.dst[1] = b->f0; c.src[1] = b->aux[1]; c.n[1] = nBD;
        if (has_lp) {
            c.dst[2] = b->lp; c.src[2] = b->aux_lp[0]; c.n[2] = nB;
            c.dst[3] = b->fl0; c.src[3] = b->aux_lp[1]; c.n[3] = nB;
        }
        hipLaunchKernelGGL(ff::adapt_commit_kernel, dim3(ff::copy_grid(nBD)), dim3(256), 0, d.stream, c);
        if (hipGetLastError() != hipSuccess) return FF_ERR_HIP;
    }
    if (what & FF_ADAPT_FINISH) {
        ff::FinishArgs f;
        memset(&f, 0, sizeof(f));
        f.state = b->state;
        f.y0[0] = b->y; f.y1[0] = b->aux[0]; f.ymid[0] = b->aux[2]; f.f0[0] = b->f0; f.f1[0] = b->aux[1];
        f.out[0] = b->out_y; f.n[0] = nBD;
        if (has_lp) {
            f.y0[1] = b->lp; f.y1[1] = b->aux_lp[0]; f.ymid[1] = b->aux_lp[2]; f.f0[1] = b->fl0; f.f1[1] = b->aux_lp[1];
            f.out[1] = b->out_lp; f.n[1] = nB;
        }
        hipLaunchKernelGGL(ff::adapt_finish_kernel, dim3(ff::copy_grid(4 * nBD)), dim3(256), 0, d.stream, f);
        if (hipGetLastError() != hipSuccess) return FF_ERR_HIP;
    }
    return FF_OK;
}
