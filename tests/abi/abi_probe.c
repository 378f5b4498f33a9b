/* abi_probe.c -- compiled as C11 by tests/test_abi.py against include/flowfusion_amd.h and linked with
 * libflowfusion_amd.so: proves the header is valid C (not only C++), that the plan / packing entry points can
 * be called from C without a GPU, and prints the struct layouts so the test can compare them with the ctypes
 * mirrors (flowfusion_amd/_native.py and the stub in INTEGRATION.md). */
#include <stddef.h>
#include <stdio.h>
#include "flowfusion_amd.h"

#define SZ(T) printf("sizeof " #T " %zu\n", sizeof(T))
#define OFF(T, f) printf("offsetof " #T " " #f " %zu\n", offsetof(T, f))

int main(void)
{
    SZ(ff_mlp_plan_t);
    OFF(ff_mlp_plan_t, dim); OFF(ff_mlp_plan_t, cond_dim); OFF(ff_mlp_plan_t, n_hidden); OFF(ff_mlp_plan_t, width);
    OFF(ff_mlp_plan_t, dregs); OFF(ff_mlp_plan_t, cregs); OFF(ff_mlp_plan_t, kernel_id); OFF(ff_mlp_plan_t, tile);
    OFF(ff_mlp_plan_t, activation); OFF(ff_mlp_plan_t, act_param); OFF(ff_mlp_plan_t, precision);
    SZ(ff_ode_args);
    OFF(ff_ode_args, x_in); OFF(ff_ode_args, x_out); OFF(ff_ode_args, cond); OFF(ff_ode_args, probe);
    OFF(ff_ode_args, dlogp_out); OFF(ff_ode_args, noise); OFF(ff_ode_args, wpack); OFF(ff_ode_args, etab);
    OFF(ff_ode_args, in_shift); OFF(ff_ode_args, in_scale); OFF(ff_ode_args, out_scale); OFF(ff_ode_args, out_shift);
    OFF(ff_ode_args, status); OFF(ff_ode_args, batch); OFF(ff_ode_args, noise_stride); OFF(ff_ode_args, n_evals);
    OFF(ff_ode_args, mode); OFF(ff_ode_args, tangent_first); OFF(ff_ode_args, tangent_count); OFF(ff_ode_args, k1_in);
    OFF(ff_ode_args, kl1_in); OFF(ff_ode_args, dlogp_in); OFF(ff_ode_args, aux_out); OFF(ff_ode_args, aux_lp_out);
    OFF(ff_ode_args, n_aux); OFF(ff_ode_args, rng_noise_base); OFF(ff_ode_args, rng_seed);
    OFF(ff_ode_args, rng_sample_offset); OFF(ff_ode_args, jac_out); OFF(ff_ode_args, jac_all); OFF(ff_ode_args, stage_slots);
    OFF(ff_ode_args, gate);
    SZ(ff_combine_args);
    OFF(ff_combine_args, x); OFF(ff_combine_args, k); OFF(ff_combine_args, coef); OFF(ff_combine_args, x_coef);
    OFF(ff_combine_args, out); OFF(ff_combine_args, n);
    SZ(ff_norm_term);
    OFF(ff_norm_term, num); OFF(ff_norm_term, sub); OFF(ff_norm_term, scale0); OFF(ff_norm_term, scale1); OFF(ff_norm_term, n);
    printf("normws %zu\n", ff_scaled_rms_workspace_bytes());
    SZ(ff_adapt_state);
    OFF(ff_adapt_state, t); OFF(ff_adapt_state, dt); OFF(ff_adapt_state, t_prev); OFF(ff_adapt_state, dt_prev);
    OFF(ff_adapt_state, t_end); OFF(ff_adapt_state, h0); OFF(ff_adapt_state, d0); OFF(ff_adapt_state, d1);
    OFF(ff_adapt_state, active); OFF(ff_adapt_state, commit); OFF(ff_adapt_state, done); OFF(ff_adapt_state, error);
    OFF(ff_adapt_state, n_attempts); OFF(ff_adapt_state, n_accepted); OFF(ff_adapt_state, n_steps);
    OFF(ff_adapt_state, reserved0); OFF(ff_adapt_state, last_ratio); OFF(ff_adapt_state, reserved1);
    SZ(ff_adapt_config);
    OFF(ff_adapt_config, n_stages); OFF(ff_adapt_config, order); OFF(ff_adapt_config, alpha); OFF(ff_adapt_config, beta);
    OFF(ff_adapt_config, c_sol); OFF(ff_adapt_config, c_mid); OFF(ff_adapt_config, c_err); OFF(ff_adapt_config, rtol);
    OFF(ff_adapt_config, atol); OFF(ff_adapt_config, min_step); OFF(ff_adapt_config, max_step);
    OFF(ff_adapt_config, first_step); OFF(ff_adapt_config, max_num_steps); OFF(ff_adapt_config, sched);
    OFF(ff_adapt_config, no_sigma); OFF(ff_adapt_config, sign); OFF(ff_adapt_config, p); OFF(ff_adapt_config, emb_w);
    OFF(ff_adapt_config, n_emb); OFF(ff_adapt_config, pi); OFF(ff_adapt_config, w0t); OFF(ff_adapt_config, b0);
    OFF(ff_adapt_config, h_real); OFF(ff_adapt_config, n_tcols);
    SZ(ff_adapt_buffers);
    OFF(ff_adapt_buffers, y); OFF(ff_adapt_buffers, f0); OFF(ff_adapt_buffers, lp); OFF(ff_adapt_buffers, fl0);
    OFF(ff_adapt_buffers, aux); OFF(ff_adapt_buffers, aux_lp); OFF(ff_adapt_buffers, aux_lp_pass);
    OFF(ff_adapt_buffers, scratch_x); OFF(ff_adapt_buffers, scratch_lp); OFF(ff_adapt_buffers, etab);
    OFF(ff_adapt_buffers, out_y); OFF(ff_adapt_buffers, out_lp); OFF(ff_adapt_buffers, state);
    OFF(ff_adapt_buffers, norm_workspace); OFF(ff_adapt_buffers, norm_only); OFF(ff_adapt_buffers, norm_only_n);
    OFF(ff_adapt_buffers, n_passes); OFF(ff_adapt_buffers, pass_first); OFF(ff_adapt_buffers, pass_count);
    OFF(ff_adapt_buffers, exchange_sums); OFF(ff_adapt_buffers, exchange); OFF(ff_adapt_buffers, exchange_user);
    OFF(ff_adapt_buffers, est_kind); OFF(ff_adapt_buffers, est_r); OFF(ff_adapt_buffers, est_m); OFF(ff_adapt_buffers, est_reserved);
    OFF(ff_adapt_buffers, est_probes0); OFF(ff_adapt_buffers, est_probes1); OFF(ff_adapt_buffers, est_jac);
    OFF(ff_adapt_buffers, est_div); OFF(ff_adapt_buffers, est_workspace);
    SZ(ff_trace_args);
    OFF(ff_trace_args, kind); OFF(ff_trace_args, dim); OFF(ff_trace_args, n_rows); OFF(ff_trace_args, r); OFF(ff_trace_args, m);
    OFF(ff_trace_args, reserved); OFF(ff_trace_args, batch); OFF(ff_trace_args, jac); OFF(ff_trace_args, probes0);
    OFF(ff_trace_args, probes1); OFF(ff_trace_args, out); OFF(ff_trace_args, workspace); OFF(ff_trace_args, gate);

    /* plan + packed size for BASELINE config 2's network, from C */
    const int hidden[4] = {256, 256, 256, 256};
    ff_mlp_plan_t plan;
    int rc = ff_mlp_plan(16, 0, 4, hidden, FF_MODE_STATE, &plan);
    printf("plan rc %d tile %d width %d dregs %d cregs %d\n", rc, plan.tile, plan.width, plan.dregs, plan.cregs);
    printf("wpack_floats %zu\n", ff_mlp_wpack_floats(&plan));
    printf("spw %d\n", ff_mlp_samples_per_workgroup(&plan, FF_MODE_STATE));
    rc = ff_mlp_plan(16, 0, 4, hidden, 99, &plan);
    printf("badmode rc %d\n", rc);
    ff_ode_args a = {0};
    printf("nullargs rc %d\n", ff_mlp_ode_launch(&plan, &a, NULL));
    printf("nulladapt rc %d\n", ff_mlp_ode_adaptive(&plan, &a, NULL, NULL, 0.0, 1.0, FF_ADAPT_START, 1, NULL));
    {   /* Hutch++ with a full-rank sketch IS the trace: 2 x 2, S = (e1 + e2, e1 - e2), from C without a GPU */
        const float A[4] = {1.f, 2.f, 3.f, 4.f}, S[4] = {1.f, 1.f, 1.f, -1.f}, G[2] = {1.f, -1.f};
        float out = 0.f, ws[16];
        ff_trace_args t = {0};
        t.kind = FF_TRACE_HUTCHPP; t.dim = 2; t.n_rows = 1; t.r = 2; t.m = 1; t.batch = 1;
        t.jac = A; t.probes0 = S; t.probes1 = G; t.out = &out; t.workspace = ws;
        printf("tracews %zu\n", ff_trace_workspace_floats(FF_TRACE_HUTCHPP, 2, 2, 1));
        rc = ff_trace_estimate_host(&t);
        printf("trace rc %d value %.5f\n", rc, (double)out);
        t.r = 3;
        printf("badtrace rc %d\n", ff_trace_estimate_host(&t));
    }
    printf("version %s\n", ff_version());
    return 0;
}
