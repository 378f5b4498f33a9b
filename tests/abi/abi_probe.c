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
    SZ(ff_combine_args);
    OFF(ff_combine_args, x); OFF(ff_combine_args, k); OFF(ff_combine_args, coef); OFF(ff_combine_args, x_coef);
    OFF(ff_combine_args, out); OFF(ff_combine_args, n);
    SZ(ff_norm_term);
    OFF(ff_norm_term, num); OFF(ff_norm_term, sub); OFF(ff_norm_term, scale0); OFF(ff_norm_term, scale1); OFF(ff_norm_term, n);
    printf("normws %zu\n", ff_scaled_rms_workspace_bytes());

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
    printf("version %s\n", ff_version());
    return 0;
}
