// ff_kernel_args.h -- argument block of the mlp_ode kernels (host + device).
#pragma once
#include <stdint.h>

namespace ff {

constexpr int kSlots = 7;   // == FF_MAX_SLOTS (Dormand-Prince 5(4) with its FSAL stage)
constexpr int kAux = 4;     // == FF_MAX_AUX auxiliary outputs

struct KernelArgs {
    const float* x_in;
    float* x_out;
    const float* cond;
    const float* probe;
    float* dlogp_out;
    const float* noise;
    const float* wpack;
    const float* etab;
    const float* in_shift;
    const float* in_scale;
    const float* out_scale;
    const float* out_shift;
    uint32_t* status;
    long long batch;
    long long noise_stride;
    int n_evals;
    int n_hidden;
    int dim;
    int cond_dim;
    int n_tangent;       // T: tangent columns per sample (0 in FF_MODE_STATE)
    int unit_tangents;   // 1: tangents are unit vectors (exact trace); 0: tangent = probe[sample]
    int tangent_first;   // unit tangents cover dimensions [tangent_first, tangent_first + n_tangent)
    int etab_stride;     // floats per evaluation row = FF_ROW_HDR + H
    int wpack_floats;    // size of wpack (bounds of the buffer resource)
    // adaptive stepping support: preloaded first stage, and auxiliary outputs that are linear
    // combinations of the stage slots (see ff_ode_args in include/flowfusion_amd.h)
    const float* k1_in;
    const float* kl1_in;
    const float* dlogp_in;
    float* aux_out[kAux];
    float* aux_lp_out[kAux];
    int n_aux;
    // in-kernel noise (used when `noise` is NULL): Philox key, global index of row 0, noise index of row 0
    unsigned long long rng_seed;
    long long rng_sample_offset;
    int rng_noise_base;
    float* jac_out;      // [batch, dim, dim] Jacobian of the last evaluation's RHS (exact mode) or NULL
    int jac_all;         // 1: jac_out is [n_evals, batch, dim, dim] and every evaluation row writes its Jacobian
    int act_kind;        // FF_ACT_* (read by the run-time-choice instantiations only)
    float act_p0, act_p1;   // parameters of the hidden activation (FF_ACT_LEAKY_RELU / ELU / SOFTPLUS)
    unsigned long long* debug_stamps;   // diagnostic builds only (FF_DEBUG_STAMPS); NULL in the product
    const int* gate;     // ff_ode_args.gate: device word read at kernel start; 0 = this launch does nothing (NULL = run)
};

// bits of the status word (ff_ode_args.status)
constexpr unsigned kStatusNaN = 1u;        // a final state holds a NaN
constexpr unsigned kStatusBadSlot = 2u;    // an evaluation row named a stage slot this kernel does not keep on chip

} // namespace ff
