/*
 * flowfusion_amd.h -- C ABI of the MI355X (gfx950) sampling / log-density hot path.
 *
 * This is the drop-in boundary for flowfusion's probability-flow ODE / reverse-SDE
 * sampling and log-density path.  The reference (Cosmo-Pop/flowfusion) is pure Python
 * and has no FFI; what a native replacement has to take over is the time-stepping loop
 * that the reference delegates to torchdiffeq / runs in Python:
 *
 *   ScoreModel.sample_ode_from_base   flowfusion/diffusion.py:566-640  (odeint call :631-639)
 *   ScoreModel.solve_odes_forward     flowfusion/diffusion.py:642-754  (odeint call :744-752)
 *   ScoreModel.sample_sde             flowfusion/diffusion.py:510-563  (Python EM loop :543-562)
 *   ScoreModel.forward (ODE RHS)      flowfusion/diffusion.py:281-334,505-508
 *   MLP.forward                       flowfusion/diffusion.py:82-121
 *   ODEFlow.sample / solve_ode_forward            flowfusion/flow.py:259-306, 308-384
 *   ConditionalODEFlow.sample / solve_ode_forward flowfusion/flow.py:750-799, 801-883
 *   ODEFlow.dynamics / ConditionalODEFlow.dynamics flowfusion/flow.py:89-120, 553-596
 *
 * One fused kernel family ("mlp_ode") covers all of them.  It integrates, for every
 * sample independently,
 *
 *        dy/ds = a_e * y + b_e * NET(y, cond ; c1_e)            (e = RHS evaluation index)
 *
 * where NET is the Linear/SiLU stack of the reference's MLP with the part of the first
 * layer that depends only on time (time embedding or the raw `t` column, plus bias)
 * pre-reduced on the host into the per-evaluation vector c1_e, and (a_e, b_e) are the
 * scalar SDE schedule terms (-beta/2, -g^2/(2 sigma), ...; 0 and 1 for the flows).  The
 * whole explicit Runge-Kutta / Euler-Maruyama loop runs on-chip: the state is read from
 * HBM once and written once.
 *
 * Conventions: all pointers in ff_ode_args are DEVICE pointers to contiguous fp32
 * row-major arrays owned by the caller; launchers enqueue on the given hipStream_t and
 * return immediately (0 = enqueued, <0 = error, nothing enqueued); no global state, safe
 * to call from several host threads on different streams.
 */
#ifndef FLOWFUSION_AMD_H
#define FLOWFUSION_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FF_OK               0
#define FF_ERR_BADARG      -1   /* null pointer / negative size / inconsistent plan        */
#define FF_ERR_UNSUPPORTED -2   /* no gfx950 kernel instantiation covers this shape        */
#define FF_ERR_HIP         -3   /* HIP runtime refused the launch (see ff_last_hip_error)  */
#define FF_ERR_EXCHANGE    -4   /* the caller's ff_adapt_buffers.exchange hook returned non-zero */

#define FF_MAX_SLOTS   7        /* Runge-Kutta stage slots kept on chip (dopri5 + FSAL)    */
#define FF_MAX_AUX     4        /* auxiliary linear-combination outputs per launch         */
#define FF_ROW_HDR     32       /* 4-byte words in the header of one evaluation row        */

/* mode flags for ff_ode_args.mode */
#define FF_MODE_STATE      0    /* integrate the state only (`tile` samples per wavefront) */
#define FF_MODE_HUTCH      1    /* state + Hutchinson divergence e^T J e in forward mode
                                   (16 samples + 16 tangent columns per wavefront)         */
#define FF_MODE_EXACT      2    /* state + exact divergence tr J by unit tangents (reference
                                   default: diffusion.py:483-503, flow.py:158-161); see
                                   ff_ode_args.tangent_first/count                          */

/* Hidden-layer activation (the `activation` argument of the reference constructors,
 * diffusion.py:38,77,118; flow.py:41,70; 478,518).  FF_ACT_SILU, the reference default, has
 * instantiations of its own; the others run on instantiations that select the function at
 * run time.  Values and slopes follow torch.nn's definitions. */
#define FF_ACT_SILU        0    /* a * sigmoid(a)                                          */
#define FF_ACT_TANH        1
#define FF_ACT_SIGMOID     2
#define FF_ACT_RELU        3
#define FF_ACT_LEAKY_RELU  4    /* act_param[0] = negative_slope                           */
#define FF_ACT_ELU         5    /* act_param[0] = alpha                                    */
#define FF_ACT_SOFTPLUS    6    /* act_param[0] = beta, act_param[1] = threshold           */
#define FF_ACT_GELU        7    /* a * Phi(a); erf by a 1.5e-7-accurate rational form      */
#define FF_ACT_GELU_TANH   8    /* nn.GELU(approximate="tanh")                             */
#define FF_ACT_COUNT       9

/* Arithmetic of the Linear layers (ff_mlp_plan_t.precision).
 *   FF_PREC_F32     exact fp32 FMA chains on the f32 MFMAs -- what the reference computes; the default.
 *   FF_PREC_BF16X3  opt-in: every operand is split into three bf16 parts (hi + mid + lo carry 24 mantissa
 *                   bits) and a product is the sum of the six bf16 MFMAs whose parts' exponents sum to more than
 *                   2^-24 (hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi), accumulated in fp32: fp32-class
 *                   accuracy (error ~1e-7 relative to sum |w||x|, like an fp32 dot product) at the bf16 matrix
 *                   rate.  SiLU networks of 1-4 hidden layers up to 256 wide, dim <= 16, cond_dim <= 16; state-only
 *                   solves (FF_MODE_STATE: noise rows and the adaptive-step fields included); see DESIGN.md section 3.2.
 *   FF_PREC_BF16X2  opt-in: two bf16 parts by round-to-nearest (hi + mid = the operand to 16 significand bits) and
 *                   the three products hi.hi, hi.mid, mid.hi: operands rounded to 2^-17 relative (TF32 keeps 2^-11),
 *                   unbiased, fp32 accumulation; error of a 256-term layer ~4e-7 relative to sum |w||x| in the mean
 *                   (fp32: 2e-8) at half the MFMAs of FF_PREC_BF16X3.  Same networks, every mode (jac_out excepted)
 *                   and, state-only, states of up to 32 dimensions (those plans keep 4 stage slots on chip instead of
 *                   FF_MAX_SLOTS: tables must not name a slot >= 4). */
#define FF_PREC_F32        0
#define FF_PREC_BF16X3     1
#define FF_PREC_BF16X2     2

/* bits OR-ed into ff_ode_args.status by a launch */
#define FF_STATUS_NAN      1u   /* a final state holds a NaN                                */
#define FF_STATUS_BAD_SLOT 2u   /* an evaluation row named a stage slot the kernel does not
                                   keep on chip (>= FF_MAX_SLOTS; >= 4 for the plans and twins
                                   with four slots): that row's right-hand side was NOT stored,
                                   the results are invalid                                  */

/* evaluation-row flag bits (word 3 of the row header) */
#define FF_ROW_STEP_END    1u   /* after this evaluation: y += sum_s cout[s] * k[s]        */
#define FF_ROW_NOISE       2u   /* after the step update: y += gn * noise[noise_index]     */

/*
 * Kernel plan for one network shape: which compiled instantiation serves it and how its
 * weights are tiled.  Filled by ff_mlp_plan; treat as opaque apart from reading.
 */
typedef struct ff_mlp_plan_t {
    int32_t dim;         /* D  state dimension = network outputs                          */
    int32_t cond_dim;    /* C  conditional inputs (0 = unconditional)                     */
    int32_t n_hidden;    /* number of hidden (Linear+SiLU) layers, >= 1                   */
    int32_t width;       /* common padded hidden width used on chip (multiple of 32)      */
    int32_t dregs;       /* state registers per lane, padded to the kernel's              */
    int32_t cregs;       /* conditional registers per lane                                */
    int32_t kernel_id;   /* index into the compiled instantiation table                   */
    int32_t tile;        /* MFMA columns per wavefront: 32 (32x32x2 f32) or 16 (16x16x4)  */
    int32_t activation;  /* FF_ACT_*                                                      */
    float   act_param[2];/* parameters of the activation (see FF_ACT_*), else 0           */
    int32_t precision;   /* FF_PREC_*: arithmetic of the Linear layers (0 = exact fp32, the default)     */
} ff_mlp_plan_t;

/* Arguments of one fused integration launch. */
typedef struct ff_ode_args {
    const float* x_in;       /* [batch, dim]  initial state                                */
    float*       x_out;      /* [batch, dim]  final state                                  */
    const float* cond;       /* [batch, cond_dim] or NULL                                  */
    const float* probe;      /* [batch, dim]  Hutchinson probe e (FF_MODE_HUTCH) or NULL   */
    float*       dlogp_out;  /* [batch]       integrated divergence (modes 1,2) or NULL    */
    const float* noise;      /* [n_noise, batch, dim] standard normals (EM); NULL = in-kernel
                                noise (rng_* below) if the table has FF_ROW_NOISE rows      */
    const float* wpack;      /* packed weights, ff_mlp_wpack_floats() floats               */
    const float* etab;       /* [n_evals, FF_ROW_HDR + width] evaluation rows              */
    const float* in_shift;   /* [dim] or NULL: y0 = (x_in - in_shift) / in_scale           */
    const float* in_scale;   /* [dim] or NULL                                              */
    const float* out_scale;  /* [dim] or NULL: x_out = y * out_scale + out_shift           */
    const float* out_shift;  /* [dim] or NULL                                              */
    uint32_t*    status;     /* device word, OR-ed with FF_STATUS_* bits; NULL ok                */
    int64_t      batch;      /* number of samples                                          */
    int64_t      noise_stride; /* floats between consecutive noise slabs (>= batch*dim)    */
    int32_t      n_evals;    /* rows in etab                                               */
    int32_t      mode;       /* FF_MODE_*                                                  */
    int32_t      tangent_first;  /* FF_MODE_EXACT: this launch carries the unit tangents of   */
    int32_t      tangent_count;  /* dimensions [first, first+count); count 0 = all `dim`.  A
                                    trace over more dimensions than fit one wavefront
                                    (count + 1 <= plan.tile) is the sum of dlogp_out over
                                    several launches, each of which also returns x_out.     */
    /* --- adaptive stepping support (all optional; zero / NULL when unused) ---------------------
     * One attempt of an embedded Runge-Kutta step is one launch: the first stage k[0] is supplied
     * by the caller (FSAL), the evaluation rows fill the other slots WITHOUT a STEP_END flag, and
     * the results leave as linear combinations of the slots,
     *     aux_j = use_y_j * y_in + sum_s coef_j[s] * k[s]          j < n_aux <= FF_MAX_AUX
     * (new state, last stage, dense-output midpoint, error estimate).  Their coefficients live in
     * two extra rows appended to etab (so etab has n_evals + 2 rows): row n_evals carries coef_0
     * in its cin words, coef_1 in its cout words and the use_y bits in its flags word; row
     * n_evals+1 carries coef_2 / coef_3 the same way.  With a divergence mode the same
     * combinations of the divergence slots (and use_y_j * dlogp_in) go to aux_lp_out[j]. */
    const float* k1_in;      /* [batch, dim]  stage slot 0 (derivative at the step start) or NULL */
    const float* kl1_in;     /* [batch]       divergence of stage slot 0 or NULL                  */
    const float* dlogp_in;   /* [batch]       integrated divergence at the step start or NULL     */
    float*       aux_out[FF_MAX_AUX];     /* [batch, dim] each, or NULL                            */
    float*       aux_lp_out[FF_MAX_AUX];  /* [batch] each, or NULL                                 */
    int32_t      n_aux;
    int32_t      rng_noise_base;  /* added to the rows' noise_index in the Philox counter          */
    /* --- in-kernel noise (FF_MODE_STATE; used by FF_ROW_NOISE rows when `noise` is NULL) ---------
     * The standard normal of (global sample g = rng_sample_offset + row, noise index n =
     * rng_noise_base + the row header's noise_index, dimension d) is word d%4 of
     *     Philox4x32-10( counter = (g lo, g hi, n, d/4), key = (rng_seed lo, rng_seed hi) )
     * turned into normals pairwise by Box-Muller: u1 = ((w_a >> 8) + 0.5) 2^-24, u2 = (w_b >> 8) 2^-24,
     * z_a = sqrt(-2 ln u1) cos(2 pi u2), z_b = sqrt(-2 ln u1) sin(2 pi u2), (a, b) = (0, 1), (2, 3).
     * Keyed by the GLOBAL sample index, the stream of a sample does not depend on how a batch is cut
     * into launches or shards: 1-, 2-, 4- and 8-GPU runs of one seed give the same samples. */
    uint64_t     rng_seed;
    int64_t      rng_sample_offset;
    /* --- Jacobian output (FF_MODE_EXACT; optional) -----------------------------------------------
     * jac_out[b][j][i] = d rhs_i / d y_j of the LAST evaluation row, for the unit tangents j of this
     * launch (rows tangent_first .. tangent_first + count - 1; several launches fill the matrix).
     * Row j is the product J^T e_j that the reference's Hutch++ / XTrace estimators obtain by
     * reverse mode (`vjp_fn`, diffusion.py:360-372, 431-443); with the whole matrix on hand those
     * estimators are small per-sample linear algebra (flowfusion_amd/trace_estimators.py). */
    float*       jac_out;
    int32_t      jac_all;    /* 0: jac_out = [batch][dim][dim], the Jacobian of the LAST evaluation row (above);
                                1: jac_out = [n_evals][batch][dim][dim], the Jacobian of EVERY row.  The state never
                                depends on the divergence, so one launch can integrate a whole fixed-grid table and
                                hand back everything the Hutch++ / XTrace estimators need; they are then evaluated
                                for all rows at once and combined with the tableau's weights by the caller. */
    int32_t      stage_slots; /* stage slots the table uses (highest slot index + 1, <= FF_MAX_SLOTS); 0 = unknown.
                                 Plans whose kernel keeps fewer slots on chip than FF_MAX_SLOTS (FF_PREC_BF16X2 with dim > 16:
                                 4) require the table to stay within them (a larger value is FF_ERR_UNSUPPORTED); for the others
                                 a value <= 4 lets the launcher pick a twin that trades unused slots for occupancy.  The value is
                                 a PROMISE: a row that names a slot the chosen kernel does not keep is refused by the kernel
                                 (FF_STATUS_BAD_SLOT, nothing stored) -- a wrong hint is an error, not a no-op.  With a correct
                                 hint the results do not depend on it. */
    const int32_t* gate;     /* optional DEVICE word read when the kernel starts: 0 = this launch does nothing.  For launches
                                enqueued ahead of a decision taken on the device (ff_mlp_ode_adaptive); NULL = run. */
} ff_ode_args;

/* Library / build identification: returns e.g. "flowfusion_amd 0.1 gfx950". */
const char* ff_version(void);

/* Number of compiled kernel instantiations, and a printable description of each (ids [0, n_f32) are the fp32
 * family in ff_mlp_plan_t.kernel_id order, the split-precision family follows). */
int ff_kernel_count(void);
const char* ff_kernel_name(int kernel_id);
/* Name of the instantiation a plan selects (either family), NULL for an invalid plan. */
const char* ff_plan_kernel_name(const struct ff_mlp_plan_t* plan);

/*
 * Choose the kernel instantiation for a network: `hidden_widths[n_hidden]` are the
 * reference's `units` / `hidden_units` (diffusion.py:61-72, flow.py:64-71).  Widths are
 * padded with zero rows/columns up to a compiled width; FF_ERR_UNSUPPORTED if none fits.
 */
int ff_mlp_plan(int dim, int cond_dim, int n_hidden, const int* hidden_widths, int mode,
                ff_mlp_plan_t* plan_out);

/* The same for a network whose hidden layers use `activation` (FF_ACT_*) with parameters
 * `act_param[2]` (NULL = zeros); ff_mlp_plan(...) is ff_mlp_plan_act(..., FF_ACT_SILU, NULL, ...). */
int ff_mlp_plan_act(int dim, int cond_dim, int n_hidden, const int* hidden_widths, int mode,
                    int activation, const float* act_param, ff_mlp_plan_t* plan_out);

/* The same with an explicit arithmetic (FF_PREC_*); FF_ERR_UNSUPPORTED if no instantiation of that family
 * covers the shape / mode / activation.  ff_mlp_plan_act(...) is ff_mlp_plan_prec(..., FF_PREC_F32, ...). */
int ff_mlp_plan_prec(int dim, int cond_dim, int n_hidden, const int* hidden_widths, int mode,
                     int activation, const float* act_param, int precision, ff_mlp_plan_t* plan_out);

/* Floats (4-byte units) in the packed weight buffer of a plan. */
size_t ff_mlp_wpack_floats(const ff_mlp_plan_t* plan);

/*
 * Repack nn.Linear weights (HOST pointers, row-major [out_features, in_features]) into
 * the MFMA operand order of the plan.  W[0] is the first layer with `in_features0`
 * columns, of which columns [x_col0, x_col0+dim) multiply the state and
 * [c_col0, c_col0+cond_dim) the conditional (reference column order: MLP =
 * [time-embedding | x | cond] diffusion.py:109-113; flows = [x | t | cond]
 * flow.py:112-115,583-586); its remaining columns and its bias are NOT packed -- the
 * caller folds them into c1_e.  W[1..n_hidden-1] are hidden layers, W[n_hidden] the output
 * layer.  b[0] is ignored (may be NULL).  `out` is a HOST buffer of
 * ff_mlp_wpack_floats(plan) floats.
 */
int ff_mlp_wpack(const ff_mlp_plan_t* plan, const float* const* W, const float* const* b,
                 const int* hidden_widths, int in_features0, int x_col0, int c_col0,
                 float* out);

/*
 * Enqueue the fused integration on `hip_stream` (a hipStream_t; NULL = default stream).
 * Replaces, per mode and table (see the header comment and INTEGRATION.md):
 *   odeint(self, (z,), t, method, ...)                     diffusion.py:631-639   FF_MODE_STATE
 *   odeint(self, (x0, delta_logpx), t, method, ...)        diffusion.py:744-752   FF_MODE_HUTCH / FF_MODE_EXACT
 *   the Euler-Maruyama loop of sample_sde                  diffusion.py:543-562   FF_MODE_STATE + FF_ROW_NOISE rows
 *   odeint(self.dynamics, (xT[, cond]), t)                 flow.py:299-303, 792-796            FF_MODE_STATE
 *   odeint(self.dynamics_with_jacobian, (x[, cond], logJ)) flow.py:371-382, 869-881            FF_MODE_EXACT
 * Returns FF_OK once enqueued; an argument or shape error enqueues nothing (FF_ERR_HIP from the second of two launches,
 * below, leaves the first enqueued and the outputs undefined).
 * One call may enqueue TWO kernels: the rows left over after the full rounds of tiles the chip runs at once go to the
 * small-batch twin of the kernel when that is faster (same arithmetic, bitwise the same results; FF_TAIL_SPLIT=0 in the
 * environment keeps one kernel).  Batches below one round run on the twin alone where it wins (FF_COOP=0/1 pins that).
 */
int ff_mlp_ode_launch(const ff_mlp_plan_t* plan, const ff_ode_args* args, void* hip_stream);

/* Which kernel(s) ff_mlp_ode_launch enqueues for `batch` samples of a plan in `mode` (`tangent_count`: as in ff_ode_args,
 * 0 = the mode's default; `jac_out` != 0: a launch with a Jacobian output, which is never split): the launcher's rule as a
 * query, so that callers, benchmarks and tests can name the kernel that ran without timing it.  Honours FF_COOP /
 * FF_TAIL_SPLIT in the environment like the launcher does.  Negative: FF_ERR_*. */
#define FF_LAUNCH_ONE_WAVE          0   /* the one-wavefront-per-tile kernel                                         */
#define FF_LAUNCH_TWIN              1   /* its cooperative small-batch twin (or a wide catch-all): a tile per workgroup */
#define FF_LAUNCH_ONE_WAVE_AND_TWIN 2   /* full rounds on the first, the leftover rows on the twin (two launches)     */
int ff_mlp_launch_kind(const ff_mlp_plan_t* plan, int64_t batch, int32_t mode, int32_t tangent_count, int32_t jac_out);

/* Samples handled by one workgroup of a plan/mode (for sizing and roofline accounting). */
int ff_mlp_samples_per_workgroup(const ff_mlp_plan_t* plan, int mode);

/* hipError_t of the most recent failing launch on this thread (0 if none). */
int ff_last_hip_error(void);

/* ---- streaming helpers beside the fused integrator (csrc/ff_aux.hip) --------------------------------- */

/* Noise index reserved for the prior draw of a sample in the counter-based stream (never used by the
 * Euler-Maruyama rows, which count from rng_noise_base upwards). */
#define FF_PRIOR_NOISE_INDEX 0xFFFFFFFFu
/* Noise index reserved for the Hutchinson probe of a sample: e = sign(z(seed, global row, this index, d)) replaces the
 * host draw `torch.sign(torch.randn(shape)).to(device)` (diffusion.py:701) when a log-density batch is sharded over GPUs
 * or the host draw should leave the critical path (host side: probe="philox"). */
#define FF_PROBE_NOISE_INDEX 0xFFFFFFFEu
/* Noise indices reserved for the Hutch++ / XTrace probes of a sample (diffusion.py:703-719: S, G, O drawn once per solve as
 * `torch.sign(torch.randn(n, B, D))`): probe c of the FIRST set (S, or O) is the sign of z(seed, global row,
 * FF_TRACE_PROBE_NOISE_BASE + c, d), probe c of the SECOND set (G) that of index FF_TRACE_PROBE_NOISE_BASE + 0x8000 + c
 * (host side: probe="philox" on a model with hutchpp=True / xtrace=True) -- the same probes whatever the sharding. */
#define FF_TRACE_PROBE_NOISE_BASE 0xFFFE0000u

/*
 * out[r][d] = scale * z(seed, global row sample_offset + r, noise_index, d)  for r < batch, d < dim, with z the
 * standard normal of the in-kernel stream defined at ff_ode_args.rng_seed (Philox4x32-10 + Box-Muller).
 * Replaces the prior draw `self.sde.prior(dims).sample([batch])` of sample_sde (diffusion.py:532-536; the
 * priors are Normal(0, 1) :1093 and Normal(0, sigma_max) :1003, hence `scale`) when a batch is sharded over
 * GPUs: each rank fills only its rows and every world size starts from the same points.  `out` is a DEVICE
 * pointer to [batch, dim] contiguous fp32.  Enqueues on hip_stream.
 */
int ff_normal_fill(float* out, int64_t batch, int32_t dim, uint64_t seed, int64_t sample_offset,
                   uint32_t noise_index, float scale, void* hip_stream);

/*
 * out = x_coef * x + sum_s coef[s] * k[s]   over n contiguous fp32 elements (DEVICE pointers; x may be NULL,
 * k[s] is not read when coef[s] == 0; out may alias x or a k[s]).
 * The stage-input, step-update, dense-output and error-estimate algebra of an explicit Runge-Kutta step for a
 * right-hand side evaluated OUTSIDE the library -- the reference accepts any `model=` module in ScoreModel
 * (diffusion.py:201,233-238) and hands the stepping to torchdiffeq (call sites diffusion.py:631-639, 744-752),
 * which spends one elementwise pass over the state per term; this is one pass per combination.
 */
typedef struct ff_combine_args {
    const float* x;
    const float* k[FF_MAX_SLOTS];
    float        coef[FF_MAX_SLOTS];
    float        x_coef;
    float*       out;
    int64_t      n;
} ff_combine_args;
int ff_stage_combine(const ff_combine_args* args, void* hip_stream);

/*
 * Scaled RMS norms of an adaptive step, in one launch:
 *     out[i] = sqrt( mean_k ( (num_i[k] - sub_i[k]) / (atol + rtol * max(|scale0_i[k]|, |scale1_i[k]|)) )^2 )   i < n_terms
 *     out[n_terms] = 1 if `check[0 .. n_check)` holds a NaN or an infinity, else 0
 * (`sub` and `scale1` may be NULL).  These are the norms torchdiffeq's adaptive solvers take of the tuple state behind
 * the reference's default `method="dopri5"` (call sites diffusion.py:631-639, 744-752; flow.py:299-303, 371-382):
 * `_compute_error_ratio` (num = error estimate, scale0 / scale1 = y0 / y1) and `_select_initial_step` (num = y or f or
 * f1 - f0, scale0 = y); one term per component of the tuple, the caller takes the maximum (the mixed norm).
 * Deterministic (no floating-point atomics).  All pointers are DEVICE pointers; `out` has n_terms + 1 floats;
 * `workspace` is ff_scaled_rms_workspace_bytes() bytes of device memory whose first 16 bytes are zero before the first
 * use (the kernel leaves them zero); one workspace per stream.
 */
#define FF_NORM_TERMS 3
typedef struct ff_norm_term {
    const float* num;
    const float* sub;
    const float* scale0;
    const float* scale1;
    int64_t      n;
} ff_norm_term;
size_t ff_scaled_rms_workspace_bytes(void);
int ff_scaled_rms(const ff_norm_term* terms, int32_t n_terms, float atol, float rtol, const float* check,
                  int64_t n_check, float* out, void* workspace, void* hip_stream);

/* ---- Hutch++ / XTrace divergence estimates from recorded Jacobians (csrc/ff_trace.hip, csrc/ff_trace_est.h) ----
 *
 * Replaces the two estimator branches of ScoreModel.forward, diffusion.py:336-400 (Hutch++) and :402-481 (XTrace), for
 * every right-hand-side evaluation of a launch at once: with ff_ode_args.jac_all = 1 a fused FF_MODE_EXACT launch leaves
 * A[e][b] = J^T of every evaluation row e and sample b; one launch of this kernel returns the estimates
 *     out[e][b] = hutchpp(A[e][b]; S[:, b], G[:, b])   or   xtrace(A[e][b]; O[:, b])
 * with the probes laid out as the reference stores them ([n_probes][batch][dim], entries +-1; drawn once per solve,
 * diffusion.py:703-719).  The QR behind both is LAPACK's Householder scheme, what `torch.linalg.qr` (:378, :441) runs on
 * the reference's CPU path.  All pointers are DEVICE pointers (HOST pointers for ff_trace_estimate_host).
 */
#define FF_TRACE_HUTCHPP 1
#define FF_TRACE_XTRACE  2
typedef struct ff_trace_args {
    int32_t kind;            /* FF_TRACE_*                                                                     */
    int32_t dim;             /* D                                                                               */
    int32_t n_rows;          /* evaluation rows recorded in `jac`                                               */
    int32_t r;               /* Hutch++: sketch probes S (`hpp_rank`, <= D); XTrace: probes O (`xt_vecs`, <= D)  */
    int32_t m;               /* Hutch++: residual probes G (`hpp_vecs`, >= 1); XTrace: unused                   */
    int32_t reserved;
    int64_t batch;
    const float* jac;        /* [n_rows][batch][D][D]: row j of a matrix is J^T e_j (ff_ode_args.jac_out)        */
    const float* probes0;    /* S or O: [r][batch][D]                                                           */
    const float* probes1;    /* G: [m][batch][D], or NULL (XTrace)                                              */
    float*       out;        /* [n_rows][batch]                                                                 */
    float*       workspace;  /* ff_trace_workspace_floats(kind, dim, r, n_rows * batch) floats                   */
    const int32_t* gate;     /* optional DEVICE word: 0 = the launch does nothing (see ff_ode_args.gate)         */
} ff_trace_args;
size_t ff_trace_workspace_floats(int32_t kind, int32_t dim, int32_t r, int64_t items);
int ff_trace_estimate(const ff_trace_args* args, void* hip_stream);
/* The same arithmetic on the HOST (tests without a GPU): HOST pointers; `workspace` holds
 * ff_trace_workspace_floats(kind, dim, r, 1) floats; `gate` is ignored. */
int ff_trace_estimate_host(const ff_trace_args* args);

/* ---- adaptive embedded Runge-Kutta solves with the step control ON THE DEVICE (csrc/ff_adaptive.hip) ---------
 *
 * Replaces `odeint(func, state, t, method="dopri5" | "bosh3" | "fehlberg2" | "adaptive_heun", rtol=, atol=, options=)`
 * -- the reference's DEFAULT solver at every call site (diffusion.py:572,631-639; 649,744-752; 762; flow.py:299-303,
 * 313,371-382) -- including torchdiffeq's batch-global step control: `_select_initial_step`, `_compute_error_ratio`
 * (mixed norm over the tuple state), accept / reject, `_optimal_step_size`, min / max step clamps, the fourth-order dense
 * output at t[-1].  One attempted step is three launches enqueued back to back with NO host round trip:
 *     the fused attempt (ff_mlp_ode_launch semantics, gated by ff_adapt_state.active)
 *  -> norms + controller (one reduction launch; its last block decides, advances (t, dt) and writes the NEXT attempt's
 *     evaluation rows: stage times, the SDE schedule scalars a_e / b_e and the first-layer time part c1_e, computed in
 *     fp32 in the reference's operation order: diffusion.py:276-278,905,1131,1316; MLP.forward :109-113; flow.py:112-118)
 *  -> commit (an accepted step's (y1, f1, lp1, fl1) become the current state).
 * The host enqueues a chunk of attempts, then reads ff_adapt_state once; attempts behind the end of the solve are no-ops.
 */
#define FF_SCHED_FLOW   0   /* flows: a = 0, b = 1, c1 = w_t t + b1 (flow.py:112-118)                       */
#define FF_SCHED_VE     1   /* VESDE (diffusion.py:818-1003):    p = {sigma_min, sigma_max, T}               */
#define FF_SCHED_VP     2   /* VPSDE (diffusion.py:1006-1180):   p = {beta_min, beta_max, T}                 */
#define FF_SCHED_SUBVP  3   /* SUBVPSDE (diffusion.py:1183-1366): p = {beta_min, beta_max, T}                */

#define FF_ADAPT_OK            0
#define FF_ADAPT_ERR_UNDERFLOW 1   /* torchdiffeq: AssertionError "underflow in dt {dt}" (state.dt holds it)   */
#define FF_ADAPT_ERR_NONFINITE 2   /* torchdiffeq: "non-finite values in state `y`"                            */
#define FF_ADAPT_ERR_MAXSTEPS  3   /* torchdiffeq: "max_num_steps exceeded"                                     */

#define FF_ADAPT_MAX_PASSES    8   /* unit-tangent passes of one attempt (FF_MODE_EXACT with dim + 1 > tile)    */

/* Controller state: 128 bytes of DEVICE memory, written by the controller, read back by the host. */
typedef struct ff_adapt_state {
    double  t;           /* solver time reached: the end of the last accepted step (rk_state.t1)            */
    double  dt;          /* step of the next attempt (clamped to [min_step, max_step])                      */
    double  t_prev;      /* start of the last accepted step (rk_state.t0)                                   */
    double  dt_prev;     /* its length                                                                      */
    double  t_end;       /* where the solve stops (solver time: a decreasing span is solved negated)        */
    double  h0;          /* initial-step rule: the trial step                                               */
    double  d0;          /* ... the norm of the scaled state                                                */
    double  d1;          /* ... the norm of the scaled derivative                                           */
    int32_t active;      /* gate of the launches that follow: 1 = attempts run                              */
    int32_t commit;      /* 1 = the latest attempt was accepted and is not the last: commit adopts it       */
    int32_t done;        /* the solve reached t_end: the buffers hold the last accepted step                */
    int32_t error;       /* FF_ADAPT_ERR_*                                                                  */
    int32_t n_attempts;
    int32_t n_accepted;
    int32_t n_steps;     /* loop iterations (torchdiffeq counts them against max_num_steps)                 */
    int32_t reserved0;
    float   last_ratio;  /* error ratio of the latest attempt                                               */
    float   reserved1[7];
} ff_adapt_state;

/* The embedded pair, the tolerances and the time-dependent part of the right-hand side (HOST struct). */
typedef struct ff_adapt_config {
    int32_t n_stages;    /* stages including the first-same-as-last one, 2 .. FF_MAX_SLOTS                  */
    int32_t order;       /* order of the pair (torchdiffeq `order`): dopri5 5, bosh3 3, fehlberg2 / adaptive_heun 2 */
    float   alpha[FF_MAX_SLOTS - 1];        /* stage i + 1 sits at t0 + alpha[i] dt (exactly t1 when alpha[i] == 1) */
    float   beta[FF_MAX_SLOTS - 1][8];      /* ... on y0 + dt sum_j beta[i][j] k_j                           */
    float   c_sol[8];    /* y1 = y0 + dt sum_j c_sol[j] k_j                                                 */
    float   c_mid[8];    /* dense-output midpoint                                                           */
    float   c_err[8];    /* error estimate                                                                  */
    float   rtol, atol;
    double  min_step, max_step;             /* torchdiffeq options (0, +inf by default)                      */
    double  first_step;  /* option first_step; NaN = torchdiffeq's `_select_initial_step`                   */
    int32_t max_num_steps;
    int32_t sched;       /* FF_SCHED_*                                                                      */
    int32_t no_sigma;    /* ScoreModel(no_sigma=): the network output is NOT divided by sigma(t) (diffusion.py:268-272) */
    float   sign;        /* +1, or -1 for a decreasing span (solved in -t with the right-hand side negated)  */
    double  p[4];        /* schedule parameters (FF_SCHED_*)                                                */
    const float* emb_w;  /* DEVICE [n_emb]: Gaussian-Fourier frequencies W (MLP.W, diffusion.py:73-76); NULL for flows */
    int32_t n_emb;
    float   pi;          /* MLP.pi as the model holds it (fp32)                                             */
    const float* w0t;    /* DEVICE [h_real][n_tcols]: the first layer's time columns ([sin | cos] features, or the
                            single t column of a flow), row-major                                           */
    const float* b0;     /* DEVICE [h_real]: the first layer's bias                                         */
    int32_t h_real;      /* rows of the first layer (<= plan.width; c1 is zero-padded to the width)          */
    int32_t n_tcols;     /* 2 * n_emb, or 1 for flows                                                        */
} ff_adapt_config;

/* Work buffers (all DEVICE memory owned by the caller; [B] = batch, [B, D] = batch x dim floats). */
typedef struct ff_adapt_buffers {
    float* y;            /* [B, D] in: the initial state; during the solve: the current state               */
    float* f0;           /* [B, D] derivative at the current state (FSAL)                                   */
    float* lp;           /* [B]    integrated divergence (modes 1, 2; in: its initial value) or NULL        */
    float* fl0;          /* [B]    its derivative or NULL                                                   */
    float* aux[FF_MAX_AUX];      /* [B, D] each: proposal y1, last stage f1, dense-output midpoint, error estimate */
    float* aux_lp[FF_MAX_AUX];   /* [B] each, or NULL                                                       */
    float* aux_lp_pass;  /* [n_passes][FF_MAX_AUX][B] partial divergences when n_passes > 1, else NULL      */
    float* scratch_x;    /* [B, D]                                                                          */
    float* scratch_lp;   /* [B] or NULL                                                                     */
    float* etab;         /* [FF_MAX_SLOTS + 1][FF_ROW_HDR + plan.width] evaluation rows, written by the controller */
    float* out_y;        /* [B, D] the solution at t_end (dense output of the last step)                    */
    float* out_lp;       /* [B] or NULL                                                                     */
    ff_adapt_state* state;
    void*  norm_workspace;       /* ff_scaled_rms_workspace_bytes() bytes, first 16 zero                    */
    const float* norm_only[2];   /* components the reference carries in the solver state with a zero derivative (the raw */
    int64_t norm_only_n[2];      /* `conditional` of ConditionalODEFlow, flow.py:779-796): they enter d0 of the initial step */
    int32_t n_passes;            /* 1, or the unit-tangent passes of FF_MODE_EXACT                          */
    int32_t pass_first[FF_ADAPT_MAX_PASSES];
    int32_t pass_count[FF_ADAPT_MAX_PASSES];
    /* Step control over SEVERAL shards of one batch (one process per GPU).  torchdiffeq's norms are over the whole batch
     * it is handed (one step size for all samples), so when the batch is cut over ranks the sums of squares behind every
     * norm have to meet: the one exchange step of this path.  With `exchange` set each norm is taken in two launches --
     * reduce into exchange_sums (FF_EXCHANGE_DOUBLES doubles: FF_NORM_TERMS sums of squares, the count of non-finite
     * values, FF_NORM_TERMS element counts, one spare), then the controller -- and `exchange(exchange_user, hip_stream)`
     * is called on the HOST in between: it must ENQUEUE, ordered with hip_stream, an in-place sum of those doubles over
     * all ranks (an RCCL all-reduce) and return 0; every rank then takes the same decisions and attempts the same number
     * of steps.  NULL: the norms are over this call's batch. */
    double* exchange_sums;
    int (*exchange)(void* user, void* hip_stream);
    void* exchange_user;
    /* Divergence by a Hutch++ / XTrace ESTIMATE instead of the exact trace (ScoreModel(hutchpp=True) / (xtrace=True),
     * diffusion.py:336-481; FF_MODE_EXACT only).  The state never depends on the divergence, so an attempted step stays
     * the fused launch(es) of the exact trace -- now recording the Jacobian of every evaluation row into est_jac -- followed
     * by ONE ff_trace_estimate launch over all rows and one launch that combines the rows' estimates with the attempt's
     * tail coefficients into aux_lp[0..3] (proposal, last stage, midpoint, error estimate), the same linear combinations
     * the fused kernel takes of its own divergence slots.  est_kind = 0: the exact trace. */
    int32_t est_kind;            /* 0 or FF_TRACE_*                                                          */
    int32_t est_r, est_m;        /* ff_trace_args.r / .m                                                     */
    int32_t est_reserved;
    const float* est_probes0;    /* ff_trace_args.probes0 / probes1                                          */
    const float* est_probes1;
    float* est_jac;              /* [n_stages - 1][B][D][D]                                                  */
    float* est_div;              /* [n_stages - 1][B]                                                        */
    float* est_workspace;        /* ff_trace_workspace_floats(est_kind, D, est_r, (n_stages - 1) * B) floats  */
} ff_adapt_buffers;
#define FF_EXCHANGE_DOUBLES 8

#define FF_ADAPT_START   1   /* initialise the state, evaluate f(t0, y), choose the first step              */
#define FF_ADAPT_FINISH  2   /* evaluate the dense output at t_end into out_y / out_lp (a no-op until done)  */

/*
 * Enqueue on hip_stream: [FF_ADAPT_START:] the start-up launches, then `n_attempts` attempted steps, then
 * [FF_ADAPT_FINISH:] the dense output.  `base` supplies cond / probe / wpack / mode / batch (its state, table, aux and
 * adaptive fields are ignored).  t0 / t_end are solver times (already negated for a decreasing span) and only read with
 * FF_ADAPT_START.  Nothing is read back: the caller copies *buffers->state to the host when it wants to know (done,
 * error, counters) and calls again with what = 0 or FF_ADAPT_FINISH while `done` and `error` are both zero.
 */
int ff_mlp_ode_adaptive(const ff_mlp_plan_t* plan, const ff_ode_args* base, const ff_adapt_config* config,
                        const ff_adapt_buffers* buffers, double t0, double t_end, int32_t what, int32_t n_attempts,
                        void* hip_stream);

/* The controller's arithmetic on the HOST (tests without a GPU): one evaluation row for real time `t_real` -- a_e, b_e
 * (already multiplied by config->sign) and c1[config->h_real] -- with emb_w / w0t / b0 read as HOST pointers. */
int ff_adapt_host_row(const ff_adapt_config* config, float t_real, float* a_out, float* b_out, float* c1_out);
/* ... and one controller transition: `phase` 1 = after the initial-step norms d0 (norms[0]) and d1 (norms[1]), 2 = after
 * d2 (norms[0] = the unscaled norm), 3 = after an attempted step (norms[0] = error ratio, norms[1] != 0: non-finite y1),
 * 4 = first_step given.  Updates *state; returns 1 if an attempt must follow. */
int ff_adapt_host_transition(const ff_adapt_config* config, ff_adapt_state* state, int32_t phase, const float* norms);

#ifdef __cplusplus
}
#endif
#endif /* FLOWFUSION_AMD_H */
