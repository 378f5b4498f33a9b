// ff_api.cpp -- C ABI of libflowfusion_amd.so (see include/flowfusion_amd.h).
#include <hip/hip_runtime_api.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include "flowfusion_amd.h"
#include "ff_layout.h"
#include "ff_split_layout.h"
#include "ff_registry.h"
#include "ff_adapt_logic.h"

static_assert(FF_MAX_SLOTS == ff::kSlots, "slot count mismatch between header and kernel");
static_assert(FF_MAX_AUX == ff::kAux, "aux count mismatch between header and kernel");
static_assert(FF_ROW_HDR * 4 == sizeof(ff::RowHdr), "row header mismatch");
static_assert(FF_STATUS_NAN == ff::kStatusNaN && FF_STATUS_BAD_SLOT == ff::kStatusBadSlot, "status bits mismatch");
static_assert(sizeof(ff_adapt_state) == 128, "controller state is 128 bytes");

static thread_local int t_last_hip_error = 0;

extern "C" const char* ff_version(void) { return "flowfusion_amd 0.4 gfx950 (f32 MFMA 16x16x4 / 32x32x2, opt-in bf16x3 / bf16x2 split on 16x16x32 bf16; in-register layer chaining; device-side adaptive step control; Hutch++ / XTrace estimator kernels)"; }

extern "C" int ff_kernel_count(void) { return ff::g_n_kernels + ff::g_n_split_kernels; }

// ids [0, n) are the fp32 instantiations (plan.kernel_id of an FF_PREC_F32 plan), [n, n + m) the split-precision ones
extern "C" const char* ff_kernel_name(int id)
{
    if (id < 0 || id >= ff::g_n_kernels + ff::g_n_split_kernels) return NULL;
    return id < ff::g_n_kernels ? ff::g_kernels[id].name : ff::g_split_kernels[id - ff::g_n_kernels].name;
}

extern "C" int ff_last_hip_error(void) { return t_last_hip_error; }

// tangent columns per sample for a mode; `tile` bounds how many unit tangents one launch can carry
static int tangents_of_mode(int mode, int dim, int tile, int* n_tangent, int* unit)
{
    switch (mode) {
    case FF_MODE_STATE: *n_tangent = 0; *unit = 0; return 0;
    case FF_MODE_HUTCH: *n_tangent = 1; *unit = 0; return 0;
    case FF_MODE_EXACT: *n_tangent = dim < tile - 1 ? dim : tile - 1; *unit = 1; return 0;
    default: return FF_ERR_BADARG;
    }
}

static inline int split_parts(int precision) { return precision == FF_PREC_BF16X2 ? 2 : 3; }

// FF_PREC_BF16X3 / FF_PREC_BF16X2: the split-precision family (ff_mlp_ode_split.hpp) -- SiLU, width <= 256, dim <= 16,
// cond_dim <= 16, state-only or Hutchinson, the number of hidden layers compiled in
static int plan_split(int dim, int cond_dim, int n_hidden, const int* hidden_widths, int mode, int activation,
                      int precision, ff_mlp_plan_t* plan)
{
    if (!plan || !hidden_widths || dim < 1 || cond_dim < 0 || n_hidden < 1) return FF_ERR_BADARG;
    if (mode != FF_MODE_STATE && mode != FF_MODE_HUTCH && mode != FF_MODE_EXACT) return FF_ERR_BADARG;
    if (activation < 0 || activation >= FF_ACT_COUNT) return FF_ERR_BADARG;
    if (activation != FF_ACT_SILU || dim > 32 || cond_dim > 16) return FF_ERR_UNSUPPORTED;
    const int need_dt = dim > 16 ? 2 : 1;              // 16-dimension tiles of the state
    int wmax = 0;
    for (int i = 0; i < n_hidden; ++i) {
        if (hidden_widths[i] < 1) return FF_ERR_BADARG;
        if (hidden_widths[i] > wmax) wmax = hidden_widths[i];
    }
    if (wmax > ff::split::kWidth) return FF_ERR_UNSUPPORTED;
    const int need_t = mode == FF_MODE_STATE ? 0 : (mode == FF_MODE_HUTCH ? 1 : 2);
    int best = -1;                                     // the narrowest instantiation that holds the network
    for (int i = 0; i < ff::g_n_split_kernels; ++i) {
        const ff::SplitKernelEntry& k = ff::g_split_kernels[i];
        if (k.n_hidden == n_hidden && k.tangents == need_t && k.parts == split_parts(precision) && k.dt == need_dt &&
            k.width >= wmax && (best < 0 || k.width < ff::g_split_kernels[best].width))
            best = i;
    }
    if (best < 0) return FF_ERR_UNSUPPORTED;
    memset(plan, 0, sizeof(*plan));
    plan->dim = dim;
    plan->cond_dim = cond_dim;
    plan->n_hidden = n_hidden;
    plan->width = ff::g_split_kernels[best].width;
    plan->dregs = 8 * need_dt;                         // 4 dimensions x 2 column blocks per lane and 16-dimension tile
    plan->cregs = cond_dim > 0 ? 8 : 0;
    plan->kernel_id = best;
    plan->tile = need_t == 2 ? 16 : 32;                // exact trace: a sample and its unit tangents share a column block of 16
    plan->activation = FF_ACT_SILU;
    plan->precision = precision;
    return FF_OK;
}

extern "C" int ff_mlp_plan(int dim, int cond_dim, int n_hidden, const int* hidden_widths, int mode,
                           ff_mlp_plan_t* plan)
{
    return ff_mlp_plan_act(dim, cond_dim, n_hidden, hidden_widths, mode, FF_ACT_SILU, NULL, plan);
}

extern "C" int ff_mlp_plan_act(int dim, int cond_dim, int n_hidden, const int* hidden_widths, int mode,
                               int activation, const float* act_param, ff_mlp_plan_t* plan)
{
    return ff_mlp_plan_prec(dim, cond_dim, n_hidden, hidden_widths, mode, activation, act_param, FF_PREC_F32, plan);
}

extern "C" int ff_mlp_plan_prec(int dim, int cond_dim, int n_hidden, const int* hidden_widths, int mode,
                                int activation, const float* act_param, int precision, ff_mlp_plan_t* plan)
{
    if (precision != FF_PREC_F32 && precision != FF_PREC_BF16X3 && precision != FF_PREC_BF16X2) return FF_ERR_BADARG;
    if (precision != FF_PREC_F32) return plan_split(dim, cond_dim, n_hidden, hidden_widths, mode, activation, precision, plan);
    if (!plan || !hidden_widths || dim < 1 || cond_dim < 0 || n_hidden < 1) return FF_ERR_BADARG;
    if (activation < 0 || activation >= FF_ACT_COUNT) return FF_ERR_BADARG;
    if (mode != FF_MODE_STATE && mode != FF_MODE_HUTCH && mode != FF_MODE_EXACT) return FF_ERR_BADARG;
    int wmax = 0;
    for (int i = 0; i < n_hidden; ++i) {
        if (hidden_widths[i] < 1) return FF_ERR_BADARG;
        if (hidden_widths[i] > wmax) wmax = hidden_widths[i];
    }
    const int need_t = mode != FF_MODE_STATE;
    // Preference among kernels that fit: narrowest width first (it dominates the FLOPs); at equal
    // width the 16x16x4 two-waves-per-SIMD kernels (measured ~3% faster than 32x32x2 at width 256),
    // then fewer first-layer k-steps.  FF_TILE=32|16 pins the tile (experiments).
    const char* pin = getenv("FF_TILE");
    const int pin_tile = pin ? atoi(pin) : 0;
    // FF_ACT_ANY=1 / 0 (A/B runs on a library built with FF_BUILD_FULL): only / never the run-time-choice instantiations
    const char* pin_any = getenv("FF_ACT_ANY");
    int best = -1;
    for (int i = 0; i < ff::g_n_kernels; ++i) {
        const ff::KernelEntry& k = ff::g_kernels[i];
        const int need_d = ff::regs_for(k.tile, dim);
        const int need_c = cond_dim > 0 ? ff::regs_for(k.tile, cond_dim) : 0;
        if (k.H < wmax || k.dregs < need_d || k.cregs < need_c || k.tangents != need_t ||
            !(k.act == activation || (k.act == 9 && activation != FF_ACT_SILU))) continue;
        if (pin_tile && k.tile != pin_tile) continue;
        if (pin_any && activation != FF_ACT_SILU && (k.act == 9) != (atoi(pin_any) != 0)) continue;
        if (best < 0) { best = i; continue; }
        const ff::KernelEntry& b = ff::g_kernels[best];
        const int kc = k.dregs * (64 / k.tile) + k.cregs * (64 / k.tile);     // first-layer features covered
        const int bc = b.dregs * (64 / b.tile) + b.cregs * (64 / b.tile);
        if (k.H < b.H || (k.H == b.H && (k.tile < b.tile || (k.tile == b.tile && kc < bc)))) best = i;
    }
    if (best < 0) return FF_ERR_UNSUPPORTED;
    memset(plan, 0, sizeof(*plan));
    plan->dim = dim;
    plan->cond_dim = cond_dim;
    plan->n_hidden = n_hidden;
    plan->width = ff::g_kernels[best].H;
    plan->dregs = ff::g_kernels[best].dregs;
    plan->cregs = ff::g_kernels[best].cregs;
    plan->kernel_id = best;
    plan->tile = ff::g_kernels[best].tile;
    plan->activation = activation;
    if (act_param) { plan->act_param[0] = act_param[0]; plan->act_param[1] = act_param[1]; }
    return FF_OK;
}

static bool plan_ok_split(const ff_mlp_plan_t* p)
{
    if (!p || (p->precision != FF_PREC_BF16X3 && p->precision != FF_PREC_BF16X2) || p->kernel_id < 0 ||
        p->kernel_id >= ff::g_n_split_kernels)
        return false;
    const ff::SplitKernelEntry& k = ff::g_split_kernels[p->kernel_id];
    return k.parts == split_parts(p->precision) && p->width == k.width && p->tile == (k.tangents == 2 ? 16 : 32) &&
           p->dregs == 8 * k.dt &&
           p->cregs == (p->cond_dim > 0 ? 8 : 0) &&
           p->n_hidden == k.n_hidden && p->activation == FF_ACT_SILU && p->dim >= 1 && p->dim <= 16 * k.dt &&
           p->cond_dim >= 0 && p->cond_dim <= 16;
}

static bool plan_ok(const ff_mlp_plan_t* p)
{
    if (!p || p->precision != FF_PREC_F32) return false;
    if (p->kernel_id < 0 || p->kernel_id >= ff::g_n_kernels) return false;
    const ff::KernelEntry& k = ff::g_kernels[p->kernel_id];
    const int per_reg = 64 / k.tile;
    return k.H == p->width && k.dregs == p->dregs && k.cregs == p->cregs && k.tile == p->tile && p->n_hidden >= 1 &&
           p->activation >= 0 && p->activation < FF_ACT_COUNT && (p->activation == k.act || (k.act == 9 && p->activation != FF_ACT_SILU)) &&
           p->dim >= 1 && p->dim <= per_reg * p->dregs && p->cond_dim >= 0 && p->cond_dim <= per_reg * p->cregs;
}

static ff::Layout plan_layout(const ff_mlp_plan_t* p)
{
    return ff::make_layout(p->tile, p->width, p->dregs, p->cregs, p->n_hidden);
}

extern "C" const char* ff_plan_kernel_name(const ff_mlp_plan_t* plan)
{
    if (plan_ok_split(plan)) return ff::g_split_kernels[plan->kernel_id].name;
    if (plan_ok(plan)) return ff::g_kernels[plan->kernel_id].name;
    return NULL;
}

extern "C" size_t ff_mlp_wpack_floats(const ff_mlp_plan_t* plan)
{
    if (plan_ok_split(plan)) {
        return ff::split::total_words(plan->n_hidden, split_parts(plan->precision), plan->dregs / 8, plan->width);
    }
    if (!plan_ok(plan)) return 0;
    return plan_layout(plan).total_floats;
}

// three-way bf16 split by truncation: v = hi + mid + lo exactly (8 + 8 + 8 significand bits)
static inline uint16_t bf16_top(float v, float* rest)
{
    uint32_t u;
    memcpy(&u, &v, 4);
    const uint32_t t = u & 0xFFFF0000u;
    float tf;
    memcpy(&tf, &t, 4);
    *rest = v - tf;
    return (uint16_t)(t >> 16);
}

// round-to-nearest-even bf16 of v (weights are finite), and what is left
static inline uint16_t bf16_rne(float v, float* rest)
{
    uint32_t u;
    memcpy(&u, &v, 4);
    const uint32_t t = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    float tf;
    memcpy(&tf, &t, 4);
    *rest = v - tf;
    return (uint16_t)(t >> 16);
}

// FF_PREC_BF16X3 / BF16X2 packing: the fragment stream of ff_split_layout.h followed by the fp32 biases.  Three parts:
// truncation (hi + mid + lo = the fp32 weight exactly); two parts: round to nearest (hi + mid = the weight to 16 bits)
static int wpack_split(const ff_mlp_plan_t* plan, const float* const* W, const float* const* b,
                       const int* hidden_widths, int in_features0, int x_col0, int c_col0, float* out)
{
    namespace sp = ff::split;
    const int D = plan->dim, C = plan->cond_dim, H = plan->width, NH = plan->n_hidden;
    const int NP = split_parts(plan->precision), DT = plan->dregs / 8;
    uint32_t* words = (uint32_t*)out;
    const int NR = sp::row_tiles(H), NS = sp::ksteps(H);
    memset(out, 0, sp::total_words(NH, NP, DT, H) * 4);
    size_t group = 0;                                  // running group index in the stream
    // one group: fragments [hi, mid, lo] of 16-row tile rt; element (quad q, j) multiplies input column col(q, j)
    auto put_group = [&](const float* Wl, int rows, int ld, int rt, auto col) {
        uint32_t* g = words + group * (NP * sp::kFragBytes / 4);
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int row = 16 * rt + (lane & 15), c = col(lane >> 4, j);
                const float v = (row < rows && c >= 0) ? Wl[(size_t)row * ld + c] : 0.f;
                float r1, r2, r3;
                uint16_t part[3];
                if (NP == 3) {
                    part[0] = bf16_top(v, &r1); part[1] = bf16_top(r1, &r2); part[2] = bf16_top(r2, &r3);
                } else {
                    part[0] = bf16_rne(v, &r1); part[1] = bf16_rne(r1, &r2); part[2] = 0;
                }
                for (int p = 0; p < NP; ++p) {
                    uint32_t& w = g[p * 256 + lane * 4 + (j >> 1)];
                    w = (j & 1) ? ((w & 0x0000FFFFu) | ((uint32_t)part[p] << 16)) : ((w & 0xFFFF0000u) | part[p]);
                }
            }
        ++group;
    };
    // layer 1: ONE k-step -- features 0..15 the state dimensions, 16..31 the conditional inputs; states of up to 32
    // dimensions: TWO k-steps -- features 0..31 the state, then 0..15 the conditional inputs
    for (int s1 = 0; s1 < DT; ++s1)
        for (int rt = 0; rt < NR; ++rt)
            put_group(W[0], hidden_widths[0], in_features0, rt, [&](int q, int j) {
                const int f = sp::kidx(0, q, j);
                if (DT == 1) return f < 16 ? (f < D ? x_col0 + f : -1) : (f - 16 < C ? c_col0 + f - 16 : -1);
                return s1 == 0 ? (f < D ? x_col0 + f : -1) : (f < C ? c_col0 + f : -1);
            });
    // hidden -> hidden, k-major: k-step s, row tile rt
    for (int l = 1; l < NH; ++l) {
        const int win = hidden_widths[l - 1], wout = hidden_widths[l];
        for (int s = 0; s < NS; ++s)
            for (int rt = 0; rt < NR; ++rt)
                put_group(W[l], wout, win, rt, [&](int q, int j) {
                    const int kk = sp::kidx(s, q, j);
                    return kk < win ? kk : -1;
                });
        float* bo = out + sp::stream_words(NH, NP, DT, H) + (size_t)(l - 1) * H;
        for (int row = 0; row < wout; ++row) bo[row] = b[l][row];
    }
    // output layer: DT row tiles (the state's dimensions), k-major
    {
        const int win = hidden_widths[NH - 1];
        for (int s = 0; s < NS; ++s)
            for (int t = 0; t < DT; ++t)
                put_group(W[NH], D, win, t, [&](int q, int j) {
                    const int kk = sp::kidx(s, q, j);
                    return kk < win ? kk : -1;
                });
        float* bo = out + sp::stream_words(NH, NP, DT, H) + (size_t)(NH - 1) * H;
        for (int row = 0; row < D; ++row) bo[row] = b[NH][row];
    }
    return group == (size_t)sp::granules_per_eval(NH, DT, H) * sp::granule_groups(H) ? FF_OK : FF_ERR_BADARG;
}

extern "C" int ff_mlp_wpack(const ff_mlp_plan_t* plan, const float* const* W, const float* const* b,
                            const int* hidden_widths, int in_features0, int x_col0, int c_col0, float* out)
{
    if (plan_ok_split(plan)) {
        if (!W || !b || !hidden_widths || !out) return FF_ERR_BADARG;
        const int D = plan->dim, C = plan->cond_dim, NH = plan->n_hidden;
        if (x_col0 < 0 || x_col0 + D > in_features0) return FF_ERR_BADARG;
        if (C > 0 && (c_col0 < 0 || c_col0 + C > in_features0)) return FF_ERR_BADARG;
        for (int i = 0; i < NH; ++i)
            if (hidden_widths[i] < 1 || hidden_widths[i] > plan->width) return FF_ERR_BADARG;
        for (int i = 0; i <= NH; ++i)
            if (!W[i] || (i > 0 && !b[i])) return FF_ERR_BADARG;
        return wpack_split(plan, W, b, hidden_widths, in_features0, x_col0, c_col0, out);
    }
    if (!plan_ok(plan) || !W || !b || !hidden_widths || !out) return FF_ERR_BADARG;
    const int D = plan->dim, C = plan->cond_dim, H = plan->width, NH = plan->n_hidden;
    if (x_col0 < 0 || x_col0 + D > in_features0) return FF_ERR_BADARG;
    if (C > 0 && (c_col0 < 0 || c_col0 + C > in_features0)) return FF_ERR_BADARG;
    for (int i = 0; i < NH; ++i)
        if (hidden_widths[i] < 1 || hidden_widths[i] > H) return FF_ERR_BADARG;
    for (int i = 0; i <= NH; ++i)
        if (!W[i] || (i > 0 && !b[i])) return FF_ERR_BADARG;
    const ff::Layout L = plan_layout(plan);
    const int TL = plan->tile, PHYS = ff::tile_phys(TL), CF = L.chunk_fl;
    memset(out, 0, L.total_floats * sizeof(float));

    // Fill the chunks of one layer in consumption order (ff_layout.h).  `kcol(r, h)` maps operand
    // register r on lane-half h to a column of `Wl` (or -1 for padding), rows beyond `rows` are zero.
    auto fill = [&](const ff::LayerGeom& G, float* o, const float* Wl, int rows, int ld, auto kcol) {
        for (int c = 0; c < G.NC; ++c) {
            const int g = ff::chunk_group(G, c), ob = ff::chunk_block(G, c);
            for (int p = 0; p < PHYS; ++p)
                for (int lane = 0; lane < 64; ++lane)
                    for (int q = 0; q < 4; ++q) {
                        const int row = ob * 32 + TL * p + (lane & (TL - 1));
                        const int col = kcol(4 * g + q, lane / TL);
                        o[(size_t)c * CF + ((size_t)p * 64 + lane) * 4 + q] =
                            (row < rows && col >= 0) ? Wl[(size_t)row * ld + col] : 0.f;
                    }
        }
    };
    // first layer: operand registers = [state | conditional]
    fill(L.g1, out, W[0], hidden_widths[0], in_features0, [&](int r, int h) {
        if (r < plan->dregs) {
            const int d = ff::feat_of_reg(TL, r, h);
            return d < D ? x_col0 + d : -1;
        }
        const int d = ff::feat_of_reg(TL, r - plan->dregs, h);
        return d < C ? c_col0 + d : -1;
    });
    // hidden -> hidden
    for (int l = 1; l < NH; ++l) {
        const int win = hidden_widths[l - 1], wout = hidden_widths[l];
        fill(L.gh, out + (size_t)L.chunk_off_hid(l - 1) * CF, W[l], wout, win, [&](int r, int h) {
            const int k = ff::feat_of_reg(TL, r, h);
            return k < win ? k : -1;
        });
        float* bo = out + L.bias_off_hid(l - 1);
        for (int row = 0; row < wout; ++row) bo[row] = b[l][row];
    }
    // output layer
    {
        const int win = hidden_widths[NH - 1];
        fill(L.go, out + (size_t)L.chunk_off_out() * CF, W[NH], D, win, [&](int r, int h) {
            const int k = ff::feat_of_reg(TL, r, h);
            return k < win ? k : -1;
        });
        float* bo = out + L.bias_off_out();
        for (int row = 0; row < D; ++row) bo[row] = b[NH][row];
    }
    return FF_OK;
}

extern "C" int ff_mlp_samples_per_workgroup(const ff_mlp_plan_t* plan, int mode)
{
    if (plan_ok_split(plan)) {
        const int kt = ff::g_split_kernels[plan->kernel_id].tangents;
        if (mode == FF_MODE_STATE && kt == 0) return 128;
        if (mode == FF_MODE_HUTCH && kt == 1) return 64;
        if (mode == FF_MODE_EXACT && kt == 2) return 8 * (16 / (1 + (plan->dim < 15 ? plan->dim : 15)));
        return FF_ERR_BADARG;
    }
    if (!plan_ok(plan)) return FF_ERR_BADARG;
    int nt, unit;
    int rc = tangents_of_mode(mode, plan->dim, plan->tile, &nt, &unit);
    if (rc) return rc;
    const bool wide = ff::g_kernels[plan->kernel_id].launch == nullptr;      // a tile per workgroup
    return (wide ? 1 : 4) * (plan->tile / (1 + nt));
}

// FF_PREC_BF16X3 / BF16X2 launch: state-only (Euler-Maruyama noise rows included) / Hutchinson integration of a table
static int launch_split(const ff_mlp_plan_t* plan, const ff_ode_args* a, void* hip_stream)
{
    if (!a->x_in || !a->x_out || !a->wpack || !a->etab || a->batch < 0 || a->n_evals < 0) return FF_ERR_BADARG;
    if (plan->cond_dim > 0 && !a->cond) return FF_ERR_BADARG;
    const ff::SplitKernelEntry& k = ff::g_split_kernels[plan->kernel_id];
    if (a->mode != FF_MODE_STATE && a->mode != FF_MODE_HUTCH && a->mode != FF_MODE_EXACT) return FF_ERR_BADARG;
    if (k.tangents != (a->mode == FF_MODE_STATE ? 0 : (a->mode == FF_MODE_HUTCH ? 1 : 2))) return FF_ERR_BADARG;
    if (a->mode == FF_MODE_HUTCH && !a->probe) return FF_ERR_BADARG;
    if (a->mode != FF_MODE_STATE && !a->dlogp_out) return FF_ERR_BADARG;
    int nt = k.tangents == 1 ? 1 : 0, tfirst = 0;
    if (a->mode == FF_MODE_EXACT) {                     // unit tangents of dimensions [tfirst, tfirst + nt): at most 15 per launch
        nt = plan->dim < 15 ? plan->dim : 15;
        tfirst = a->tangent_first;
        if (a->tangent_count > 0) nt = a->tangent_count;
        else if (plan->dim > nt) return FF_ERR_BADARG;      // must be split by the caller
        if (tfirst < 0 || tfirst + nt > plan->dim || nt + 1 > 16) return FF_ERR_BADARG;
    }
    // what this family does not carry: the Jacobian output; noise rows with tangent columns
    if (a->jac_out || (a->noise && k.tangents)) return FF_ERR_UNSUPPORTED;
    if (a->noise && a->noise_stride < a->batch * (int64_t)plan->dim) return FF_ERR_BADARG;
    if (a->n_aux < 0 || a->n_aux > FF_MAX_AUX) return FF_ERR_BADARG;
    // stage slots: the table's promise must fit what the kernel keeps on chip (a row naming a slot beyond it would
    // land on the parked stage input and the state: the kernel refuses such a row, FF_STATUS_BAD_SLOT)
    if (a->stage_slots < 0 || a->stage_slots > FF_MAX_SLOTS) return FF_ERR_BADARG;
    if (a->stage_slots > ff::split::slots_on_chip(k.dt)) return FF_ERR_UNSUPPORTED;
    if (a->batch == 0) return FF_OK;
    ff::KernelArgs ka;
    memset(&ka, 0, sizeof(ka));
    ka.gate = a->gate;
    ka.x_in = a->x_in; ka.x_out = a->x_out; ka.cond = a->cond; ka.probe = a->probe;
    ka.dlogp_out = a->dlogp_out; ka.wpack = a->wpack; ka.etab = a->etab;
    ka.in_shift = a->in_shift; ka.in_scale = a->in_scale; ka.out_scale = a->out_scale; ka.out_shift = a->out_shift;
    ka.status = a->status; ka.batch = a->batch; ka.dlogp_in = a->dlogp_in;
    ka.k1_in = a->k1_in; ka.kl1_in = a->kl1_in; ka.n_aux = a->n_aux;
    ka.noise = a->noise; ka.noise_stride = a->noise_stride;
    ka.rng_seed = a->rng_seed; ka.rng_sample_offset = a->rng_sample_offset; ka.rng_noise_base = a->rng_noise_base;
    for (int j = 0; j < FF_MAX_AUX; ++j) { ka.aux_out[j] = a->aux_out[j]; ka.aux_lp_out[j] = a->aux_lp_out[j]; }
    ka.n_evals = a->n_evals; ka.n_hidden = plan->n_hidden; ka.dim = plan->dim; ka.cond_dim = plan->cond_dim;
    ka.n_tangent = nt; ka.unit_tangents = a->mode == FF_MODE_EXACT ? 1 : 0; ka.tangent_first = tfirst;
    ka.etab_stride = FF_ROW_HDR + plan->width;
    if ((size_t)(a->n_evals + 2) * ka.etab_stride * 4 > 0x7fffffffull) return FF_ERR_UNSUPPORTED;
    ka.wpack_floats = (int)ff::split::total_words(k.n_hidden, k.parts, k.dt, k.width);
    const long long spw = k.tangents == 0 ? 128 : (k.tangents == 1 ? 64 : 8 * (16 / (1 + nt)));
    const long long grid = (a->batch + spw - 1) / spw;
    if (grid > 0x7fffffffll) return FF_ERR_UNSUPPORTED;
    // the caller may say how many stage slots the table uses (ff_ode_args.stage_slots): up to four, the twin that keeps
    // four slots on chip and shares a CU between two workgroups serves the launch (same arithmetic, same results)
    const bool four = k.launch4 && a->stage_slots >= 1 && a->stage_slots <= 4 && !getenv("FF_SPLIT_NO_TWIN");
    const unsigned lds = (unsigned)ff::split::lds_map(plan->width, plan->n_hidden, k.parts, k.dt, four ? 4 : 0).total;
    const int herr = (four ? k.launch4 : k.launch)(&ka, (unsigned)grid, lds, (hipStream_t)hip_stream);
    if (herr != 0) { t_last_hip_error = herr; return FF_ERR_HIP; }
    return FF_OK;
}

// Which kernel(s) serve a launch of `tiles` tiles (16 or 32 columns each) of an f32 plan.
//
// Small batches: when the tiles of the batch would leave at least half the chip's 1024 SIMDs without one, the
// cooperative twin (one tile per WORKGROUP, the layer's rows split over its four wavefronts) finishes an evaluation
// in about a third of the time.  Same packed weights, bitwise the same results.  FF_COOP=0 / 1 pins the choice.
// One rule for whole launches and for tails (below), fitted to measurements at widths 128 / 256 / 512 (scratch/tail_split.py,
// scratch/tail_margin.py, scratch/coop_threshold.py; in units of a full round of the one-wavefront kernel):
//   one-wavefront kernel, n tiles:  ceil(n / 1024) / wps      (n tiles run with that many wavefronts per SIMD; a wavefront
//                                                              does not finish sooner for having fewer neighbours)
//   twin, n tiles:                  max(least, c0 + n / (0.95 chip)),  chip = 1024 wps tiles in flight,
//                                   c0 = 0.10 (0.20 at one wavefront per SIMD), least = 0.22 / 0.30 / 0.35 for wps = 3 / 2 / 1
// The twin serves whatever it is faster at.
// The tail of a launch.  The chip runs 1024 * wps tiles at once; the tiles left over after the full rounds run as a last
// round with w = ceil(leftover / 1024) wavefronts per SIMD, which takes w / wps of a full round's time (measured: the
// dispatcher fills SIMDs evenly, a wavefront does not finish sooner for having fewer neighbours than wps allows).
// The cooperative twin -- a tile per workgroup, bitwise the same results -- gets through about 0.8-0.9 of a chip's worth
// of tiles in a round's time and never needs less than ~0.3 of it.  Whenever that is the shorter of the two, the
// leftover rows go to the twin as a second launch (scratch/tail_split.py: up to +34 % just above a whole number of
// rounds, +5 % at eight rounds).  FF_TAIL_SPLIT=0 switches it off (A/B runs, tests).
struct LaunchChoice {
    bool coop;                 // the main launch is the cooperative twin (or the wide catch-all)
    long long main_tiles;      // tiles of the main launch
    long long tail_tiles;      // tiles of a second launch on the twin (0 = none)
};

static LaunchChoice choose_launch(const ff::KernelEntry& k, long long tiles, bool jac_out)
{
    const long long chip = 1024ll * (k.wps > 0 ? k.wps : 1);
    auto twin_wins = [&](long long n) {
        const int wps = k.wps > 0 ? k.wps : 1;
        const double one_wave = (double)((n + 1023) / 1024) / (double)wps;
        const double least = wps >= 3 ? 0.22 : (wps == 2 ? 0.30 : 0.35);
        const double line = (wps == 1 ? 0.20 : 0.10) + n / (0.95 * (double)chip);
        return (line > least ? line : least) < one_wave;
    };
    bool coop = k.launch_coop != nullptr && tiles <= chip && twin_wins(tiles);
    if (const char* pin = getenv("FF_COOP")) coop = k.launch_coop != nullptr && atoi(pin) != 0;
    if (k.launch == nullptr) coop = true;                  // wide catch-all: cooperative at every batch size, one exchange buffer
    long long tail_tiles = 0;
    if (!coop && k.launch_coop != nullptr && k.wps > 0 && !jac_out) {
        const long long rem = tiles % chip;
        const char* pin = getenv("FF_TAIL_SPLIT");
        if (tiles > chip && rem > 0 && !(pin && atoi(pin) == 0)) {
            bool split = twin_wins(rem);
            if (const char* m = getenv("FF_TAIL_MAX")) split = rem <= atoll(m);          // (experiments: scratch/tail_split.py)
            if (split) tail_tiles = rem;
        }
    }
    return LaunchChoice{coop, tiles - tail_tiles, tail_tiles};
}

// What ff_mlp_ode_launch would enqueue for `batch` samples in `mode`: FF_LAUNCH_* (see the header).
extern "C" int ff_mlp_launch_kind(const ff_mlp_plan_t* plan, int64_t batch, int32_t mode, int32_t tangent_count, int32_t jac_out)
{
    if (plan_ok_split(plan)) return FF_LAUNCH_ONE_WAVE;
    if (!plan_ok(plan) || batch < 0) return FF_ERR_BADARG;
    int nt, unit;
    const int rc = tangents_of_mode(mode, plan->dim, plan->tile, &nt, &unit);
    if (rc) return rc;
    if (mode == FF_MODE_EXACT && tangent_count > 0) nt = tangent_count;
    if (nt + 1 > plan->tile) return FF_ERR_BADARG;
    const long long spt = plan->tile / (1 + nt);
    const LaunchChoice ch = choose_launch(ff::g_kernels[plan->kernel_id], (batch + spt - 1) / spt, jac_out != 0);
    return ch.coop ? FF_LAUNCH_TWIN : (ch.tail_tiles ? FF_LAUNCH_ONE_WAVE_AND_TWIN : FF_LAUNCH_ONE_WAVE);
}

extern "C" int ff_mlp_ode_launch(const ff_mlp_plan_t* plan, const ff_ode_args* a, void* hip_stream)
{
    if (a && plan_ok_split(plan)) return launch_split(plan, a, hip_stream);
    if (!plan_ok(plan) || !a) return FF_ERR_BADARG;
    if (!a->x_in || !a->x_out || !a->wpack || !a->etab || a->batch < 0 || a->n_evals < 0) return FF_ERR_BADARG;
    if (plan->cond_dim > 0 && !a->cond) return FF_ERR_BADARG;
    const ff::KernelEntry& k = ff::g_kernels[plan->kernel_id];
    int nt, unit;
    int rc = tangents_of_mode(a->mode, plan->dim, plan->tile, &nt, &unit);
    if (rc) return rc;
    int tfirst = 0;
    if (a->mode == FF_MODE_EXACT) {
        tfirst = a->tangent_first;
        if (a->tangent_count > 0) nt = a->tangent_count;
        else if (plan->dim > nt) return FF_ERR_BADARG;      // must be split by the caller
        if (tfirst < 0 || tfirst + nt > plan->dim || nt + 1 > plan->tile) return FF_ERR_BADARG;
    }
    if ((a->mode != FF_MODE_STATE) != (k.tangents != 0)) return FF_ERR_BADARG;
    if (a->mode == FF_MODE_HUTCH && !a->probe) return FF_ERR_BADARG;
    if (a->mode != FF_MODE_STATE && !a->dlogp_out) return FF_ERR_BADARG;
    if (a->stage_slots < 0 || a->stage_slots > FF_MAX_SLOTS) return FF_ERR_BADARG;
    if (a->batch == 0) return FF_OK;

    ff::KernelArgs ka;
    memset(&ka, 0, sizeof(ka));
    ka.x_in = a->x_in; ka.x_out = a->x_out; ka.cond = a->cond; ka.probe = a->probe;
    ka.dlogp_out = a->dlogp_out; ka.noise = a->noise; ka.wpack = a->wpack; ka.etab = a->etab;
    ka.in_shift = a->in_shift; ka.in_scale = a->in_scale; ka.out_scale = a->out_scale; ka.out_shift = a->out_shift;
    ka.status = a->status; ka.batch = a->batch; ka.noise_stride = a->noise_stride;
    ka.n_evals = a->n_evals; ka.n_hidden = plan->n_hidden; ka.dim = plan->dim; ka.cond_dim = plan->cond_dim;
    ka.n_tangent = nt; ka.unit_tangents = unit; ka.tangent_first = tfirst;
    if (a->n_aux < 0 || a->n_aux > FF_MAX_AUX) return FF_ERR_BADARG;
    ka.k1_in = a->k1_in; ka.kl1_in = a->kl1_in; ka.dlogp_in = a->dlogp_in; ka.n_aux = a->n_aux;
    for (int j = 0; j < FF_MAX_AUX; ++j) { ka.aux_out[j] = a->aux_out[j]; ka.aux_lp_out[j] = a->aux_lp_out[j]; }
    ka.rng_seed = a->rng_seed; ka.rng_sample_offset = a->rng_sample_offset; ka.rng_noise_base = a->rng_noise_base;
    if (a->jac_out && a->mode != FF_MODE_EXACT) return FF_ERR_BADARG;
    ka.jac_out = a->jac_out;
    ka.jac_all = a->jac_out && a->jac_all ? 1 : 0;
    ka.act_kind = plan->activation; ka.act_p0 = plan->act_param[0]; ka.act_p1 = plan->act_param[1];
    ka.gate = a->gate;
    ka.etab_stride = FF_ROW_HDR + plan->width;
    const ff::Layout L = plan_layout(plan);
    if (L.total_floats * 4 > 0x7fffffffull) return FF_ERR_UNSUPPORTED;
    if ((size_t)(a->n_evals + 2) * ka.etab_stride * 4 > 0x7fffffffull) return FF_ERR_UNSUPPORTED;
    ka.wpack_floats = (int)L.total_floats;

    const long long spt = plan->tile / (1 + nt);           // samples per tile
    const LaunchChoice ch = choose_launch(k, (a->batch + spt - 1) / spt, a->jac_out != nullptr);
    const bool coop = ch.coop, wide = k.launch == nullptr;
    const long long tiles = ch.main_tiles + ch.tail_tiles, tail_tiles = ch.tail_tiles;
    const unsigned slots = ff::kSlots * (plan->dregs / 4) * 64 * 16;
    const unsigned kh = (plan->width / 32) * ff::tile_rb(plan->tile);                 // operand registers of a hidden layer
    const unsigned lds_coop = slots + (wide ? 1u : 2u) * (kh / 4) * 64 * 16, lds_wave = 4u * slots;
    if ((coop || tail_tiles ? lds_coop : 0u) > 160u * 1024u || (!coop ? lds_wave : 0u) > 160u * 1024u) return FF_ERR_UNSUPPORTED;
    const long long main_tiles = tiles - tail_tiles;
    const long long grid = coop ? main_tiles : (main_tiles + 3) / 4;
    if (grid > 0x7fffffffll) return FF_ERR_UNSUPPORTED;
    if (tail_tiles) ka.batch = main_tiles * spt;                                       // (full tiles only: < a->batch)
    int herr = (coop ? k.launch_coop : k.launch)(&ka, (unsigned)grid, coop ? lds_coop : lds_wave, (hipStream_t)hip_stream);
    if (herr == 0 && tail_tiles) {
        // the same launch over rows [row0, batch): every per-row array moves on by row0 rows, the counter-based noise by row0 samples
        const long long row0 = main_tiles * spt, D = plan->dim, C = plan->cond_dim;
        ff::KernelArgs t = ka;
        t.batch = a->batch - row0;
        t.x_in += row0 * D; t.x_out += row0 * D;
        if (t.cond) t.cond += row0 * C;
        if (t.probe) t.probe += row0 * D;
        if (t.dlogp_out) t.dlogp_out += row0;
        if (t.dlogp_in) t.dlogp_in += row0;
        if (t.noise) t.noise += row0 * D;
        if (t.k1_in) t.k1_in += row0 * D;
        if (t.kl1_in) t.kl1_in += row0;
        for (int j = 0; j < FF_MAX_AUX; ++j) {
            if (t.aux_out[j]) t.aux_out[j] += row0 * D;
            if (t.aux_lp_out[j]) t.aux_lp_out[j] += row0;
        }
        t.rng_sample_offset += row0;
        herr = k.launch_coop(&t, (unsigned)tail_tiles, lds_coop, (hipStream_t)hip_stream);
    }
    if (herr != 0) { t_last_hip_error = herr; return FF_ERR_HIP; }
    return FF_OK;
}

// ---- the device-side adaptive controller's arithmetic, on the host (tests without a GPU; ff_adapt_logic.h) ------------
extern "C" int ff_adapt_host_row(const ff_adapt_config* c, float t_real, float* a_out, float* b_out, float* c1_out)
{
    if (!c || !a_out || !b_out || !c1_out || !c->w0t || !c->b0 || c->h_real < 1 || c->n_tcols < 1 || c->n_tcols > 64) return FF_ERR_BADARG;
    if (c->sched != FF_SCHED_FLOW && (!c->emb_w || c->n_tcols != 2 * c->n_emb)) return FF_ERR_BADARG;
    float a, b, feat[64];
    ff::adapt::schedule_ab(*c, t_real, &a, &b);
    *a_out = c->sign * a;
    *b_out = c->sign * b;
    for (int k = 0; k < c->n_tcols; ++k) feat[k] = ff::adapt::time_feature(*c, t_real, k);
    for (int h = 0; h < c->h_real; ++h) c1_out[h] = ff::adapt::c1_from_features(*c, feat, h);
    return FF_OK;
}

extern "C" int ff_adapt_host_transition(const ff_adapt_config* c, ff_adapt_state* s, int32_t phase, const float* norms)
{
    if (!c || !s || (phase != ff::adapt::kPhaseFirst && !norms)) return FF_ERR_BADARG;
    switch (phase) {
    case ff::adapt::kPhaseInit1:
        s->d0 = (double)norms[0];
        return ff::adapt::transition(*c, *s, phase, norms + 1, 1) == ff::adapt::kRowsDerivAtH0 ? 1 : 0;
    case ff::adapt::kPhaseInit2:
        return ff::adapt::transition(*c, *s, phase, norms, 1) == ff::adapt::kRowsAttempt ? 1 : 0;
    case ff::adapt::kPhaseStep:
        return ff::adapt::transition(*c, *s, phase, norms, 2) == ff::adapt::kRowsAttempt ? 1 : 0;
    case ff::adapt::kPhaseFirst:
        return ff::adapt::transition(*c, *s, phase, norms, 0) == ff::adapt::kRowsAttempt ? 1 : 0;
    default:
        return FF_ERR_BADARG;
    }
}
