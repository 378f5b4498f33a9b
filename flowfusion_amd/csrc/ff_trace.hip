// ff_trace.hip -- Hutch++ / XTrace divergence estimates from recorded Jacobians, one launch for every (evaluation row,
// sample) of a fused launch (gfx950).
//
// The reference evaluates the two estimators inside ScoreModel.forward (flowfusion/diffusion.py:336-481), i.e. once per
// right-hand-side evaluation of the solver, through 2r + m reverse-mode products and a batched `torch.linalg.qr`.  On the
// fused path a launch in FF_MODE_EXACT with ff_ode_args.jac_all = 1 leaves A = J^T of EVERY evaluation row in memory
// ([n_rows][batch][D][D]); this kernel turns them into the estimates [n_rows][batch] in one pass, so that an attempted
// step of the adaptive solver stays on the device (ff_adaptive.hip) and a fixed-grid solve needs no torch linear algebra.
//
// Roofline: HBM.  A work item reads its D x D matrix (three times at most; the second and third pass hit L2) and keeps
// its D x r factor in a workspace laid out item-fastest, so consecutive lanes touch consecutive words.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flowfusion_amd.h"
#include "ff_trace_est.h"

namespace ff {

__global__ __launch_bounds__(256) void trace_estimate_kernel(const ff_trace_args a, long long items)
{
    if (a.gate && *(const volatile int32_t*)a.gate == 0) return;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= items) return;
    const long long b = t % a.batch;
    trace::Item it;
    it.A = a.jac + (size_t)t * a.dim * a.dim;
    it.p0 = a.probes0 + (size_t)b * a.dim;
    it.p1 = a.probes1 ? a.probes1 + (size_t)b * a.dim : nullptr;
    it.pstride = (size_t)a.batch * a.dim;
    it.ws = a.workspace + t;
    it.stride = (size_t)items;
    it.D = a.dim; it.r = a.r; it.m = a.m;
    a.out[t] = trace::estimate(a.kind, it);
}

} // namespace ff

static int trace_args_ok(const ff_trace_args* a)
{
    if (!a || !a->jac || !a->probes0 || !a->out || !a->workspace) return FF_ERR_BADARG;
    if (a->kind != FF_TRACE_HUTCHPP && a->kind != FF_TRACE_XTRACE) return FF_ERR_BADARG;
    if (a->dim < 1 || a->n_rows < 0 || a->batch < 0 || a->r < 1 || a->r > a->dim) return FF_ERR_BADARG;
    if (a->kind == FF_TRACE_HUTCHPP && (a->m < 1 || !a->probes1)) return FF_ERR_BADARG;
    return FF_OK;
}

extern "C" size_t ff_trace_workspace_floats(int32_t kind, int32_t dim, int32_t r, int64_t items)
{
    if (dim < 1 || r < 1 || items < 0) return 0;
    return ff::trace::workspace_per_item(kind, dim, r) * (size_t)items;
}

extern "C" int ff_trace_estimate(const ff_trace_args* a, void* hip_stream)
{
    const int rc = trace_args_ok(a);
    if (rc != FF_OK) return rc;
    const long long items = (long long)a->n_rows * a->batch;
    if (items == 0) return FF_OK;
    const long long grid = (items + 255) / 256;
    if (grid > 0x7fffffffll) return FF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ff::trace_estimate_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)hip_stream, *a, items);
    return hipGetLastError() == hipSuccess ? FF_OK : FF_ERR_HIP;
}

// the same arithmetic on the host, HOST pointers (tests without a GPU); `workspace` needs workspace_per_item floats only
extern "C" int ff_trace_estimate_host(const ff_trace_args* a)
{
    const int rc = trace_args_ok(a);
    if (rc != FF_OK) return rc;
    const long long items = (long long)a->n_rows * a->batch;
    for (long long t = 0; t < items; ++t) {
        const long long b = t % a->batch;
        ff::trace::Item it;
        it.A = a->jac + (size_t)t * a->dim * a->dim;
        it.p0 = a->probes0 + (size_t)b * a->dim;
        it.p1 = a->probes1 ? a->probes1 + (size_t)b * a->dim : nullptr;
        it.pstride = (size_t)a->batch * a->dim;
        it.ws = a->workspace;
        it.stride = 1;
        it.D = a->dim; it.r = a->r; it.m = a->m;
        a->out[t] = ff::trace::estimate(a->kind, it);
    }
    return FF_OK;
}
