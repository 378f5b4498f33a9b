// ff_aux.hip -- small memory-bound kernels beside the fused integrator (gfx950).
//
//   ff_normal_fill     standard normals of the library's counter-based stream (ff_philox.h) for a block of
//                      global rows: the PRIOR draw of the sharded Euler-Maruyama sampler
//                      (reference: `self.sde.prior(dims).sample([batch])`, flowfusion/diffusion.py:532-536),
//                      keyed by the global row so that 1-, 2-, 4- and 8-GPU runs of a seed start from the same
//                      points without any rank materialising the whole batch.
//   ff_stage_combine   out = x_coef * x + sum_s coef[s] * k[s]  over flat fp32 arrays: the stage-input /
//                      step-update / error-estimate algebra of an explicit Runge-Kutta step for right-hand sides
//                      the fused kernel cannot hold (an arbitrary `model=` module of ScoreModel,
//                      flowfusion/diffusion.py:201,233-238): torchdiffeq performs it as one torch op per term
//                      (rk_common.py `_runge_kutta_step`), here it is one pass over HBM.
//
//   ff_scaled_rms      the error / step-size norms of an adaptive step (torchdiffeq: `_compute_error_ratio`,
//                      `_select_initial_step`; ~10 elementwise and reduction launches and 2-4 host syncs per attempted
//                      step when written in torch ops) as ONE deterministic reduction launch and one small read-back.
//
// All are pure streaming kernels (roofline: HBM): 16-byte accesses, consecutive lanes on consecutive
// addresses, grid sized to a few workgroups per CU, no LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flowfusion_amd.h"
#include "ff_philox.h"
#include "ff_norm.h"

namespace ff {

// one thread per (row, block of 4 dimensions): counter = (global row lo/hi, noise index, dim / 4)
__global__ __launch_bounds__(256) void normal_fill_kernel(float* __restrict__ out, long long batch, int dim, int nblk,
                                                          unsigned long long seed, long long sample_offset,
                                                          uint32_t noise_index, float scale)
{
    const long long total = batch * nblk;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / nblk;
        const int blk = (int)(i - row * nblk);
        const unsigned long long gs = (unsigned long long)(row + sample_offset);
        uint32_t c[4] = {(uint32_t)gs, (uint32_t)(gs >> 32), noise_index, (uint32_t)blk};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        float z[4];
        box_muller(c[0], c[1], z[0], z[1]);
        box_muller(c[2], c[3], z[2], z[3]);
        float* o = out + row * dim + 4 * blk;
        if (4 * blk + 4 <= dim && (dim & 3) == 0) {
            *(f32x4a*)o = f32x4a{z[0] * scale, z[1] * scale, z[2] * scale, z[3] * scale};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * blk + j < dim) o[j] = z[j] * scale;
        }
    }
}

struct CombineArgs {
    const float* x;
    const float* k[FF_MAX_SLOTS];
    float coef[FF_MAX_SLOTS];
    float x_coef;
    float* out;
    long long n;
    int vec_ok;      // every pointer 16-byte aligned: the body runs on float4
};

__global__ __launch_bounds__(256) void stage_combine_kernel(const CombineArgs a)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n4 = a.vec_ok ? a.n / 4 : 0;
    // two 16-byte elements per thread and trip: every array's two loads are in flight before the first is used (a one-term
    // combine otherwise waits on two loads per trip)
    auto combine = [&](long long i) __attribute__((always_inline)) {
        f32x4a v = a.x ? a.x_coef * ((const f32x4a*)a.x)[i] : f32x4a{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < FF_MAX_SLOTS; ++s)
            if (a.k[s]) v += a.coef[s] * ((const f32x4a*)a.k[s])[i];
        return v;
    };
    long long i = tid;
    for (; i + stride < n4; i += 2 * stride) {
        const f32x4a v0 = combine(i), v1 = combine(i + stride);
        ((f32x4a*)a.out)[i] = v0;
        ((f32x4a*)a.out)[i + stride] = v1;
    }
    if (i < n4) ((f32x4a*)a.out)[i] = combine(i);
    for (long long j = 4 * n4 + tid; j < a.n; j += stride) {     // unaligned input or the last n % 4 elements
        float v = a.x ? a.x_coef * a.x[j] : 0.f;
#pragma unroll
        for (int s = 0; s < FF_MAX_SLOTS; ++s)
            if (a.k[s]) v = __builtin_fmaf(a.coef[s], a.k[s][j], v);
        a.out[j] = v;
    }
}

__global__ __launch_bounds__(256) void scaled_rms_kernel(const NormArgs a)
{
    __shared__ double sh[4];
    __shared__ bool last;
    __shared__ float res[FF_NORM_TERMS + 1];
    scaled_rms_reduce(a, sh, &last, res);
}

static unsigned stream_grid(long long work_items)
{
    // a few workgroups per CU (256 CUs) is enough to saturate HBM with 16-byte accesses; never more than needed
    const long long want = (work_items + 255) / 256;
    const long long cap = 256 * 8;
    return (unsigned)(want < 1 ? 1 : (want > cap ? cap : want));
}

} // namespace ff

extern "C" int ff_normal_fill(float* out, int64_t batch, int32_t dim, uint64_t seed, int64_t sample_offset,
                              uint32_t noise_index, float scale, void* hip_stream)
{
    if (!out || batch < 0 || dim < 1) return FF_ERR_BADARG;
    if (batch == 0) return FF_OK;
    const int nblk = (dim + 3) / 4;
    hipLaunchKernelGGL(ff::normal_fill_kernel, dim3(ff::stream_grid(batch * nblk)), dim3(256), 0, (hipStream_t)hip_stream,
                       out, (long long)batch, dim, nblk, (unsigned long long)seed, (long long)sample_offset, noise_index, scale);
    return hipGetLastError() == hipSuccess ? FF_OK : FF_ERR_HIP;
}

extern "C" size_t ff_scaled_rms_workspace_bytes(void)
{
    return 16 + (size_t)ff::kNormBlocks * (FF_NORM_TERMS + 1) * sizeof(double);
}

extern "C" int ff_scaled_rms(const ff_norm_term* terms, int32_t n_terms, float atol, float rtol, const float* check,
                             int64_t n_check, float* out, void* workspace, void* hip_stream)
{
    if (!terms || n_terms < 1 || n_terms > FF_NORM_TERMS || !out || !workspace || n_check < 0 || (n_check > 0 && !check))
        return FF_ERR_BADARG;
    ff::NormArgs k;
    const unsigned grid = ff::norm_args_from_terms(k, terms, n_terms, atol, rtol, check, n_check, out, workspace);
    if (grid == 0) return FF_ERR_BADARG;
    hipLaunchKernelGGL(ff::scaled_rms_kernel, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, k);
    return hipGetLastError() == hipSuccess ? FF_OK : FF_ERR_HIP;
}

extern "C" int ff_stage_combine(const ff_combine_args* a, void* hip_stream)
{
    if (!a || !a->out || a->n < 0) return FF_ERR_BADARG;
    if (a->n == 0) return FF_OK;
    ff::CombineArgs k;
    k.x = a->x; k.x_coef = a->x_coef; k.out = a->out; k.n = a->n;
    uintptr_t bits = (uintptr_t)a->out | (uintptr_t)a->x;
    for (int s = 0; s < FF_MAX_SLOTS; ++s) {
        k.k[s] = a->coef[s] != 0.f ? a->k[s] : nullptr;       // a zero coefficient never reads its array
        k.coef[s] = a->coef[s];
        bits |= (uintptr_t)k.k[s];
    }
    k.vec_ok = (bits & 15) == 0;
    hipLaunchKernelGGL(ff::stage_combine_kernel, dim3(ff::stream_grid((a->n + 3) / 4)), dim3(256), 0,
                       (hipStream_t)hip_stream, k);
    return hipGetLastError() == hipSuccess ? FF_OK : FF_ERR_HIP;
}
