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

namespace ff {

typedef float f32x4a __attribute__((ext_vector_type(4)));

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
    for (long long i = tid; i < n4; i += stride) {
        f32x4a v = a.x ? a.x_coef * ((const f32x4a*)a.x)[i] : f32x4a{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < FF_MAX_SLOTS; ++s)
            if (a.k[s]) v += a.coef[s] * ((const f32x4a*)a.k[s])[i];
        ((f32x4a*)a.out)[i] = v;
    }
    for (long long i = 4 * n4 + tid; i < a.n; i += stride) {     // unaligned input or the last n % 4 elements
        float v = a.x ? a.x_coef * a.x[i] : 0.f;
#pragma unroll
        for (int s = 0; s < FF_MAX_SLOTS; ++s)
            if (a.k[s]) v = __builtin_fmaf(a.coef[s], a.k[s][i], v);
        a.out[i] = v;
    }
}

// ---- scaled RMS norms of an adaptive step (one launch, one small read-back) ----------------------------------------
// out[i] = sqrt(mean_k (((num_i[k] - sub_i[k]) / (atol + rtol * max(|s0_i[k]|, |s1_i[k]|)))^2)),  out[n_terms] = 1 if
// `check` holds a non-finite value else 0.  Deterministic: every block reduces its grid-stride share in a fixed tree,
// writes one partial per term to the workspace, and the block that arrives last adds the partials up in a fixed order
// (double accumulation) -- no floating-point atomics, so equal inputs give equal norms and the accept / reject
// decisions of two runs agree.  Workspace: 1 counter word (left at zero) + kNormBlocks x (FF_NORM_TERMS + 1) doubles.
constexpr int kNormBlocks = 2048;      // 8 workgroups per CU: enough 16-byte loads in flight to stream at the HBM rate

struct NormArgs {
    const float* num[FF_NORM_TERMS];
    const float* sub[FF_NORM_TERMS];
    const float* s0[FF_NORM_TERMS];
    const float* s1[FF_NORM_TERMS];
    long long n[FF_NORM_TERMS];
    int vec_ok[FF_NORM_TERMS];   // every array of the term 16-byte aligned: the body runs on float4
    const float* check;
    long long n_check;
    int check_vec_ok;
    int n_terms;
    float atol, rtol;
    float* out;
    unsigned* counter;
    double* partial;       // [kNormBlocks][FF_NORM_TERMS + 1]
};

__device__ __forceinline__ double block_sum(double v, double* sh)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) {
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    }
    return t;              // valid on thread 0
}

__device__ __forceinline__ double scaled_sq(float num, float sub, float s0, float s1, float atol, float rtol)
{
    const float q = (num - sub) / (atol + rtol * fmaxf(fabsf(s0), fabsf(s1)));
    return (double)q * (double)q;
}

__global__ __launch_bounds__(256) void scaled_rms_kernel(const NormArgs a)
{
    __shared__ double sh[4];
    __shared__ bool last;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const f32x4a zero4 = f32x4a{0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < a.n_terms; ++t) {
        double acc = 0.0;
        const float* num = a.num[t]; const float* sub = a.sub[t]; const float* s0 = a.s0[t]; const float* s1 = a.s1[t];
        const long long n4 = a.vec_ok[t] ? a.n[t] / 4 : 0;
        for (long long i = tid; i < n4; i += stride) {
            const f32x4a vn = ((const f32x4a*)num)[i];
            const f32x4a vs = sub ? ((const f32x4a*)sub)[i] : zero4;
            const f32x4a v0 = ((const f32x4a*)s0)[i];
            const f32x4a v1 = s1 ? ((const f32x4a*)s1)[i] : zero4;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += scaled_sq(vn[j], vs[j], v0[j], v1[j], a.atol, a.rtol);
        }
        for (long long i = 4 * n4 + tid; i < a.n[t]; i += stride)      // unaligned input or the last n % 4 elements
            acc += scaled_sq(num[i], sub ? sub[i] : 0.f, s0[i], s1 ? s1[i] : 0.f, a.atol, a.rtol);
        const double tot = block_sum(acc, sh);
        if (threadIdx.x == 0) a.partial[(size_t)blockIdx.x * (FF_NORM_TERMS + 1) + t] = tot;
    }
    {
        double bad = 0.0;
        const long long n4 = a.check_vec_ok ? a.n_check / 4 : 0;
        for (long long i = tid; i < n4; i += stride) {
            const f32x4a v = ((const f32x4a*)a.check)[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) bad += (v[j] - v[j] == 0.f) ? 0.0 : 1.0;        // NaN or infinity
        }
        for (long long i = 4 * n4 + tid; i < a.n_check; i += stride) {
            const float v = a.check[i];
            bad += (v - v == 0.f) ? 0.0 : 1.0;
        }
        const double tot = block_sum(bad, sh);
        if (threadIdx.x == 0) a.partial[(size_t)blockIdx.x * (FF_NORM_TERMS + 1) + FF_NORM_TERMS] = tot;
    }
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(a.counter, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last) {
        // the block that arrives last adds the partials up: thread i takes blocks i, i + 256, ... in that order, then the
        // same fixed tree as above -- the grid size depends on the array sizes only, so equal inputs give equal sums
        __threadfence();
        for (int t = 0; t <= a.n_terms; ++t) {
            const int col = t < a.n_terms ? t : FF_NORM_TERMS;
            double acc = 0.0;
            for (unsigned b = threadIdx.x; b < gridDim.x; b += blockDim.x)
                acc += __builtin_nontemporal_load(&a.partial[(size_t)b * (FF_NORM_TERMS + 1) + col]);
            const double tot = block_sum(acc, sh);
            if (threadIdx.x == 0) {
                if (t < a.n_terms) a.out[t] = a.n[t] > 0 ? (float)sqrt(tot / (double)a.n[t]) : 0.f;
                else a.out[a.n_terms] = tot > 0.0 ? 1.f : 0.f;
            }
        }
        if (threadIdx.x == 0) *a.counter = 0u;          // ready for the next launch on this stream
    }
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
    long long most = n_check;
    for (int t = 0; t < FF_NORM_TERMS; ++t) {
        const bool on = t < n_terms;
        if (on && (terms[t].n < 0 || (terms[t].n > 0 && (!terms[t].num || !terms[t].scale0)))) return FF_ERR_BADARG;
        k.num[t] = on ? terms[t].num : nullptr; k.sub[t] = on ? terms[t].sub : nullptr;
        k.s0[t] = on ? terms[t].scale0 : nullptr; k.s1[t] = on ? terms[t].scale1 : nullptr;
        k.n[t] = on ? terms[t].n : 0;
        k.vec_ok[t] = ((((uintptr_t)k.num[t]) | ((uintptr_t)k.sub[t]) | ((uintptr_t)k.s0[t]) | ((uintptr_t)k.s1[t])) & 15) == 0;
        if (k.n[t] > most) most = k.n[t];
    }
    k.check_vec_ok = (((uintptr_t)check) & 15) == 0;
    k.check = check; k.n_check = n_check; k.n_terms = n_terms; k.atol = atol; k.rtol = rtol; k.out = out;
    k.counter = (unsigned*)workspace;
    k.partial = (double*)((char*)workspace + 16);
    long long want = (most / 4 + 255) / 256;              // one 16-byte access per thread and trip
    const unsigned grid = (unsigned)(want < 1 ? 1 : (want > ff::kNormBlocks ? ff::kNormBlocks : want));
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
