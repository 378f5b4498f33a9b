// ff_norm.h -- the deterministic scaled-RMS reduction shared by ff_scaled_rms (ff_aux.hip) and the device-side adaptive
// controller (ff_adaptive.hip): every block reduces its grid-stride share in a fixed tree and writes one partial per term;
// the block that arrives last adds the partials up in a fixed order (double accumulation, no floating-point atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flowfusion_amd.h"

namespace ff {

typedef float f32x4a __attribute__((ext_vector_type(4)));

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
    double* sums;          // optional [2 * FF_NORM_TERMS + 2]: the raw sums of squares, the non-finite count and the element
                           // counts instead of the norms (they meet the other ranks' before the norms are taken)
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

// Returns true in every thread of the block that arrived last; that block holds the results in res[0 .. n_terms)
// (the norms) and res[FF_NORM_TERMS] (1 = `check` holds a non-finite value) after the call.  `a.out` (optional) receives
// them as ff_scaled_rms documents.  The counter word is left at zero for the next launch on the stream.
__device__ __forceinline__ bool scaled_rms_reduce(const NormArgs& a, double* sh, bool* last_flag, float* res)
{
    bool& last = *last_flag;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const f32x4a zero4 = f32x4a{0.f, 0.f, 0.f, 0.f};
    // The finiteness check of an attempted step looks at y1, which is also the second scale array of the first term
    // (err / (atol + rtol max(|y0|, |y1|))): when `check` IS that array the check rides in the term's pass instead of reading
    // the array a second time (12 instead of 16 bytes per element, one pass instead of two).
    const bool fold_check = a.n_terms > 0 && a.check != nullptr && a.check == a.s1[0] && a.n_check == a.n[0];
    double bad = 0.0;
    // a grand total becomes a result (thread 0; t == n_terms: the non-finite count)
    auto publish = [&](int t, double tot) __attribute__((always_inline)) {
        const int col = t < a.n_terms ? t : FF_NORM_TERMS;
        if (a.sums) {
            a.sums[col] = tot;
            if (t < a.n_terms) a.sums[FF_NORM_TERMS + 1 + t] = (double)a.n[t];
        } else if (t < a.n_terms) {
            res[t] = a.n[t] > 0 ? (float)sqrt(tot / (double)a.n[t]) : 0.f;
            if (a.out) a.out[t] = res[t];
        } else {
            res[FF_NORM_TERMS] = tot > 0.0 ? 1.f : 0.f;
            if (a.out) a.out[a.n_terms] = res[FF_NORM_TERMS];
        }
    };
    // One block (small states: the notebook-scale solves): its totals ARE the grand totals -- no partials, no arrival
    // counter, four dependent trips to memory less per launch.  The same numbers: the partial path would add this block's
    // total to zeros.
    const bool solo = gridDim.x == 1;
    for (int t = 0; t < a.n_terms; ++t) {
        double acc = 0.0;
        const float* num = a.num[t]; const float* sub = a.sub[t]; const float* s0 = a.s0[t]; const float* s1 = a.s1[t];
        const long long n4 = a.vec_ok[t] ? a.n[t] / 4 : 0;
        const bool chk = fold_check && t == 0;
        // two 16-byte elements per thread and trip (up to eight loads in flight).
        // The order in which a thread adds its elements is fixed by (grid, n) alone, so equal inputs still give equal sums.
        auto term4 = [&](long long i) __attribute__((always_inline)) {
            const f32x4a vn = ((const f32x4a*)num)[i];
            const f32x4a vs = sub ? ((const f32x4a*)sub)[i] : zero4;
            const f32x4a v0 = ((const f32x4a*)s0)[i];
            const f32x4a v1 = s1 ? ((const f32x4a*)s1)[i] : zero4;
            double r = 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r += scaled_sq(vn[j], vs[j], v0[j], v1[j], a.atol, a.rtol);
                if (chk) bad += (v1[j] - v1[j] == 0.f) ? 0.0 : 1.0;        // NaN or infinity
            }
            return r;
        };
        long long i = tid;
        for (; i + stride < n4; i += 2 * stride) {
            const double r0 = term4(i), r1 = term4(i + stride);
            acc += r0;
            acc += r1;
        }
        if (i < n4) acc += term4(i);
        for (long long i = 4 * n4 + tid; i < a.n[t]; i += stride) {    // unaligned input or the last n % 4 elements
            acc += scaled_sq(num[i], sub ? sub[i] : 0.f, s0[i], s1 ? s1[i] : 0.f, a.atol, a.rtol);
            if (chk) bad += (s1[i] - s1[i] == 0.f) ? 0.0 : 1.0;
        }
        const double tot = block_sum(acc, sh);
        if (threadIdx.x == 0) {
            if (solo) publish(t, tot);
            else a.partial[(size_t)blockIdx.x * (FF_NORM_TERMS + 1) + t] = tot;
        }
    }
    {
        if (!fold_check) {
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
        }
        const double tot = block_sum(bad, sh);
        if (threadIdx.x == 0) {
            if (solo) publish(a.n_terms, tot);
            else a.partial[(size_t)blockIdx.x * (FF_NORM_TERMS + 1) + FF_NORM_TERMS] = tot;
        }
    }
    if (solo) {
        if (threadIdx.x == 0) {
            last = true;
            if (a.sums)
                for (int t = a.n_terms; t < FF_NORM_TERMS; ++t) { a.sums[t] = 0.0; a.sums[FF_NORM_TERMS + 1 + t] = 0.0; }
        }
        __syncthreads();
        return true;
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
            if (threadIdx.x == 0) publish(t, tot);
        }
        if (threadIdx.x == 0 && a.sums)
            for (int t = a.n_terms; t < FF_NORM_TERMS; ++t) { a.sums[t] = 0.0; a.sums[FF_NORM_TERMS + 1 + t] = 0.0; }
        if (threadIdx.x == 0) *a.counter = 0u;          // ready for the next launch on this stream
        __syncthreads();
    }
    return last;
}

// fill a NormArgs from the public term list; returns the grid size, 0 on a bad argument
inline unsigned norm_args_from_terms(NormArgs& k, const ff_norm_term* terms, int n_terms, float atol, float rtol,
                                     const float* check, long long n_check, float* out, void* workspace)
{
    long long most = n_check;
    for (int t = 0; t < FF_NORM_TERMS; ++t) {
        const bool on = t < n_terms;
        if (on && (terms[t].n < 0 || (terms[t].n > 0 && (!terms[t].num || !terms[t].scale0)))) return 0;
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
    k.sums = nullptr;
    if (most / 4 <= 2048) return 1;                       // up to eight trips of one block: the single-block path (no arrival counter)
    long long want = (most / 4 + 255) / 256;              // one 16-byte access per thread and trip
    return (unsigned)(want < 1 ? 1 : (want > kNormBlocks ? kNormBlocks : want));
}


} // namespace ff
