// ff_trace.hip -- Hutch++ / XTrace divergence estimates from recorded Jacobians, one launch for every (evaluation row,
// sample) of a fused launch (gfx950).
//
// The reference evaluates the two estimators inside ScoreModel.forward (flowfusion/diffusion.py:336-481), i.e. once per
// right-hand-side evaluation of the solver, through 2r + m reverse-mode products and a batched `torch.linalg.qr`.  On the
// fused path a launch in FF_MODE_EXACT with ff_ode_args.jac_all = 1 leaves A = J^T of EVERY evaluation row in memory
// ([n_rows][batch][D][D]); this file turns them into the estimates [n_rows][batch] in one pass, so that an attempted step of
// the adaptive solver stays on the device (ff_adaptive.hip) and a fixed-grid solve needs no torch linear algebra.
//
// Roofline: HBM -- algorithmically one read of the Jacobians (4 D^2 bytes per work item) and 4 bytes written.
//
//   trace_estimate_tile_kernel<DC, ITEMS>   the fast path (D <= 32, and the tile fits LDS): one wavefront = 64 consecutive work
//       items (32 beyond 16 dimensions).  Their matrices are one contiguous stretch of memory: the wavefront streams it with coalesced loads (lane l
//       takes element l of every 64) and scatters it into LDS TRANSPOSED to item-fastest with an odd pitch (65 words), so
//       that both the scatter (consecutive elements of one matrix -> consecutive banks) and every later read (lane = item ->
//       consecutive banks) are conflict-free.  The factorisation's scratch lives in LDS the same way.  Then each lane runs its
//       item's estimator (ff_trace_est.h) out of LDS with the dimension fixed at compile time: the operand vector of every
//       product sits in registers, the dot products unroll and their LDS reads are issued in batches instead of one
//       dependent load at a time.
//   trace_estimate_kernel            the general path (any D, r, m): one thread per item straight from global memory, scratch
//       in a caller-provided workspace laid out item-fastest.  Latency-bound (dependent uncoalesced loads); correct for
//       shapes the tile path does not hold.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include "flowfusion_amd.h"
#include "ff_registry.h"
#include "ff_trace_est.h"

namespace ff {

// words between consecutive LDS rows of a tile of ITEMS work items (odd: conflict-free both ways)
__host__ __device__ constexpr int tile_pitch(int items) { return items + 1; }

__global__ __launch_bounds__(256) void trace_estimate_kernel(const ff_trace_args a, long long items)
{
    if (a.gate && *(const volatile int32_t*)a.gate == 0) return;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= items) return;
    const long long b = t % a.batch;
    trace::Item it;
    it.A = a.jac + (size_t)t * a.dim * a.dim;
    it.astride = 1;
    it.p0 = a.probes0 + (size_t)b * a.dim;
    it.pstride0 = (size_t)a.batch * a.dim; it.pk0 = 1;
    it.p1 = a.probes1 ? a.probes1 + (size_t)b * a.dim : nullptr;
    it.pstride1 = (size_t)a.batch * a.dim; it.pk1 = 1;
    it.ws = a.workspace + t;
    it.stride = (size_t)items;
    it.D = a.dim; it.r = a.r; it.m = a.m;
    a.out[t] = trace::estimate<0>(a.kind, it);
}

// LDS words of a tile of `items` work items: matrices and scratch (the probes are read from global memory: with the operand
// vectors of the products held in registers every probe entry is read once or twice)
__host__ __device__ inline size_t tile_lds_words(int kind, int D, int r, int items)
{
    return (size_t)tile_pitch(items) * ((size_t)D * D + trace::workspace_per_item(kind, D, r));
}

typedef float f32x4g __attribute__((ext_vector_type(4)));

// ITEMS = work items per tile: 64 (one per lane) up to 16 dimensions; 32 for 17-32 dimensions, whose matrices are four times
// the size (half the lanes only help with the staging: still far ahead of one dependent global load at a time)
template <int DC, int ITEMS>
__global__ __launch_bounds__(64) void trace_estimate_tile_kernel(const ff_trace_args a, long long items)
{
    if (a.gate && *(const volatile int32_t*)a.gate == 0) return;
    extern __shared__ float tile[];
    constexpr int kTilePitch = tile_pitch(ITEMS);
    const int D = DC ? DC : a.dim, DD = D * D;
    const int lane = threadIdx.x;
    const long long t0 = (long long)blockIdx.x * ITEMS;
    const int n_here = (int)(items - t0 < ITEMS ? items - t0 : ITEMS);
    float* const a_lds = tile;                                   // [D * D][pitch]
    float* const w_lds = a_lds + (size_t)DD * kTilePitch;        // [workspace_per_item][pitch]

    // the tile's matrices: n_here * D * D contiguous floats; element e = item * DD + idx -> a_lds[idx * pitch + item].
    // Sixteen 16-byte loads per lane in flight (16 KiB per wavefront) when the stretch is 16-byte aligned, else eight dwords.
    const float* src = a.jac + (size_t)t0 * DD;
    const int n_el = n_here * DD;
    if ((DD & 3) == 0 && (((uintptr_t)src) & 15) == 0) {
        const int n4 = n_el >> 2;
        for (int e0 = 0; e0 < n4; e0 += 64 * 16) {
            f32x4g v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = e0 + u * 64 + lane;
                v[u] = e < n4 ? ((const f32x4g*)src)[e] : f32x4g{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = (e0 + u * 64 + lane) * 4;
                if (e < n_el) {
                    const int item = e / DD, idx = e - item * DD;          // (DD % 4 == 0: the four words share an item)
#pragma unroll
                    for (int q = 0; q < 4; ++q) a_lds[(idx + q) * kTilePitch + item] = v[u][q];
                }
            }
        }
    } else {
        for (int e0 = 0; e0 < n_el; e0 += 64 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * 64 + lane;
                v[u] = e < n_el ? src[e] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * 64 + lane;
                if (e < n_el) {
                    const int item = e / DD, idx = e - item * DD;
                    a_lds[idx * kTilePitch + item] = v[u];
                }
            }
        }
    }
    __syncthreads();
    if (lane >= n_here) return;
    const long long t = t0 + lane, b = t % a.batch;
    trace::Item it;
    it.A = a_lds + lane; it.astride = kTilePitch;
    it.p0 = a.probes0 + (size_t)b * D;
    it.pstride0 = (size_t)a.batch * D; it.pk0 = 1;
    it.p1 = a.probes1 ? a.probes1 + (size_t)b * D : nullptr;
    it.pstride1 = (size_t)a.batch * D; it.pk1 = 1;
    it.ws = w_lds + lane; it.stride = kTilePitch;
    it.D = D; it.r = a.r; it.m = a.m;
    a.out[t] = trace::estimate<DC>(a.kind, it);
}

template <int DC, int ITEMS = 64>
static hipError_t launch_tile(const ff_trace_args* a, long long items, size_t lds_bytes, hipStream_t s)
{
    auto kern = trace_estimate_tile_kernel<DC, ITEMS>;
    // the dynamic-LDS limit is a per-device attribute of the function: set it once per device of this process
    static std::atomic<unsigned char> ready[kMaxDevices];
    int dev = 0;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess) return err;
    if (dev < 0 || dev >= kMaxDevices || !ready[dev].load(std::memory_order_acquire)) {
        err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (err != hipSuccess) return err;
        if (dev >= 0 && dev < kMaxDevices) ready[dev].store(1, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((items + ITEMS - 1) / ITEMS)), dim3(64), lds_bytes, s, *a, items);
    return hipGetLastError();
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
    hipStream_t s = (hipStream_t)hip_stream;
    // the tile path: the matrices and the scratch of a tile of work items within a CU's LDS -- 64 items up to 16 dimensions
    // (up to 80 KB two tiles share a CU and one loads while the other computes), 32 items for 17-32 dimensions.
    // FF_TRACE_GENERIC=1 pins the general kernel (A/B runs, tests).
    const char* pin = getenv("FF_TRACE_GENERIC");
    const int tile_items = a->dim <= 16 ? 64 : 32;
    const size_t lds_bytes = 4 * ff::tile_lds_words(a->kind, a->dim, a->r, tile_items);
    if (a->dim <= 32 && lds_bytes <= 158 * 1024 && !(pin && atoi(pin) != 0) && (items + tile_items - 1) / tile_items <= 0x7fffffffll) {
        hipError_t err;
        switch (a->dim) {
        case 1: err = ff::launch_tile<1>(a, items, lds_bytes, s); break;
        case 2: err = ff::launch_tile<2>(a, items, lds_bytes, s); break;
        case 3: err = ff::launch_tile<3>(a, items, lds_bytes, s); break;
        case 4: err = ff::launch_tile<4>(a, items, lds_bytes, s); break;
        case 8: err = ff::launch_tile<8>(a, items, lds_bytes, s); break;
        case 16: err = ff::launch_tile<16>(a, items, lds_bytes, s); break;
        case 32: err = ff::launch_tile<32, 32>(a, items, lds_bytes, s); break;
        default:                                                                  // run-time dimension, same layout
            err = a->dim <= 16 ? ff::launch_tile<0>(a, items, lds_bytes, s) : ff::launch_tile<0, 32>(a, items, lds_bytes, s);
            break;
        }
        return err == hipSuccess ? FF_OK : FF_ERR_HIP;
    }
    const long long grid = (items + 255) / 256;
    if (grid > 0x7fffffffll) return FF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ff::trace_estimate_kernel, dim3((unsigned)grid), dim3(256), 0, s, *a, items);
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
        it.astride = 1;
        it.p0 = a->probes0 + (size_t)b * a->dim;
        it.pstride0 = (size_t)a->batch * a->dim; it.pk0 = 1;
        it.p1 = a->probes1 ? a->probes1 + (size_t)b * a->dim : nullptr;
        it.pstride1 = (size_t)a->batch * a->dim; it.pk1 = 1;
        it.ws = a->workspace;
        it.stride = 1;
        it.D = a->dim; it.r = a->r; it.m = a->m;
        a->out[t] = ff::trace::estimate<0>(a->kind, it);
    }
    return FF_OK;
}
