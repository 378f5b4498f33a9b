// Scratch (planning for a later round): what would a split-precision layer chain
// run at?  One 256x256 layer = 8 row tiles x 16 k-steps of v_mfma_f32_32x32x16_bf16 on 32 samples per
// wavefront; fp32 operands are cut into three bf16 parts (hi, mid, lo) and the six products that matter
// (hi.hi, hi.mid, hi.lo, mid.hi, mid.mid, lo.hi) are accumulated in fp32.  Activations stay in registers
// (accumulator tile -> SiLU -> split -> B fragments of the next layer); weight fragments are read from LDS
// (filled once: timing only, every row tile reuses the same 48 KB -- the L2 -> LDS refill of a real kernel,
// 16 B/clk/CU, is not modelled).  Results are meaningless; the time per layer is the point.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__device__ __forceinline__ f32x16 mm(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// top halves of (a, b) -> one register of two bf16 (a low, b high): truncation split
__device__ __forceinline__ unsigned pack_hi(float a, float b)
{
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
__device__ __forceinline__ float top(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xFFFF0000u); }

// SiLU of 16 accumulator registers, then the three-way split, packed as the B fragments of two k-steps
template <bool ACT>
__device__ __forceinline__ void finish_tile(const f32x16& acc, u32x4 (&o0)[3], u32x4 (&o1)[3])
{
    float h[16], m[16], l[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float v = acc[i];
        if constexpr (ACT) {
            const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504f));
            v = v * r;
        }
        h[i] = v;
        const float r1 = v - top(v);
        m[i] = r1;
        l[i] = r1 - top(r1);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o0[0][j] = pack_hi(h[2 * j], h[2 * j + 1]);
        o0[1][j] = pack_hi(m[2 * j], m[2 * j + 1]);
        o0[2][j] = pack_hi(l[2 * j], l[2 * j + 1]);
        o1[0][j] = pack_hi(h[8 + 2 * j], h[8 + 2 * j + 1]);
        o1[1][j] = pack_hi(m[8 + 2 * j], m[8 + 2 * j + 1]);
        o1[2][j] = pack_hi(l[8 + 2 * j], l[8 + 2 * j + 1]);
    }
}

// SRC: LDS-resident fragments (lds) or, with GLOBAL, this layer's own 384 KB streamed by every wavefront from L2
template <int NPROD, bool ACT, bool GLOBAL = false>
__device__ __forceinline__ void layer(const u32x4* lds, int lane, const u32x4 (&cur)[16][3], u32x4 (&nxt)[16][3])
{
#pragma unroll
    for (int tile = 0; tile < 8; ++tile) {
        f32x16 a0, a1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const u32x4 wh = lds[(GLOBAL ? (tile * 48 + ks * 3 + 0) : ((ks * 3 + 0 + tile * 5) % 48)) * 64 + lane];     // a different fragment per tile
            a0 = mm(wh, cur[ks][0], a0);
            if constexpr (NPROD >= 3) {
                const u32x4 wm = lds[(GLOBAL ? (tile * 48 + ks * 3 + 1) : ((ks * 3 + 1 + tile * 5) % 48)) * 64 + lane];
                a1 = mm(wh, cur[ks][1], a1);
                a0 = mm(wm, cur[ks][0], a0);
                if constexpr (NPROD >= 6) {
                    const u32x4 wl = lds[(GLOBAL ? (tile * 48 + ks * 3 + 2) : ((ks * 3 + 2 + tile * 5) % 48)) * 64 + lane];
                    a1 = mm(wh, cur[ks][2], a1);
                    a0 = mm(wm, cur[ks][1], a0);
                    a1 = mm(wl, cur[ks][0], a1);
                }
            }
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = a0[r] + a1[r];
        finish_tile<ACT>(acc, nxt[2 * tile], nxt[2 * tile + 1]);
    }
}

template <int NPROD, bool ACT>
__global__ __launch_bounds__(256, 1) void kg(const unsigned* __restrict__ wsrc, const float* __restrict__ xin,
                                               float* __restrict__ xout, int pairs)
{
    const int lane = threadIdx.x & 63;
    const u32x4* w = (const u32x4*)wsrc;
    u32x4 A[16][3], B[16][3];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) A[ks][p][j] = pack_hi(xin[lane + 64 * ((ks + p + j) & 7)], xin[lane + 64 * ((ks * 3 + j) & 7) + 512]);
    for (int it = 0; it < pairs; ++it) {
        layer<NPROD, ACT, true>(w, lane, A, B);
        layer<NPROD, ACT, true>(w + 8 * 48 * 64, lane, B, A);
    }
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p) s += __builtin_bit_cast(float, A[ks][p][0] << 16);
    xout[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// Hand-ordered variant of the LDS-resident chain: fragments of k-step ks+1 are read while the six MFMAs of
// k-step ks issue; the activation + split of the PREVIOUS tile's accumulator is spread over the first eight
// k-steps of the current tile (two registers per step), so the VALU work sits in the MFMAs' shadow; a masked
// sched_barrier per k-step pins the MFMA / LDS order and leaves VALU placement to the scheduler.
template <bool ACT>
__device__ __forceinline__ void split_pair(float v0, float v1, u32x4 (&o)[3], int j)
{
    if constexpr (ACT) {
        v0 = v0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v0 * -1.44269504f));
        v1 = v1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v1 * -1.44269504f));
    }
    const float m0 = v0 - top(v0), m1 = v1 - top(v1);
    const float l0 = m0 - top(m0), l1 = m1 - top(m1);
    o[0][j] = pack_hi(v0, v1);
    o[1][j] = pack_hi(m0, m1);
    o[2][j] = pack_hi(l0, l1);
}

template <bool ACT>
__global__ __launch_bounds__(256, 1) void kh(const unsigned* __restrict__ wsrc, const float* __restrict__ xin,
                                               float* __restrict__ xout, int pairs)
{
    extern __shared__ u32x4 lds[];
    for (int i = threadIdx.x; i < 16 * 3 * 64; i += 256) lds[i] = ((const u32x4*)wsrc)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    u32x4 A[16][3], B[16][3];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) A[ks][p][j] = pack_hi(xin[lane + 64 * ((ks + p + j) & 7)], xin[lane + 64 * ((ks * 3 + j) & 7) + 512]);
    f32x16 p0;                                      // the previous tile's accumulator
#pragma unroll
    for (int r = 0; r < 16; ++r) p0[r] = 0.f;
    // one layer; `prev_out` receives the previous tile's fragments: for tile 0 that is the LAST tile of the layer
    // before (k-steps 14, 15 of `cur`, complete before this tile reaches k-step 14), otherwise tile-1 of `nxt`
    auto one_layer = [&](u32x4 (&cur)[16][3], u32x4 (&nxt)[16][3]) {
#pragma unroll
        for (int tile = 0; tile < 8; ++tile) {
            f32x16 a0;                                  // one accumulator: dependent MFMAs issue back to back
#pragma unroll
            for (int r = 0; r < 16; ++r) a0[r] = 0.f;
            u32x4 wh = lds[((0 + tile * 5) % 48) * 64 + lane], wm = lds[((1 + tile * 5) % 48) * 64 + lane],
                  wl = lds[((2 + tile * 5) % 48) * 64 + lane];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                u32x4 nh = wh, nm = wm, nl = wl;
                if (ks < 15) {
                    nh = lds[((ks * 3 + 3 + tile * 5) % 48) * 64 + lane];
                    nm = lds[((ks * 3 + 4 + tile * 5) % 48) * 64 + lane];
                    nl = lds[((ks * 3 + 5 + tile * 5) % 48) * 64 + lane];
                }
                a0 = mm(wh, cur[ks][0], a0);
                a0 = mm(wh, cur[ks][1], a0);
                a0 = mm(wm, cur[ks][0], a0);
                a0 = mm(wh, cur[ks][2], a0);
                a0 = mm(wm, cur[ks][1], a0);
                a0 = mm(wl, cur[ks][0], a0);
                if (ks < 8) {                          // previous tile: registers 2ks, 2ks+1
                    const float v0 = p0[2 * ks], v1 = p0[2 * ks + 1];
                    if (tile == 0) split_pair<ACT>(v0, v1, cur[14 + (ks >> 2)], ks & 3);
                    else split_pair<ACT>(v0, v1, nxt[2 * (tile - 1) + (ks >> 2)], ks & 3);
                }
                wh = nh; wm = nm; wl = nl;
                __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x80 | 0x100 | 0x200);   // VALU / SALU / trans may move; MFMA, DS pinned
            }
            p0 = a0;
        }
    };
    for (int it = 0; it < pairs; ++it) {
        one_layer(A, B);
        one_layer(B, A);
    }
    float s = p0[0];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p) s += __builtin_bit_cast(float, A[ks][p][0] << 16);
    xout[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool ACT>
__global__ __launch_bounds__(256, 1) void khp(const unsigned* __restrict__ wsrc, const float* __restrict__ xin,
                                               float* __restrict__ xout, int pairs)
{
    // as kh, plus the real weight path: THREE LDS buffers of one row tile's 48 fragments; at the start of tile t the
    // four wavefronts start the LDS-DMA (global_load_lds, 12 fragments each, no registers) of tile t+2; at the end
    // of tile t a counted wait (vmcnt(12): everything but the newest tile's DMA) and one raw barrier publish tile t+1
    extern __shared__ u32x4 lds[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // scalar: addresses stay in SGPRs
    const u32x4* w = (const u32x4*)wsrc;
#pragma unroll
    for (int f = 0; f < 12; ++f) {
        lds[(wv * 12 + f) * 64 + lane] = w[(wv * 12 + f) * 64 + lane];                              // tile 0
        lds[3072 + (wv * 12 + f) * 64 + lane] = w[48 * 64 + (wv * 12 + f) * 64 + lane];            // tile 1
    }
    __syncthreads();
    u32x4 A[16][3], B[16][3];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) A[ks][p][j] = pack_hi(xin[lane + 64 * ((ks + p + j) & 7)], xin[lane + 64 * ((ks * 3 + j) & 7) + 512]);
    f32x16 p0;                                      // the previous tile's accumulator
#pragma unroll
    for (int r = 0; r < 16; ++r) p0[r] = 0.f;
    // one layer; `prev_out` receives the previous tile's fragments: for tile 0 that is the LAST tile of the layer
    // before (k-steps 14, 15 of `cur`, complete before this tile reaches k-step 14), otherwise tile-1 of `nxt`
    auto one_layer = [&](auto lidx, u32x4 (&cur)[16][3], u32x4 (&nxt)[16][3]) {
        constexpr int LI = decltype(lidx)::value;
#pragma unroll
        for (int tile = 0; tile < 8; ++tile) {
            f32x16 a0;                                  // one accumulator: dependent MFMAs issue back to back
#pragma unroll
            for (int r = 0; r < 16; ++r) a0[r] = 0.f;
            constexpr int GT = LI * 8;                  // tiles done before this layer (mod 48 = 16 stored tiles x 3 buffers)
            const u32x4* lb = lds + ((GT + tile) % 3) * 3072;
            const u32x4* wn = w + (size_t)((LI * 8 + tile + 2) & 15) * 48 * 64 + (wv * 12) * 64;          // wave-uniform
            u32x4 wh = lb[0 * 64 + lane], wm = lb[1 * 64 + lane], wl = lb[2 * 64 + lane];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                u32x4 nh = wh, nm = wm, nl = wl;
                if (ks < 15) {
                    nh = lb[(ks * 3 + 3) * 64 + lane];
                    nm = lb[(ks * 3 + 4) * 64 + lane];
                    nl = lb[(ks * 3 + 5) * 64 + lane];
                }
                if (ks == 0) {                                   // LDS-DMA of tile t+2 into the buffer tile t-1 used
                    // inline asm (M0 = LDS base of the fragment, lanes land at base + 16*lane): as a builtin the DMA
                    // makes hipcc spill ~2 KB per lane here
#pragma unroll
                    for (int fr = 0; fr < 12; ++fr) {
                        const unsigned ldsb = (((GT + tile + 2) % 3) * 3072 + (wv * 12 + fr) * 64) * 16;
                        const u32x4* g = wn + fr * 64;
                        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsb), "v"(lane * 16), "s"(g) : "m0");
                    }
                }
                a0 = mm(wh, cur[ks][0], a0);
                a0 = mm(wh, cur[ks][1], a0);
                a0 = mm(wm, cur[ks][0], a0);
                a0 = mm(wh, cur[ks][2], a0);
                a0 = mm(wm, cur[ks][1], a0);
                a0 = mm(wl, cur[ks][0], a0);
                if (ks < 8) {                          // previous tile: registers 2ks, 2ks+1
                    const float v0 = p0[2 * ks], v1 = p0[2 * ks + 1];
                    if (tile == 0) split_pair<ACT>(v0, v1, cur[14 + (ks >> 2)], ks & 3);
                    else split_pair<ACT>(v0, v1, nxt[2 * (tile - 1) + (ks >> 2)], ks & 3);
                }
                wh = nh; wm = nm; wl = nl;
                __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x80 | 0x100 | 0x200);   // VALU / SALU / trans may move; MFMA, DS pinned
            }
            p0 = a0;
            __builtin_amdgcn_s_waitcnt(0x007c);            // vmcnt(12) lgkmcnt(0): all but the newest tile's DMA have landed
            __builtin_amdgcn_s_barrier();
        }
    };
    for (int it = 0; it < pairs; ++it) {
        one_layer(std::integral_constant<int, 0>{}, A, B);
        one_layer(std::integral_constant<int, 1>{}, B, A);
    }
    float s = p0[0];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p) s += __builtin_bit_cast(float, A[ks][p][0] << 16);
    xout[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// Weight pipeline as a real kernel would have it: tile-major, the 48 fragments (48 KiB) of the next row tile are
// fetched from L2 by the four wavefronts together (12 KiB each, held in registers while the current tile is
// computed), written to the other LDS buffer, one barrier per tile.
template <bool ACT>
__global__ __launch_bounds__(256, 1) void kp(const unsigned* __restrict__ wsrc, const float* __restrict__ xin,
                                               float* __restrict__ xout, int pairs)
{
    extern __shared__ u32x4 lds[];                      // 2 x 3072 fragments-lanes of 16 B = 96 KiB
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32x4* w = (const u32x4*)wsrc;
    u32x4 A[16][3], B[16][3];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) A[ks][p][j] = pack_hi(xin[lane + 64 * ((ks + p + j) & 7)], xin[lane + 64 * ((ks * 3 + j) & 7) + 512]);
    // tile 0 of layer 0 into buffer 0
#pragma unroll
    for (int f = 0; f < 12; ++f) lds[(wv * 12 + f) * 64 + lane] = w[(wv * 12 + f) * 64 + lane];
    __syncthreads();
    auto one_layer = [&](int layer_idx, const u32x4 (&cur)[16][3], u32x4 (&nxt)[16][3]) {
#pragma unroll
        for (int tile = 0; tile < 8; ++tile) {
            const int buf = tile & 1;                    // 8 tiles per layer: parity carries across layers
            // next tile's fragments (wrapping over the two layers' 16 tiles): global -> registers
            const int nt = (layer_idx * 8 + tile + 1) & 15;
            u32x4 pre[12];
#pragma unroll
            for (int f = 0; f < 12; ++f) pre[f] = w[(nt * 48 + wv * 12 + f) * 64 + lane];
            const u32x4* lb = lds + buf * 3072;
            f32x16 a0, a1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const u32x4 wh = lb[(ks * 3 + 0) * 64 + lane], wm = lb[(ks * 3 + 1) * 64 + lane], wl = lb[(ks * 3 + 2) * 64 + lane];
                a0 = mm(wh, cur[ks][0], a0);
                a1 = mm(wh, cur[ks][1], a1);
                a0 = mm(wm, cur[ks][0], a0);
                a1 = mm(wh, cur[ks][2], a1);
                a0 = mm(wm, cur[ks][1], a0);
                a1 = mm(wl, cur[ks][0], a1);
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = a0[r] + a1[r];
            finish_tile<ACT>(acc, nxt[2 * tile], nxt[2 * tile + 1]);
            u32x4* ob = lds + (buf ^ 1) * 3072;
#pragma unroll
            for (int f = 0; f < 12; ++f) ob[(wv * 12 + f) * 64 + lane] = pre[f];
            __syncthreads();
        }
    };
    for (int it = 0; it < pairs; ++it) {
        one_layer(0, A, B);
        one_layer(1, B, A);
    }
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p) s += __builtin_bit_cast(float, A[ks][p][0] << 16);
    xout[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NPROD, bool ACT>
__global__ __launch_bounds__(256, 1) void k(const unsigned* __restrict__ wsrc, const float* __restrict__ xin,
                                              float* __restrict__ xout, int pairs)
{
    extern __shared__ u32x4 lds[];
    for (int i = threadIdx.x; i < 16 * 3 * 64; i += 256) lds[i] = ((const u32x4*)wsrc)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    u32x4 A[16][3], B[16][3];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) A[ks][p][j] = pack_hi(xin[lane + 64 * ((ks + p + j) & 7)], xin[lane + 64 * ((ks * 3 + j) & 7) + 512]);
    for (int it = 0; it < pairs; ++it) {       // two layers per iteration: A -> B -> A
        layer<NPROD, ACT>(lds, lane, A, B);
        layer<NPROD, ACT>(lds, lane, B, A);
    }
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p) s += __builtin_bit_cast(float, A[ks][p][0] << 16);
    xout[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NPROD, bool ACT, bool GLOBAL = false>
void run(const unsigned* dw, const float* dx, float* dy, int pairs, const char* what)
{
    const int nwg = 256 * 4;               // one workgroup per CU at a time (48 KB LDS, 4 waves), 4 rounds
    auto kern = GLOBAL ? kg<NPROD, ACT> : k<NPROD, ACT>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 49152, 0, dw, dx, dy, 2);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 49152, 0, dw, dx, dy, pairs);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double layers = 2.0 * pairs;
    const double sample_layers = (double)nwg * 4 * 32 * layers;            // (sample, layer) pairs processed
    const double flop32 = sample_layers * 2.0 * 256 * 256;                  // what the fp32 kernel counts
    printf("%-34s %8.2f ms  %7.1f fp32-equivalent TFLOP/s  (%.2fx of 142)   %.0f MFMA/layer/wave\n", what, ms,
           flop32 / ms / 1e9, flop32 / ms / 1e9 / 142.0, 128.0 * NPROD);
}

int main(int argc, char** argv)
{
    const int pairs = argc > 1 ? atoi(argv[1]) : 100;
    std::vector<unsigned> hw((size_t)2 * 8 * 48 * 64 * 4);       // two layers x 8 tiles x 48 fragments of 1 KiB
    srand(2);
    for (auto& v : hw) {                    // two small bf16 per word
        const unsigned short a = 0x3C00 + (rand() & 0xFF), b = 0xBC00 + (rand() & 0xFF);
        v = a | ((unsigned)b << 16);
    }
    std::vector<float> hx(2048);
    for (auto& v : hx) v = (rand() / (float)RAND_MAX - 0.5f) * 0.5f;
    unsigned* dw; float *dx, *dy;
    CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dy, (size_t)1024 * 256 * 4));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    run<6, true>(dw, dx, dy, pairs, "6 products, SiLU + split");
    run<6, false>(dw, dx, dy, pairs, "6 products, split only");
    run<3, true>(dw, dx, dy, pairs, "3 products, SiLU + split");
    run<1, true>(dw, dx, dy, pairs, "1 product (plain bf16), SiLU");
    {
        const int nwg = 256 * 4;
        auto kern = kh<true>;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 49152, 0, dw, dx, dy, 2);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 49152, 0, dw, dx, dy, pairs);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double flop32 = (double)nwg * 4 * 32 * 2.0 * pairs * 2.0 * 256 * 256;
        printf("%-34s %8.2f ms  %7.1f fp32-equivalent TFLOP/s  (%.2fx of 142)\n", "6 products, hand-ordered (LDS)", ms,
               flop32 / ms / 1e9, flop32 / ms / 1e9 / 142.0);
    }
    {
        const int nwg = 256 * 4;
        auto kern = khp<true>;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 147456, 0, dw, dx, dy, 2);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 147456, 0, dw, dx, dy, pairs);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double flop32 = (double)nwg * 4 * 32 * 2.0 * pairs * 2.0 * 256 * 256;
        printf("%-34s %8.2f ms  %7.1f fp32-equivalent TFLOP/s  (%.2fx of 142)\n", "6 products, hand-ordered + L2 pipe", ms,
               flop32 / ms / 1e9, flop32 / ms / 1e9 / 142.0);
    }
    {
        const int nwg = 256 * 4;
        auto kern = kp<true>;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 98304, 0, dw, dx, dy, 2);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 98304, 0, dw, dx, dy, pairs);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double flop32 = (double)nwg * 4 * 32 * 2.0 * pairs * 2.0 * 256 * 256;
        printf("%-34s %8.2f ms  %7.1f fp32-equivalent TFLOP/s  (%.2fx of 142)\n", "6 products, L2->LDS pipeline", ms,
               flop32 / ms / 1e9, flop32 / ms / 1e9 / 142.0);
    }
    run<6, true, true>(dw, dx, dy, pairs, "6 products, weights from L2");
    run<3, true, true>(dw, dx, dy, pairs, "3 products, weights from L2");
    return 0;
}
