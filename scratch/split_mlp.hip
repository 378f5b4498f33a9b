// Scratch (groundwork for DESIGN.md section 8 item 1): END-TO-END CORRECTNESS of the split-precision layer chain.
// One forward pass of a 16 -> 256 x4 -> 16 SiLU MLP on v_mfma_f32_32x32x16_bf16 with every fp32 operand cut into
// three bf16 parts and six products per k-step, activations chained in registers (accumulator tile -> SiLU ->
// split -> B fragments of the next layer), weights packed on the host into the permuted fragment order.  Staging
// of the fragments is synchronous here (the pipelined timing is bench_proto5.hip); the point is the packing,
// the k permutation and the numerics: the result is checked against a float64 MLP on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

constexpr int D = 16, H = 256, NH = 4;          // state dim, width, hidden layers
constexpr int NT = H / 32;                      // row tiles per hidden layer
// k index (input feature of the layer) held in element j of lane half h of the fragment of k-step s: the order in
// which an accumulator tile's registers become the next layer's B fragments (cdna_hip_programming.md,
// "An accumulator tile as the next MFMA's operand")
static inline __host__ __device__ int kidx(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

__device__ __forceinline__ f32x16 mm(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pack_hi(float a, float b)
{
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
__device__ __forceinline__ float top(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xFFFF0000u); }
// registers 8u .. 8u+7 of a tile (already activated) -> the three bf16 fragments of one k-step
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&o)[3])
{
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float a = v[2 * m], b = v[2 * m + 1];
        const float am = a - top(a), bm = b - top(b);
        const float al = am - top(am), bl = bm - top(bm);
        o[0][m] = pack_hi(a, b);
        o[1][m] = pack_hi(am, bm);
        o[2][m] = pack_hi(al, bl);
    }
}
__device__ __forceinline__ float silu(float a) { return a * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a * -1.44269504f)); }

// wfrag: [slot][frag 48][lane 64] x 16 B.  slot 0 = layer 1 (8 tiles x 1 k-step x 3 parts, padded to 48), slots
// 1 + 8l + t = hidden layer l tile t (16 k-steps x 3 parts), last slot = output layer tile 0.
// bias: [layer][256] fp32 (layer 1, hidden.., output).
__global__ __launch_bounds__(256, 1) void split_mlp_forward(const u32x4* __restrict__ wfrag, const float* __restrict__ bias,
                                                            const float* __restrict__ x, float* __restrict__ out, int B)
{
    extern __shared__ u32x4 lds[];               // one slot: 48 fragments
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int sample = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + r;
    const int srow = sample < B ? sample : B - 1;
    auto stage = [&](int slot) {                 // synchronous: everybody waits, everybody copies, everybody waits
        __syncthreads();
        for (int i = threadIdx.x; i < 48 * 64; i += 256) lds[i] = wfrag[(size_t)slot * 48 * 64 + i];
        __syncthreads();
    };
    auto bias_tile = [&](int layer, int t) {     // accumulator tile initialised with the bias of its rows
        f32x16 a;
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = bias[layer * H + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * h];
        return a;
    };
    u32x4 cur[16][3], nxt[16][3];
    // layer 1: the state's registers 0..7 (feature (j&3) + 8(j>>2) + 4h = kidx(0, h, j)) are one k-step
    {
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = x[(size_t)srow * D + kidx(0, h, j)];
        u32x4 yf[3];
        split8(y, yf);
        stage(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x16 a = bias_tile(0, t);
            const u32x4 wh = lds[(t * 3 + 0) * 64 + lane], wm = lds[(t * 3 + 1) * 64 + lane], wl = lds[(t * 3 + 2) * 64 + lane];
            a = mm(wh, yf[0], a); a = mm(wh, yf[1], a); a = mm(wm, yf[0], a);
            a = mm(wh, yf[2], a); a = mm(wm, yf[1], a); a = mm(wl, yf[0], a);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = silu(a[8 * u + j]);
                split8(v, cur[2 * t + u]);
            }
        }
    }
    // hidden -> hidden
    for (int l = 0; l < NH - 1; ++l) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            stage(1 + l * NT + t);
            f32x16 a = bias_tile(1 + l, t);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const u32x4 wh = lds[(s * 3 + 0) * 64 + lane], wm = lds[(s * 3 + 1) * 64 + lane], wl = lds[(s * 3 + 2) * 64 + lane];
                a = mm(wh, cur[s][0], a); a = mm(wh, cur[s][1], a); a = mm(wm, cur[s][0], a);
                a = mm(wh, cur[s][2], a); a = mm(wm, cur[s][1], a); a = mm(wl, cur[s][0], a);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = silu(a[8 * u + j]);
                split8(v, nxt[2 * t + u]);
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) cur[s][p] = nxt[s][p];
    }
    // output layer: one tile, rows 0..15 are the state's dimensions (registers 0..7 of either lane half)
    stage(1 + (NH - 1) * NT);
    f32x16 a = bias_tile(NH, 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const u32x4 wh = lds[(s * 3 + 0) * 64 + lane], wm = lds[(s * 3 + 1) * 64 + lane], wl = lds[(s * 3 + 2) * 64 + lane];
        a = mm(wh, cur[s][0], a); a = mm(wh, cur[s][1], a); a = mm(wm, cur[s][0], a);
        a = mm(wh, cur[s][2], a); a = mm(wm, cur[s][1], a); a = mm(wl, cur[s][0], a);
    }
    if (sample < B) {
#pragma unroll
        for (int j = 0; j < 8; ++j) out[(size_t)sample * D + kidx(0, h, j)] = a[j];
    }
}

// ---- host: pack W[rows, cols] (row-major) into the fragments of `ntile` row tiles x `nks` k-steps ----------------
static unsigned short bf16_top(float v, float* rest)
{
    unsigned u; memcpy(&u, &v, 4);
    const unsigned t = u & 0xFFFF0000u;
    float tf; memcpy(&tf, &t, 4);
    *rest = v - tf;
    return (unsigned short)(t >> 16);
}
static void pack_layer(const std::vector<float>& W, int rows, int cols, int ntile, int nks, bool tile_major_slots,
                       std::vector<unsigned>& dst, size_t slot0)
{
    // tile_major_slots: one slot per tile (48 = 16 k-steps x 3); else all tiles in ONE slot (layer 1: 8 tiles x 1 k-step x 3)
    for (int t = 0; t < ntile; ++t)
        for (int s = 0; s < nks; ++s)
            for (int p = 0; p < 3; ++p) {
                const size_t slot = tile_major_slots ? slot0 + t : slot0;
                const int frag = tile_major_slots ? s * 3 + p : t * 3 + p;
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int row = 32 * t + (l & 31), k = kidx(s, l >> 5, j);
                        float v = (row < rows && k < cols) ? W[(size_t)row * cols + k] : 0.f, r1, r2, r3;
                        const unsigned short hi = bf16_top(v, &r1), mid = bf16_top(r1, &r2), lo = bf16_top(r2, &r3);
                        const unsigned short e = p == 0 ? hi : (p == 1 ? mid : lo);
                        unsigned& word = dst[((slot * 48 + frag) * 64 + l) * 4 + (j >> 1)];
                        word = (j & 1) ? ((word & 0x0000FFFFu) | ((unsigned)e << 16)) : ((word & 0xFFFF0000u) | e);
                    }
            }
}

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 4096;
    srand(11);
    auto rnd = [](float s) { return (rand() / (float)RAND_MAX * 2.f - 1.f) * s; };
    std::vector<std::vector<float>> W(NH + 1), b(NH + 1);
    const int ins[NH + 1] = {D, H, H, H, H}, outs[NH + 1] = {H, H, H, H, D};
    for (int l = 0; l <= NH; ++l) {
        W[l].resize((size_t)outs[l] * ins[l]); b[l].resize(outs[l]);
        for (auto& v : W[l]) v = rnd(1.0f / sqrtf((float)ins[l]) * 1.7f);
        for (auto& v : b[l]) v = rnd(0.3f);
    }
    std::vector<float> hx((size_t)B * D);
    for (auto& v : hx) v = rnd(2.0f);
    const int nslots = 1 + (NH - 1) * NT + 1;
    std::vector<unsigned> frag((size_t)nslots * 48 * 64 * 4, 0u);
    pack_layer(W[0], H, D, NT, 1, false, frag, 0);
    for (int l = 1; l < NH; ++l) pack_layer(W[l], H, H, NT, 16, true, frag, 1 + (size_t)(l - 1) * NT);
    pack_layer(W[NH], D, H, 1, 16, true, frag, 1 + (size_t)(NH - 1) * NT);
    std::vector<float> hb((size_t)(NH + 1) * H, 0.f);
    for (int l = 0; l <= NH; ++l) memcpy(&hb[(size_t)l * H], b[l].data(), b[l].size() * 4);
    u32x4* dfrag; float *dbias, *dx, *dout;
    CK(hipMalloc(&dfrag, frag.size() * 4)); CK(hipMalloc(&dbias, hb.size() * 4));
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dout, hx.size() * 4));
    CK(hipMemcpy(dfrag, frag.data(), frag.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)split_mlp_forward, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipLaunchKernelGGL(split_mlp_forward, dim3((B + 127) / 128), dim3(256), 49152, 0, dfrag, dbias, dx, dout, B);
    CK(hipDeviceSynchronize());
    std::vector<float> got((size_t)B * D);
    CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
    // float64 reference and an fp32 (fma chain) one for comparison
    double worst = 0, worst32 = 0, scale = 0;
    for (int n = 0; n < B; ++n) {
        std::vector<double> a(hx.begin() + (size_t)n * D, hx.begin() + (size_t)(n + 1) * D);
        std::vector<float> a32(hx.begin() + (size_t)n * D, hx.begin() + (size_t)(n + 1) * D);
        for (int l = 0; l <= NH; ++l) {
            std::vector<double> o(outs[l]); std::vector<float> o32(outs[l]);
            for (int i = 0; i < outs[l]; ++i) {
                double s = b[l][i]; float s32 = b[l][i];
                for (int k = 0; k < ins[l]; ++k) { s += (double)W[l][(size_t)i * ins[l] + k] * a[k]; s32 = fmaf(W[l][(size_t)i * ins[l] + k], a32[k], s32); }
                if (l < NH) { s = s / (1.0 + exp(-s)); s32 = s32 / (1.0f + expf(-s32)); }
                o[i] = s; o32[i] = s32;
            }
            a = o; a32 = o32;
        }
        for (int d = 0; d < D; ++d) {
            worst = fmax(worst, fabs(got[(size_t)n * D + d] - a[d]));
            worst32 = fmax(worst32, fabs(a32[d] - a[d]));
            scale = fmax(scale, fabs(a[d]));
        }
    }
    printf("B=%d  max |split kernel - float64| = %.3e   max |fp32 host chain - float64| = %.3e   (max |output| %.3f)\n", B,
           worst, worst32, scale);
    printf("%s\n", worst < 20 * worst32 + 1e-6 ? "PASS: split-precision chain is fp32-class" : "FAIL");
    return worst < 20 * worst32 + 1e-6 ? 0 : 1;
}
