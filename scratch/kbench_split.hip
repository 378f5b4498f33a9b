// Scratch harness: launch split::mlp_ode_split_kernel<4, false> directly with synthetic weights / table and, in
// -DFF_SPLIT_STAMPS builds, print wavefront 0's cycle stamps of evaluations 2 and 3 (timing only; numbers are arbitrary).
//   hipcc -O3 -std=c++17 -Wno-inline-asm --offload-arch=gfx950 -DFF_SPLIT_STAMPS -Iflowfusion_amd/csrc -Iinclude scratch/kbench_split.hip -o /tmp/kbs
#include "ff_mlp_ode_split.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
int main(int argc, char** argv)
{
    using namespace ff::split;
#ifndef KB_NH
#define KB_NH 4
#endif
#ifndef KB_NP
#define KB_NP 3
#endif
    const int NH = KB_NH, H = 256, D = 16, NP = KB_NP;
    long long B = argc > 1 ? atoll(argv[1]) : (1 << 20);
    int n_evals = argc > 2 ? atoi(argv[2]) : 100;
    const size_t nw = total_words(NH, NP);
    std::vector<uint32_t> hw(nw);
    srand(3);
    // fragments: two bf16 per word, magnitudes ~0.06 (hi), 2^-8 and 2^-16 of that for mid / lo: a granule is 8 x [hi, mid, lo] x 1 KiB
    for (size_t i = 0; i < stream_words(NH, NP); ++i) {
        const int part = (int)((i / 256) % NP);
        auto bf = [&](float scale) { float v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.12f * scale; uint32_t u; memcpy(&u, &v, 4); return u >> 16; };
        const float sc = part == 0 ? 1.f : (part == 1 ? 1.f / 256 : 1.f / 65536);
        hw[i] = bf(sc) | (bf(sc) << 16);
    }
    for (size_t i = stream_words(NH, NP); i < nw; ++i) { float v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.1f; memcpy(&hw[i], &v, 4); }
    const int stride = 32 + H;
    std::vector<float> ht((size_t)n_evals * stride, 0.f);
    for (int e = 0; e < n_evals; ++e) {
        ff::RowHdr* h = (ff::RowHdr*)&ht[(size_t)e * stride];
        h->a = -0.01f; h->b = 0.01f; h->slot = e % 4; h->flags = (e % 4 == 3) ? 1u : 0u;
        for (int s = 0; s < 4; ++s) { h->cin[s] = (s < e % 4) ? 0.01f : 0.f; h->cout[s] = 0.0025f; }
        for (int i = 0; i < H; ++i) ht[(size_t)e * stride + 32 + i] = ((rand() / (float)RAND_MAX) - 0.5f) * 0.1f;
    }
    if (getenv("KB_ZERO")) {          // all activations exactly zero: state 0, biases 0 (weights stay random)
        for (size_t i = stream_words(NH, NP); i < nw; ++i) hw[i] = 0;
        for (int e = 0; e < n_evals; ++e)
            for (int i = 0; i < H; ++i) ht[(size_t)e * stride + 32 + i] = 0.f;
    }
    if (getenv("KB_BIG")) {           // large pre-activations: sigmoid saturates (activation = pre-activation or 0)
        for (int e = 0; e < n_evals; ++e)
            for (int i = 0; i < H; ++i) ht[(size_t)e * stride + 32 + i] *= 400.f;
    }
    float *dw, *dt, *dx, *dy; unsigned long long* dbg;
    CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&dt, ht.size() * 4)); CK(hipMalloc(&dx, B * D * 4)); CK(hipMalloc(&dy, B * D * 4));
    CK(hipMalloc(&dbg, 4096 * 8)); CK(hipMemset(dbg, 0, 4096 * 8));
    CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hx(B * D); for (auto& v : hx) v = getenv("KB_ZERO") ? 0.f : ((rand() / (float)RAND_MAX) - 0.5f) * 2.f;
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    ff::KernelArgs a; memset(&a, 0, sizeof(a));
    a.x_in = dx; a.x_out = dy; a.wpack = dw; a.etab = dt; a.batch = B; a.n_evals = n_evals; a.n_hidden = NH; a.dim = D;
    a.etab_stride = stride; a.wpack_floats = (int)nw; a.debug_stamps = dbg;
    auto kern = mlp_ode_split_kernel<NH, 0, NP>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const unsigned lds = (unsigned)lds_map(H, NH, NP).total;   // (width 256)
    const unsigned grid = (unsigned)((B + 127) / 128);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int t = 0; t < 3; ++t) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("B=%lld evals=%d: %.3f ms  (%.1f us per evaluation and round of 32768 samples)\n", B, n_evals, best,
           best * 1e3 / n_evals / ((B + 32767) / 32768));
#ifdef FF_SPLIT_STAMP_GAPS
    {
        std::vector<unsigned long long> st(4096); CK(hipMemcpy(st.data(), dbg, 4096 * 8, hipMemcpyDeviceToHost));
        // start, stage input, layer 1, k-steps 0..2 (3 stamps) | 24 gap stamps | ...
        printf("hidden1 k-step 3: cycles from the end of k-step 2 to behind MFMA i of groups 0, 1:");
        for (int g = 0; g < 24; ++g) printf(" %llu", st[6 + g] - st[5]);
        printf("\n");
        return 0;
    }
#endif
#ifdef FF_SPLIT_STAMP_GROUPS
    {
        std::vector<unsigned long long> st(4096); CK(hipMemcpy(st.data(), dbg, 4096 * 8, hipMemcpyDeviceToHost));
        // stamps of an evaluation: start, stage input, layer 1, then per hidden layer: k-steps 0..2 (3 stamps), 16 group stamps of k-step 3,
        // k-steps 3..7 (5 stamps); ...
        int i = 3;
        for (int l = 0; l < NH - 1; ++l) {
            i += 3;
            printf("hidden%d k-step 3, cycles per group (12 MFMAs = 192 ideal):", l + 1);
            for (int g = 0; g < 16; ++g) printf(" %llu", st[i + g + 1] - st[i + g]);
            printf("\n");
            i += 16 + 5;
        }
        return 0;
    }
#endif
#ifdef FF_SPLIT_STAMPS
    std::vector<unsigned long long> st(4096); CK(hipMemcpy(st.data(), dbg, 4096 * 8, hipMemcpyDeviceToHost));
    const int per = 3 + 8 * NH + 1;          // start, stage input, layer 1, (NH-1) x 8 hidden k-steps + 8 output k-steps, before bookkeeping
    for (int e = 0; e < 2; ++e) {
        const unsigned long long* s = &st[e * per];
        printf("eval %d: stage-input %llu | layer1 %llu |", e + 2, s[1] - s[0], s[2] - s[1]);
        for (int l = 0; l < NH; ++l) {
            printf(l < NH - 1 ? " hidden%d:" : " out:", l + 1);
            for (int k = 0; k < 8; ++k) printf(" %llu", s[3 + l * 8 + k] - s[2 + l * 8 + k]);
            printf(" |");
        }
        if (e == 0) printf(" rhs+bookkeeping %llu | total %llu (MFMA-only ideal %d)\n", st[per] - s[per - 1], st[per] - s[0], (NP == 3 ? 4896 : 2448) * 16);
        else printf("\n");
    }
#endif
    return 0;
}
