// Scratch harness: launch mlp_ode_kernel<256,8,0,false> directly with synthetic weights/table and
// (in -DFF_DEBUG_STAMPS builds) dump wavefront 0's cycle stamps.
#include "../flowfusion_amd/csrc/ff_mlp_ode.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
int main(int argc, char** argv)
{
    const int TILE = 16, H = 256, DREGS = 4, NH = 4, D = 16;
    long long B = argc > 1 ? atoll(argv[1]) : (1 << 18);
    int n_evals = argc > 2 ? atoi(argv[2]) : 100;
    ff::Layout L = ff::make_layout(TILE, H, DREGS, 0, NH);
    std::vector<float> hw(L.total_floats);
    srand(3);
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.12f;
    int stride = 32 + H;
    std::vector<float> ht((size_t)n_evals * stride, 0.f);
    for (int e = 0; e < n_evals; ++e) {
        ff::RowHdr* h = (ff::RowHdr*)&ht[(size_t)e * stride];
        h->a = -0.01f; h->b = 0.01f; h->slot = e % 4; h->flags = (e % 4 == 3) ? 1u : 0u;
        for (int s = 0; s < 4; ++s) { h->cin[s] = (s < e % 4) ? 0.01f : 0.f; h->cout[s] = 0.0025f; }
        for (int i = 0; i < H; ++i) ht[(size_t)e * stride + 32 + i] = ((rand() / (float)RAND_MAX) - 0.5f) * 0.1f;
    }
    float *dw, *dt, *dx, *dy; unsigned long long* dbg;
    CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&dt, ht.size() * 4)); CK(hipMalloc(&dx, B * D * 4)); CK(hipMalloc(&dy, B * D * 4));
    CK(hipMalloc(&dbg, 4096 * 8)); CK(hipMemset(dbg, 0, 4096 * 8));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hx(B * D); for (auto& v : hx) v = ((rand() / (float)RAND_MAX) - 0.5f) * 2.f;
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    ff::KernelArgs a; memset(&a, 0, sizeof(a));
    a.x_in = dx; a.x_out = dy; a.wpack = dw; a.etab = dt; a.batch = B; a.n_evals = n_evals; a.n_hidden = NH; a.dim = D;
    a.etab_stride = stride; a.wpack_floats = (int)L.total_floats; a.debug_stamps = dbg;
    auto kern = ff::mlp_ode_kernel<16, 256, 4, 0, false, 3, 2>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    unsigned lds = 4u * ff::kSlots * (DREGS / 4) * 64 * 16;
    unsigned grid = (unsigned)((B + 63) / 64);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int t = 0; t < 3; ++t) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double flop = 2.0 * 206848 * n_evals * (double)B;
    printf("B=%lld evals=%d: %.3f ms  %.2f TFLOP/s\n", B, n_evals, best, flop / best / 1e9);
#ifdef FF_DEBUG_STAMPS
    std::vector<unsigned long long> st(4096); CK(hipMemcpy(st.data(), dbg, 4096 * 8, hipMemcpyDeviceToHost));
    // per eval: stamps: [eval start], L1:[start, (A->B none since GA=0), end], hidden x3: [start, A->B, end], out: [start, A->B, end]
    int per = 1 + 3 + 3 * 4 + 4;
    for (int e = 2; e < 6; ++e) {
        printf("eval %d:", e);
        for (int i = 0; i < per; ++i) printf(" %llu", st[e * per + i + 1] - st[e * per + i]);
        printf("\n");
    }
    printf("legend: evalstart->L1start, L1 mfma, L1 park, gap | H0 A, H0 B, park, gap | H1 A,B,park,gap | H2 A,B,park,gap | OUT A, B, park, ->next eval\n");
#endif
    return 0;
}
