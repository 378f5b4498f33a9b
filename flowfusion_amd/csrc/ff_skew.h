// ff_skew.h -- test-only skew injection for the kernels that share LDS between the wavefronts of a workgroup
// (tests/test_gpu_skew.py; the macros below are never defined in the product library).
//
// The cooperative kernels' two synchronisation holes of round 3 only showed when the wavefronts did not run in step
// (another process on the card).  -DFF_DEBUG_SKEW=<w> builds kernels whose wavefront <w> is held back with s_sleep at
// exactly the points where a late wavefront does damage if a barrier is missing:
//   * mlp_ode_kernel<..., COOP> (ff_mlp_ode.hpp): before and after its zero fill of the shared stage slots, and between the
//     barrier of every activation exchange and its reads of the exchange buffer -- while the OTHER wavefronts wait between
//     storing the caller's first stage and reading it back, so that the late zero fill falls into that window
//     deterministically, not once in a thousand runs;
//   * split::mlp_ode_split_kernel (ff_mlp_ode_split.hpp): before its first weight DMA and behind the barrier of every weight
//     granule, i.e. late to read a buffer the others' next DMAs must not touch yet.
// With correct synchronisation the results are bitwise those of the un-skewed kernels.  -DFF_DEBUG_UNFIX removes round 3's two
// fixes again (the barrier behind the zero fill; exchange buffers alternating over the whole launch) so that the test can
// show its own teeth: the skewed un-fixed kernel gives wrong numbers, solo, every time.
#pragma once

#ifdef FF_DEBUG_SKEW
namespace ff { constexpr int kSkewWave = FF_DEBUG_SKEW; }
#define FF_SKEW_HOLD(cond, units)                                                                                      \
    do {                                                                                                               \
        if (cond)                                                                                                      \
            for (int ff_skew_i = 0; ff_skew_i < 4 * (units); ++ff_skew_i) __builtin_amdgcn_s_sleep(127); /* ~15 us a unit */ \
    } while (0)
#else
namespace ff { constexpr int kSkewWave = -1; }
#define FF_SKEW_HOLD(cond, units) do { } while (0)
#endif
