// wn_stacked_table.h -- the table of instantiated stacked-layer kernels (variant 4, wn_kernel_v4.h), shared by the two translation units of the
// library: wn_stacked.hip DEFINES the table (kernel instantiations, weight packers, launchers), wn_runtime.hip plans and launches through it.
// Why two translation units (round 6): the library's persistent chains are compiled with -mllvm -align-all-nofallthru-blocks=6 (build.py: +0.5 % on the
// 64-stream headline), a per-translation-unit switch -- and the stacked kernels LOSE 1.2-1.3 % with it (same-box A/B of one source, three passes:
// cfg2 x 1 48.9 k with / 49.5 k without, cfg1 x 1 131.5 k / 133.1 k: profiles/r06_small_shape_regression_ab.txt).  They get a unit of their own, built without it.
#ifndef WN_STACKED_TABLE_H
#define WN_STACKED_TABLE_H

#include <hip/hip_runtime.h>

#include <vector>

#include "wn_plan.h"

#ifndef WN_THREADS_V4
#define WN_THREADS_V4 512   // threads of a stack workgroup (wn_kernel_v4.h)
#endif

// Shapes the stacked kernel (wn_kernel_v4.h: LPW consecutive layers per 512-thread workgroup) is instantiated for: (R, D, S, E / PA, LPW).
// LPW is what one CU's register file holds next to the working set: cfg2 60 registers per layer and lane (tap 0 lives in LDS), cfg1 32,
// the train_script.py shape 83 (its 1024 skip rows).
struct WnV4Entry {
    int R, D, S, EC, LPW, nwpl, nwh;
    void (*pack)(const WnPlan& pl, const WnHostWeights& w, std::vector<float>& out);
    const void* fn;
    int (*lds_floats)(int ns);
    int lds_pre_head;   // float offset of the head / sampler workgroups' own tables (WnV3Lds<SH>::pre)
    void (*launch)(int grid, size_t lds, hipStream_t st, const WnPlan& p, const WnRun& r);
};

const std::vector<WnV4Entry>& wn_v4_table();

#endif  // WN_STACKED_TABLE_H
