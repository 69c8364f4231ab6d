// Optional in-library kernel timing with HIP events (used by bench.py for the
// roofline line; off by default and never on inside a timed region).  Every
// launch of the three MFMA kernel classes is bracketed by hipEventRecord on the
// launch stream; inet_prof_read() synchronises and reports per class the number
// of launches, the summed event time and the summed algorithmic FLOPs.
#pragma once
#include <hip/hip_runtime.h>

enum { PROF_GEMM = 0, PROF_GRU_FWD = 1, PROF_GRU_BWD = 2, PROF_HBM = 3, PROF_NCLASS = 4 };

struct ProfScope {
    int idx;
    hipStream_t s;
    // bytes: algorithmic (compulsory) HBM bytes of the launch -- operands read once, results written once
    ProfScope(int cls, double flops, hipStream_t stream, const char* label = nullptr, double bytes = 0.0);
    ~ProfScope();
};
