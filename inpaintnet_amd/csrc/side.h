// Second HIP stream for "leaf" work of the backward pass.  The BPTT chains are latency-bound step kernels
// (one wave per SIMD, L2->CU load path bound) that leave most of the MFMA throughput idle, while the
// weight-gradient GEMMs and bias column sums have no consumer before the optimizer step.  side_fork() makes
// the (lower-priority) side stream wait for the main stream's current point and returns it; side_join() makes
// the main stream wait for everything queued on the side stream.  Every *_bwd entry point joins before it
// returns, so workspaces never outlive the work that reads them.
#pragma once
#include <hip/hip_runtime.h>

hipStream_t side_fork(hipStream_t main_stream);   // returns main_stream itself when the side stream is disabled
int side_join(hipStream_t main_stream);
void side_set_enabled(int on);
int side_enabled();
