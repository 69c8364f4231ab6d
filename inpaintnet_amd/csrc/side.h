// Second HIP stream for "leaf" work of the backward pass.  The BPTT chains are latency-bound step kernels
// (one wave per SIMD, L2->CU load path bound) that leave most of the MFMA throughput idle, while the
// weight-gradient GEMMs and bias column sums have no consumer before the optimizer step.  side_fork() makes
// the (lower-priority) side stream wait for the main stream's current point and returns it; side_join() makes
// the main stream wait for everything queued on the side stream.  Every *_bwd entry point joins before it
// returns, so workspaces never outlive the work that reads them -- unless the caller has switched joins to
// "deferred" (inet_set_option(1, 1)): then it keeps the workspaces alive itself and calls inet_side_join() once,
// before the gradients are consumed (optimizer step / all-reduce), so leaf work of one module's backward can
// overlap the next module's BPTT chain.
#pragma once
#include <hip/hip_runtime.h>

hipStream_t side_fork(hipStream_t main_stream);   // returns main_stream itself when the side stream is disabled
int side_join(hipStream_t main_stream);            // no-op in deferred mode
int side_join_now(hipStream_t main_stream);
// `other` (not the issuing stream) waits for everything queued on the side streams so far; the dirty flags stay set, so
// the issuing stream still joins that work at its own next join
int side_wait_on(hipStream_t other);
void side_set_defer(int on);
void side_set_enabled(int on);
void side_set_active(int n);                        // side streams used in rotation from now on (1 .. created; 0: all) -- option key 13
int side_enabled();
bool side_is(hipStream_t s);                         // s is the side stream
// A second stream of the main stream's kind for work that is split over two queues on purpose (row chunks of the frozen
// encoder's chain launches): twin_fork() makes it wait for the main stream's current point, twin_join() the reverse.
hipStream_t twin_fork(hipStream_t main_stream);      // returns main_stream itself if no stream could be created
int twin_join(hipStream_t main_stream);
hipStream_t twin_stream();                           // the twin stream itself (created on first use; null if it could not be)
// `waiter` waits for everything queued on `on` so far (one event)
int stream_wait(hipStream_t waiter, hipStream_t on);
// A non-atomic accumulation into `dest` is about to be queued on side stream `s`: if another side stream has queued one into
// the same tensor since the last join, `s` first waits for that stream (two applications of one module in a step).
int side_order_dest(const void* dest, hipStream_t s);
// csrc/preload.hip: first-touch of every kernel of the library on the current device (code objects + function objects, without a
// launch); idempotent per device; returns the number of kernels touched, -2 without a device
int preload_kernels();
int preload_kernel_count();
