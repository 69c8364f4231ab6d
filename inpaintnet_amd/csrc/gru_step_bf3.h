// GRU forward steps for large batches on the bf16 matrix cores (gru_step_bf3.hip): one launch per time step, gemm_bf3's main loop
// with the GRU cell as its epilogue.  Descriptor = the chain kernels' (gru_chain.h GruChainFwdProb: `hx` is the two-slot ring of
// three-piece states, chain_ring_floats(B, H) floats; W_hh is not read) plus the interleaved pieces of W_hh per direction.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "gru_chain.h"

struct GruStepsBf3 {
    int H, B, T, nprob;
    GruChainFwdProb p[2];
    const unsigned char* Wp[2];          // gru_step_bf3_split_w() of W_hh: gru_step_bf3_w_bytes(H) bytes each
};
// shape rule: bf3 products on, H % 64 == 0, B % 128 == 0, one launch fills the chip (>= 256 tiles of 128 rows x 64 units;
// inet_set_option key 12 overrides, 0 = never)
bool gru_step_bf3_ok(int H, int B, int T, int nd);
void gru_step_bf3_set_min_tiles(int n);                    // inet_set_option key 12 (0: never take this path)
size_t gru_step_bf3_w_bytes(int H);
int gru_step_bf3_split_w(int H, const float* const* W_hh, unsigned char* const* Wp, int nd, hipStream_t s);   // nd directions, one launch
// all T steps (T launches on `s`); writes ChainEmit.rows if given (nothing else of the descriptor's `em`)
int launch_gru_steps_bf3(const GruStepsBf3& L, hipStream_t s);

// Backward steps of the same layers (gru_step_bf3_bwd_kernel): descriptor = the chain kernels' BPTT descriptor (`gx` = the two-slot
// ring of three-piece gate gradients, chain_ring_floats(B, 3H) floats) + the pieces of W_hh^T + the [2][B][H] dh z buffer per
// direction.  Writes dgi, dgh, the bias gradients (column sums after the last step), dh0 if asked, ChainEmit.rows if given.
struct GruStepsBf3Bwd {
    int H, B, T, nprob;
    GruChainBwdProb p[2];
    const unsigned char* WpT[2];         // gru_step_bf3_split_wT() of W_hh: gru_step_bf3_w_bytes(H) bytes each
    float* dhz[2];
};
bool gru_step_bf3_bwd_ok(int H, int B, int T, int nd);
int gru_step_bf3_split_wT(int H, const float* W_hh, unsigned char* WpT, hipStream_t s);
int launch_gru_steps_bf3_bwd(const GruStepsBf3Bwd& L, hipStream_t s);
