// Host-side sequencing of the fused step kernels: one GRU layer (1..4 independent
// "directions"/problems per launch) forward and backward through time, the
// batched weight-gradient contractions, and the 2-layer bidirectional stack used
// by the MeasureVAE encoder and the LatentRNN context/generator GRUs.
// All activations are time-major: row (t,b) of a sequence buffer lives at
// base + t*ts + b*ld.
#pragma once
#include "common.h"
#include "pointwise.h"
#include "side.h"
#include "gru_chain.h"

#define INET_TRY(expr) do { int _rc = (expr); if (_rc != 0) return _rc; } while (0)

struct Carver {
    char* base; size_t off;
    explicit Carver(void* b) : base((char*)b), off(0) {}
    template <typename T> T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? (T*)(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
    size_t bytes() const { return (off + 255) & ~(size_t)255; }
};

struct DirFwd {
    const float* W_hh; const float* b_hh;
    const float* gi; long gi_ld, gi_ts;                       // dense input-side pre-activations, or null
    const float* table; long table_ld;                        // gathered rows, or null
    const long long* idx; long idx_bs, idx_ts;                // token of (t,b) = idx[b*idx_bs + t*idx_ts]
    const float* gvec;                                        // [3H] broadcast, or null
    const float* h0; long h0_ld;                              // initial hidden [B,H]
    float* out; long out_ld, out_ts;                          // raw outputs
    float* outm; long outm_ld, outm_ts;                       // dropout-masked outputs (or null)
    const float* mask; long mask_ld, mask_ts;
    float* hlast; long hlast_ld;                              // extra copy of the final hidden (or null)
    float* sv; long sv_astride;                               // 5 saved arrays (r,z,n,ghn,hprev), each [T][B][H]; or null
    int reverse;
    // fragment-major fast path (both or neither; H % 256 == 0): packed W_hh and a 2 x pk_floats(B,H) ping-pong buffer
    const float* Wpk_hh; float* hpk;
    // chain kernel (gru_chain.h): kChainSyncWords words for the launch's group counters, given for direction 0 (or null)
    unsigned* sync;
    int sync_prezeroed;                                       // those words are zero already (one memset per library call)
    // piece outputs for the bf16-matrix-core products (gru_chain.h ChainEmit; B_full / r0 are filled in by gru_layer_fwd);
    // `emitted` is set by gru_layer_fwd / gru_layer_bwd to what the kernels they launched wrote (bit 0: rows, bit 1: the transposed
    // pieces; 0: nothing) -- the caller splits the rest from the f32 arrays with bf3_split
    ChainEmit em; mutable int emitted;
    // big batches (gru_step_bf3.h): scratch for the interleaved bf16 pieces of W_hh, gru_step_bf3_w_bytes(H) bytes; with it (and
    // hpk) a layer whose single time step fills the chip runs one bf16-pipe product per step instead of chunked chain launches
    unsigned char* wp3;
};

struct DirBwd {
    const float* W_hhT;                                       // [H,3H]
    const float* dout; long dout_ld, dout_ts;                 // dLoss/d out(t), or null
    const float* dhn; long dhn_ld;                            // dLoss/d final hidden, or null
    const float* sv; long sv_astride;
    float* dgi; long dgi_ld, dgi_ts;                          // [T][B][3H] (strided)
    float* dgh;                                               // [T][B][3H] dense
    float* dhz;                                               // [2][B][H]
    float* db_ih; float* db_hh;                               // bias gradients (fused into the step kernel), or null
    float* dh0; long dh0_ld; int dh0_acc;                     // dLoss/d initial hidden, or null
    int reverse;
    // fragment-major fast path (both or neither; H % 256 == 0): packed W_hh^T and a 2 x pk_floats(B,3H) ping-pong buffer
    const float* Wpk_hhT; float* dghpk;
    const float* W_hh;                                        // [3H,H] row-major (chain kernel reads it transposed once)
    unsigned* sync;                                           // as in DirFwd
    int sync_prezeroed;
    unsigned char* wp3T;                                      // big batches (gru_step_bf3.h): scratch for the pieces of W_hh^T, or null
    float* dgi_sum;                                           // optional [B,3H] sum_t dgi(t); `*dgi_sum_done` is set to 1 when the
    int* dgi_sum_done;                                        // layer's launch produced it (chain kernel), else left alone
    ChainEmit em; mutable int emitted;                        // as in DirFwd
};

// floats of a fragment-major [rows,K] operand (rows padded to 16)
inline size_t pk_floats(int rows, int K) { return (size_t)((rows + 15) / 16) * 16 * (size_t)K; }
inline bool pk_ok(int H) { return H % 256 == 0; }
// floats of a chain kernel's exchange ring over a [rows, K] state: two slots of fp32 fragments (first generation) or two slots
// of three bf16 pieces (second generation: 12 bytes per element) -- sized for the larger
inline size_t chain_ring_floats(int rows, int K) { return 3 * pk_floats(rows, K); }

// rows per chain launch for a batch of B rows (B itself when one launch holds it; 0: no chain kernel applies)
int chain_chunk_rows(int H, int B, int T, int nd, int save = 1);
int chain_chunk_rows_bwd(int H, int B, int T, int nd);
// whether gru_layer_fwd / gru_layer_bwd write ChainEmit outputs for this shape (the kernels that run are the second generation's,
// rows in multiples of 32): the same decision the layer functions make, for callers that must know it in another library call
bool gru_layer_fwd_emits(int H, int B, int T, int nd, bool save);
void bf3_set_emit_mask(int m);      // which piece outputs the chain kernels write themselves (inet_set_option key 9)
int gru_layer_fwd(int H, int B, int T, int nd, const DirFwd* d, hipStream_t s);
int gru_layer_bwd(int H, int B, int T, int nd, const DirBwd* d, hipStream_t s);
int gru_layer_bwd_range(int H, int B, int T, int nd, const DirBwd* d, int step_hi, int step_lo, hipStream_t s);
// dW_hh += dgh^T hprev       (rows = T*B, contiguous in t then b; the bias gradients come from the step kernel)
int gru_dir_wgrad(int H, int B, int T, const float* dgh, const float* sv_hprev, float* dW_hh, hipStream_t s);
int gru_dir_wgrad_range(int H, int B, int t_lo, int nt, const float* dgh, const float* sv_hprev, float* dW_hh, hipStream_t s);

inline GemmArgs gemm_args(const float* A, long lda, int akm, const float* Bm, long ldb, int bkm, float* C, long ldc,
                          int M, int N, int K, const float* bias = nullptr, int epi = EPI_NONE,
                          const float* aux = nullptr, long ldaux = 0, int acc = ACC_STORE) {
    GemmArgs g{};
    g.A = A; g.lda = lda; g.a_kmajor = akm; g.B = Bm; g.ldb = ldb; g.b_kmajor = bkm; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.bias = bias; g.aux = aux; g.ldaux = ldaux; g.epi = epi; g.acc = acc;
    g.k_per_split = K;
    return g;
}
// y[M,N] = epi(x[M,K] W[N,K]^T + b)                       (nn.Linear forward)
inline int linear_fwd(const float* x, long ldx, const float* W, long ldw, const float* b, float* y, long ldy, int M,
                      int N, int K, int epi, hipStream_t s) {
    return launch_gemm(gemm_args(x, ldx, 0, W, ldw, 0, y, ldy, M, N, K, b, epi), s);
}
inline GemmArgs linear_fwd_args(const float* x, long ldx, const float* W, long ldw, const float* b, float* y, long ldy,
                                int M, int N, int K, int epi) {
    return gemm_args(x, ldx, 0, W, ldw, 0, y, ldy, M, N, K, b, epi);
}
inline GemmArgs linear_dgrad_args(const float* dy, long lddy, const float* W, long ldw, float* dx, long lddx, int M, int N,
                                  int K, int epi, const float* aux, long ldaux, int acc) {
    return gemm_args(dy, lddy, 0, W, ldw, 1, dx, lddx, M, K, N, nullptr, epi, aux, ldaux, acc);
}
inline GemmArgs linear_wgrad_args(const float* dy, long lddy, const float* x, long ldx, float* dW, long lddw, int M, int N,
                                  int K) {
    return gemm_args(dy, lddy, 1, x, ldx, 1, dW, lddw, N, K, M, nullptr, EPI_NONE, nullptr, 0, ACC_ADD);
}
// dx[M,K] (op)= epi(dy[M,N] W[N,K])                        (dgrad)
inline int linear_dgrad(const float* dy, long lddy, const float* W, long ldw, float* dx, long lddx, int M, int N, int K,
                        int epi, const float* aux, long ldaux, int acc, hipStream_t s) {
    return launch_gemm(gemm_args(dy, lddy, 0, W, ldw, 1, dx, lddx, M, K, N, nullptr, epi, aux, ldaux, acc), s);
}
// dW[N,K] += dy[M,N]^T x[M,K]                              (wgrad)
inline int linear_wgrad(const float* dy, long lddy, const float* x, long ldx, float* dW, long lddw, int M, int N, int K,
                        hipStream_t s) {
    return launch_gemm(gemm_args(dy, lddy, 1, x, ldx, 1, dW, lddw, N, K, M, nullptr, EPI_NONE, nullptr, 0, ACC_ADD), s);
}
// two weight gradients of one shape in one launch: dW_i[N,K] += dy_i[M,N]^T x_i[M,K], i = 0, 1
inline int linear_wgrad2(const float* dy0, const float* dy1, long lddy, const float* x0, const float* x1, long ldx,
                         float* dW0, float* dW1, long lddw, int M, int N, int K, hipStream_t s) {
    GemmArgs g = gemm_args(dy0, lddy, 1, x0, ldx, 1, dW0, lddw, N, K, M, nullptr, EPI_NONE, nullptr, 0, ACC_ADD);
    g.nbatch = 2; g.batchA = dy1 - dy0; g.batchB = x1 - x0; g.batchC = dW1 - dW0;
    return launch_gemm(g, s);
}

// ---- 2-layer bidirectional GRU core --------------------------------------------------
struct GruDirPtr { const float *w_ih, *w_hh, *b_ih, *b_hh; float *dw_ih, *dw_hh, *db_ih, *db_hh; int K; };

struct BiGru2Ws {
    float *zeros, *x1raw, *x1m, *gi1, *h1, *sv[4];
    float *whhT[4], *dgi1, *dgh[4], *dhz, *dx1, *dgi0;
    float *wpk[4], *hpk[4], *wpkT[4], *dghpk[4];               // fragment-major twins (null unless pk_ok(H))
    unsigned char* wp3[4];                                       // interleaved W_hh pieces for the big-batch step kernels (gru_step_bf3.h), or null
    unsigned char* wp3T[4];                                      // ... and the pieces of W_hh^T for their backward steps (save only)
    unsigned* sync;                                            // chain-kernel counters: kSyncAreas areas (gru_chain.h)
    // piece buffers of the layer-1 input products (gemm_bf3.h; null unless the shapes tile): x1 [TB, 2H], W_ih of both
    // layer-1 directions stacked [6H, 2H]; backward: dgi1 [TB, 6H], the same weights k-major [2H, 6H]
    unsigned char *x1pk, *wih1pk, *dgi1pk, *wih1Tpk;
    // weight-gradient operands, contraction over the T*B rows (k-major sources): per layer the gate gradients transposed,
    // gT[l] [6H = dir x (r, z, n), TB] (= dgi^T; dgh^T shares its r and z blocks and takes n*r from nrT[2l + dir] [H, TB]),
    // the layer-0 output x1T [2H, TB] and the previous hidden states hpT[2l + dir] [H, TB]
    unsigned char *gT[2], *nrT[4], *x1T, *hpT[4];
};
size_t bigru2_carve(Carver& c, int B, int T, int H, int save, BiGru2Ws& w);

// Layer-0 input-side pre-activations come from the caller, per direction d:
//   dense  gi0[d]   rows (t,b) at gi0[d] + t*gi0_ts + b*gi0_ld     (or null)
//   table  tab[d] (+ idx, idx_bs, idx_ts)                           (or null)
//   gvec   gvec[d]                                                  (or null)
struct BiGru2In {
    const float* gi0[2]; long gi0_ld, gi0_ts;
    const float* tab[2]; long tab_ld; const long long* idx; long idx_bs, idx_ts;
    const float* gvec[2];
};
// h0: [4][B][H] or null.  mask: [T][B][2H] or null.  hn4: 4 destinations (ld given) for final hiddens or null.
// sync_prezeroed: the caller has zeroed w.sync (all kSyncAreas areas) on `s` already (the encoder's prologue launch does)
int bigru2_core_fwd(int B, int T, int H, const GruDirPtr* P /*[4]*/, const BiGru2In& in, const float* h0,
                    const float* mask, float* const* hn, long hn_ld, BiGru2Ws& w, int save, hipStream_t s,
                    int sync_prezeroed = 0);
// dout1: gradient wrt the top layer outputs [T][B][2H] (or null); dhn[4] (ld) gradients wrt final hiddens (or null each).
// Produces w.dgi0 [T][B][6H] (layer-0 input-side gate gradients, fwd dir cols 0..3H, reverse 3H..6H), accumulates the
// recurrent / layer-1 weight gradients into P[*].d* (skipped when P[0].dw_hh is null), dh0 [4][B][H] (or null).
// stage 0 = the whole backward pass; 1 = layer 1 only (its gradients are final afterwards: a data-parallel bucket can start);
// 2 = the rest (layer 0), on the state stage 1 left in the workspace
int bigru2_core_bwd(int B, int T, int H, const GruDirPtr* P, const float* mask, const float* dout1,
                    const float* const* dhn, long dhn_ld, float* dh0, BiGru2Ws& w, hipStream_t s, int stage = 0);
