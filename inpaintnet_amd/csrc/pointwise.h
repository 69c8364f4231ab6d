// Host-side launchers of the HBM-bound helper kernels (pointwise.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

int pw_transpose(const float* in, long ld_in, float* out, long ld_out, int rows, int cols, hipStream_t s);
// row-major [R,K] (or its transpose) -> fragment-major [ceil(R/16)][K/16][64][4] (ksplit.h); nbatch strided matrices
int pw_pack_frag(const float* in, long ld, int R, int K, float* out, int transposed, int nbatch, long in_bstride,
                 long out_bstride, hipStream_t s);
// the same for up to 8 equally shaped matrices in one launch
int pw_pack_frag_multi(const float* const* ins, float* const* outs, int n, long ld, int R, int K, int transposed,
                       hipStream_t s);
int pw_cross_entropy(const float* W, long ld_w, int rows, int V, const long long* tgt, float* dW, long ld_dw,
                     float scale, float out_scale, float* loss_sum, float* correct, hipStream_t s,
                     const float* scale_dev = nullptr, const float* add_term = nullptr, float add_scale = 0.f,
                     float* fwd_out = nullptr, float fwd_scale = 0.f);
int pw_reparam_kl(const float* mu, const float* ls, const float* eps, float* z, float* sigma, long n, float* kl_sum,
                  hipStream_t s);
int pw_latent_bwd(const float* dz, const float* mu, const float* ls, const float* eps, float kscale, const float* kdev,
                  float* dmu, float* dls, long n, hipStream_t s);
int pw_sample_multinomial(const float* W, long ld_w, int rows, int V, long long* out, long stride, uint64_t seed,
                          uint64_t offset, hipStream_t s);
// step_flag (optional device float): non-zero = skip (the ranks' summed chain status); report (optional, 4 host-mapped words
// zeroed by the caller): [0] = 1 executed, [1] = 1 skipped, [2] = 1 a parameter became non-finite
int pw_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, int step,
            float gscale, hipStream_t s, const float* step_flag = nullptr, unsigned* report = nullptr);
int pw_step_flag_export(float* dst, hipStream_t s);
int pw_epoch_stats_add(float* sums, const float* loss, const float* acc, hipStream_t s, const float* step_flag = nullptr);
int pw_colsum(const float* X, long ld, int M, int N, float* out, hipStream_t s);
// up to 8 column sums (bias gradients of one module) in one launch: out_i[n] += sum_m X_i[m*ld_i + n]
struct PwColsumJob { const float* X; long ld; int M, N; float* out; };
int pw_colsum_multi(const PwColsumJob* jobs, int n, hipStream_t s);
int pw_onehot(const long long* idx, int inner, long s_outer, long s_inner, int rows, int W, float* out, int zero_first,
              hipStream_t s);
// out[v][c] = sum of the rows of X whose token is v (out [W][ncols], W <= 128); row r has token idx[(r/inner)*s_outer + (r%inner)*s_inner]
// (+= into `out` unless zero_first; row_scale [rows] optional factor per row)
int pw_token_segsum(const float* X, long ld, const long long* idx, int inner, long s_outer, long s_inner, int rows, int W,
                    int ncols, float* out, hipStream_t s, const float* row_scale = nullptr, int zero_first = 1);
// dW[d][N3][E] += dtab[:, d*N3:(d+1)*N3]^T emb ;  demb[W][E] += sum_d dtab[:, d*N3:(d+1)*N3] Wih[d]      (E <= 16, W <= 128)
// emb_last / demb_last (optional): row W-1 of emb / demb is a separate E-vector (the decoder's start symbol behind its V rows)
int pw_table_grad(const float* dtab, int W, int N3, int ndir, int E, const float* emb, long ld_emb, const float* const* Wih,
                  float* const* dW, long ldw, float* demb, long ld_demb, hipStream_t s, const float* emb_last = nullptr,
                  float* demb_last = nullptr);
// teacher-forced input tokens of a [B,T] target: out[b][0] = first, out[b][t] = target[b][t-1]
int pw_shift_tokens(const long long* target, int B, int T, long long first, long long* out, hipStream_t s);
int pw_mul(float* x, const float* m, long n, int selu_grad, hipStream_t s);
int pw_swap01(const float* in, int A, int B, int K, float* out, hipStream_t s);
int pw_argmax(const float* W, long ld_w, int rows, int V, long long* out, long stride, hipStream_t s);
int pw_zero2d(float* p, long ld, long rows, int cols, hipStream_t s);
int pw_zero(float* p, long n, hipStream_t s);
int pw_copy_bytes(void* dst, const void* src, long nbytes, hipStream_t s);     // nbytes % 4 == 0
int pw_fill_i64(long long* p, long n, long long v, hipStream_t s);
int pw_copy2d(float* dst, long ld_d, const float* src, long ld_s, const float* pos, long ld_p, int rows, int cols,
              hipStream_t s);
int pw_dlogits_relayout(const float* dW, const float* Wt, int B, int T, int V, float* out, hipStream_t s);
int pw_group_sum(const float* in, int groups, int G, long inner, float* out, hipStream_t s);
int pw_axpb(const float* a, const float* x, long incx, const float* b, float* y, int n, hipStream_t s);
// dw[i*incw] += b0 * sv[i] (sv = column sums of X [M, n]);  db0 += sum_{r,i} X[r][i] w[i*incw] in a FIXED order
// (`partial`: 64 floats of scratch): deterministic from run to run
int pw_beat_input_grad(const float* sv, const float* w, long incw, const float* b0, float* dw, float* db0, int n,
                       const float* X, long ld, int M, float* partial, hipStream_t s);
int pw_dropout_mask(float* out, long n, float p, uint64_t seed, uint64_t offset, hipStream_t s);
int pw_scale(float* x, long n, float a, hipStream_t s);
int pw_embedding_fwd(const float* table, const long long* idx, long rows, int E, float* out, const float* row_scale,
                     hipStream_t s);
int pw_embedding_bwd(const float* dout, const long long* idx, long rows, int E, float* dtable, const float* row_scale,
                     hipStream_t s, int num_embeddings = 0);
int pw_tokens_i32_to_i64(const int* src, long long* dst, long n, hipStream_t s);
int pw_split_measures(const int* score, int B, int M, int L, int n_past, int n_target, long long* past,
                      long long* target, long long* future, hipStream_t s);

// ---- one launch for the scattered little jobs in front of a module's forward pass (each used to be its own launch of
// 5 us): gather tables  out[r][n] = emb[r,:E] . W[n,:E] + bias[n]  (the embedding -> layer-0 input projection folded
// into a table, K = E ~ 10: not MFMA work), the zero-fill of the chain kernels' sync areas, a word copy, y = a*x + b,
// an int64 fill, and the teacher-forced token copy / shift.  Absent jobs have null pointers / zero counts.
struct PwTableJob { const float* emb; long ld_emb; int rows; const float* W; long ldw; const float* bias; float* out; long ld_out; int N; int E; };
struct PwPrologue {
    PwTableJob tab[4]; int ntab;
    unsigned* zero_words; long nzero;
    const unsigned* copy_src; unsigned* copy_dst; long ncopy;
    const float* axpb_a; const float* axpb_x; long axpb_incx; const float* axpb_b; float* axpb_y; int axpb_n;
    long long* fill_ptr; long nfill; long long fill_val;
    const long long* tok_src; long long* tok_copy; long long* tok_shift; int tok_B, tok_T; long long tok_first;
    int tok_V;                // > 0: indices of tok_src outside [0, tok_V) are counted into the host-mapped token status word
                              // (decoder.py:36-45 check_index; chain.h token_host_status) -- set by pw_prologue
    unsigned* tok_bad;
};
int pw_prologue(const PwPrologue& p, hipStream_t s);
