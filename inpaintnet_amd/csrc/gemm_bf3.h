// Products on the bf16 matrix cores at fp32 accuracy (gemm_bf3.hip): operands are three exact bf16 pieces per f32 value
// (x = x0 + x1 + x2), stored in the MFMA's fragment order.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// A "piece buffer" of a [R, K] operand (R rows of the product, K = contraction):
//   piece p (0..2) at byte offset p * piece_bytes;  inside a piece: [R/16 row blocks][KB k blocks][64 lanes][8 bf16],
//   lane = (k % 32) / 8 * 16 + r % 16 holds X(16 rb + r % 16, 32 kb + 8 (lane / 16) + 0..7): 1 KB per fragment.
// KB (`kb_total`) may exceed the K of one source array: several arrays are split side by side along K (`kb0`).
inline size_t bf3_piece_bytes(long R, long K) { return (size_t)R * (size_t)K * 2; }
inline size_t bf3_bytes(long R, long K) { return 3 * bf3_piece_bytes(R, K); }

// X(r, k) = kmajor ? X[k * ld + r] : X[r * ld + k]  ->  row blocks rb0.., k blocks kb0.. of the piece buffer P.
// R % 16 == 0 (k-major sources: R % 64 == 0), K % 32 == 0.
int bf3_split(const float* X, long ld, int kmajor, int R, int K, unsigned char* P, long piece_bytes, int kb_total, int rb0,
              int kb0, hipStream_t s);

// the k-contiguous form with a destination row-block stride: source row block rb -> row block rb0 + rb * rb_mul (interleaves the
// gates of a GRU weight matrix: gru_step_bf3.hip)
int bf3_split_strided(const float* X, long ld, int R, int K, unsigned char* P, long piece_bytes, int kb_total, int rb0, int rb_mul,
                      hipStream_t s);

// ... and up to 8 such splits of one shape in one launch
struct Bf3SplitJob { const float* X; unsigned char* P; int rb0, rb_mul; };
int bf3_split_strided_batch(const Bf3SplitJob* jobs, int n, int R, int K, long ld, long piece_bytes, int kb_total, hipStream_t s);

struct Bf3Gemm {
    // C[M, N] (op)= epi(sum_k A(m, k) B(n, k) + bias):  A, B piece buffers (k blocks 0 .. K/32 of each row block)
    const unsigned char* A; long a_piece; int a_kb;
    const unsigned char* B; long b_piece; int b_kb;
    float* C; long ldc;
    int M, N, K;
    const float* bias;                        // [N] or null; columns >= bias2_from take bias2[col - bias2_from]
    const float* bias2; int bias2_from;
    const float* aux; long ldaux; int epi;    // EPI_NONE or EPI_MUL_AUX (aux indexed like C)
    int acc;                                  // ACC_STORE / ACC_ADD (a k range split over the grid accumulates with f32 atomics)
    int ksplit;                               // 0: chosen by the launcher
    int nbatch;                               // 2: a second product of the same shape in the same launch
    const unsigned char* A2; const unsigned char* B2; float* C2;
    // A's row blocks >= a_alt_from (0: none) come from another piece buffer (row block rb - a_alt_from of it, piece stride
    // a_alt_piece, the same k blocks per row block): lets two products share most of an operand (the r and z gate gradients
    // of dgi and dgh)
    int a_alt_from; const unsigned char* A_alt; const unsigned char* A2_alt; long a_alt_piece;
};
// true when a tile configuration covers the shape exactly (M % 192 == 0, N % 128 == 0, K % 32 == 0); workspaces are carved by
// shape alone, callers use the path when bf3_mode() != 0 as well
bool gemm_bf3_ok(int M, int N, int K);
int launch_gemm_bf3(const Bf3Gemm& g, hipStream_t s);
// piece products per element product: 9 (all of them: the products of fp32 arithmetic); 0 switches the bf3 products off
// (a six-product form that dropped the three terms below 2^-24 |ab| existed in round 3: faster, not fp32 arithmetic, removed)
int bf3_mode();
void bf3_set_mode(int m);
