// Shared device helpers and launch-argument structs for the gfx950 kernels.
// Everything in csrc/ targets CDNA4 (MI355X) only: 64-lane wavefronts, f32-input
// MFMA (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32), 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16-byte global load from a pointer that is only guaranteed 4-byte aligned
// (weight sub-blocks such as rnn_tick.weight_ih_l0[:, E:] start mid-row).
// gfx950 under HSA runs in unaligned-access mode; this lowers to one
// global_load_dwordx4.
struct __attribute__((packed, aligned(4))) f4u_t { float x, y, z, w; };
__device__ __forceinline__ f32x4 ld4u(const float* p) {
    f4u_t v = *reinterpret_cast<const f4u_t*>(p);
    return f32x4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void st4u(float* p, f32x4 v) {
    f4u_t s{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f4u_t*>(p) = s;
}

// NOTE on kernarg structs: never read a by-value kernel-argument struct through a type-punned pointer (an earlier
// "warm the kernarg with one batch of s_loads" helper did).  It defeats the compiler's proof that pointers stored in
// kernel arguments are global: every access through them becomes flat_load/flat_store, flat operations tick both
// vmcnt and lgkmcnt, and the waits degrade to vmcnt(0) lgkmcnt(0) -- a two-deep load pipeline silently becomes
// load -> drain -> compute (profiles/r01_f).
// hipcc sinks every `args.field` read to its first use, so a block of "if (P.ptr) v = P.ptr[i]" statements becomes a
// chain of s_load -> s_waitcnt lgkmcnt(0) -> branch round trips (3,000 cycles for the forward step's epilogue-operand
// block, tools/trace_steps.py).  kernarg_touch() names the fields such a block needs right in front of it: the empty
// asm wants each value in an SGPR, so the s_loads are issued back to back and waited for once.  Typed field reads
// only -- no punning.  (Touching ALL fields at kernel entry instead was slower: 40.1k vs 41.0k measures/s.)
template <class T>
__device__ __forceinline__ void kernarg_touch1(T v) { asm volatile("" ::"s"(v)); }
template <class... T>
__device__ __forceinline__ void kernarg_touch(T... v) { (kernarg_touch1(v), ...); }

#define SELU_ALPHA 1.6732632423543772f
#define SELU_SCALE 1.0507009873554805f

__device__ __forceinline__ float selu_f(float x) {
    return SELU_SCALE * (x > 0.f ? x : SELU_ALPHA * (expf(x) - 1.f));
}
// derivative of SELU expressed through its OUTPUT a = selu(x)
__device__ __forceinline__ float selu_grad_from_out(float a) {
    return a > 0.f ? SELU_SCALE : a + SELU_SCALE * SELU_ALPHA;
}
// Gate non-linearities of the recurrent step epilogues on the hardware transcendentals: v_exp_f32 and v_rcp_f32 are
// 1 ulp each, so both functions are good to ~3e-7 relative (absolute near tanh's zero) -- the same class of error as
// the libm-style expf/tanhf they replace, at ~8 instead of ~60 VALU instructions per gate (INET_EXACT_GATES restores
// the library forms).  Saturation is exact: exp -> inf gives sigmoid 0 / tanh 1, exp -> 0 gives sigmoid 1 / tanh -1.
#ifdef INET_EXACT_GATES
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) { return tanhf(x); }
#else
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }
#endif

// ---------------------------------------------------------------------------
// Generic fp32 MFMA GEMM:  C[M,N] (op)= epi( sum_k A(m,k) * B(n,k) )
// ---------------------------------------------------------------------------
enum { EPI_NONE = 0, EPI_SELU = 1, EPI_RELU = 2, EPI_MUL_SELU_GRAD = 3, EPI_MUL_AUX = 4, EPI_MUL_POS = 5 };
enum { ACC_STORE = 0, ACC_ADD = 1, ACC_ATOMIC = 2 };

struct GemmArgs {
    const float* A; long lda; int a_kmajor;   // 0: A(m,k) = A[m*lda + k]   1: A(m,k) = A[k*lda + m]
    const float* B; long ldb; int b_kmajor;   // 0: B(n,k) = B[n*ldb + k]   1: B(n,k) = B[k*ldb + n]
    float* C; long ldc;
    int M, N, K;
    const float* bias;                        // [N] or null
    const float* aux; long ldaux;             // epilogue operand, indexed like C
    int epi;                                  // EPI_*
    int acc;                                  // ACC_*
    int k_per_split;                          // K range handled by one blockIdx.z
    // nbatch > 1: that many products of one shape in one launch, problem i on A + i*batchA, B + i*batchB, C + i*batchC
    // (element strides of either sign; no bias / epilogue).  launch_gemm runs them one by one where no batched kernel applies.
    int nbatch; long batchA, batchB, batchC;
};

// ---------------------------------------------------------------------------
// GRU single-step kernels (one launch = one time step of up to 4 independent
// "problems": directions of a bi-GRU, the 4 beats of the tick RNN, ...)
// ---------------------------------------------------------------------------
struct GruFwdProb {
    int B;                                    // batch rows of this problem
    const float* h_prev; long ld_hprev;       // [B,H]
    const float* W_hh; const float* b_hh;     // [3H,H], [3H]
    // gate pre-activations from the input side, summed:
    const float* gi_dense; long ld_gi;        // [B,3H] or null
    const float* gi_table; long ld_table;     // table[idx[b*idx_stride]] rows of 3H, or null
    const long long* idx; long idx_stride;
    const float* gi_vec;                      // [3H] broadcast or null
    const float* x; long ldx; int K2;         // in-kernel input contraction x[B,K2] * W_ih[3H,K2]^T (+ b_ih), or null
    const float* W_ih; long ld_wih; const float* b_ih;
    // outputs
    float* h_new; long ld_hnew;               // [B,H]
    float* h_masked; long ld_hm;              // optional h_new * mask
    const float* mask; long ld_mask;
    float* h_copy; long ld_hc;                // optional second plain copy of h_new
    // saved for backward (all [B,H], row stride H), or null
    float* sv_r; float* sv_z; float* sv_n; float* sv_ghn; float* sv_hprev;
    // Fragment-major ("packed", ksplit.h) twins of the contraction operands.  When hpk_prev and Wpk_hh (and, with x,
    // xpk and Wpk_ih) are given for EVERY problem of a launch the contraction streams those instead of the row-major
    // ones (which must still be valid: the epilogue reads h_prev).  Requires H % 256 == 0 (and K2 % 256 == 0).
    const float* hpk_prev; const float* Wpk_hh;   // [ceil(B/16)][H/16][64][4], [3H/16][H/16][64][4]
    const float* xpk; const float* Wpk_ih;        // [ceil(B/16)][K2/16][64][4], [3H/16][K2/16][64][4]
    float* hpk_new;                               // optional packed copy of h_new (the next step's hpk_prev)
    float* hmpk_new;                              // optional packed copy of h_masked (the next layer's xpk)
};
struct GruFwdBatch { int H; int nprob; int tiles_per_prob; int rows_fastest; GruFwdProb p[4]; };

struct GruBwdProb {
    int B;
    // recurrent term: dh += dgh_next[B,3H] * W_hh  (given as W_hhT [H,3H])
    const float* dgh_next; long ld_dgh;       // null => no recurrent term
    const float* W_hhT;                       // [H,3H]
    const float* dhz_next;                    // [B,H] (dh' * z of the later step) or null
    const float* dout; long ld_dout;          // [B,H] external gradient into this step's output, or null
    const float* dout2; long ld_dout2;        // second external gradient (e.g. final-hidden grad), or null
    // pointwise part (null sv_r => only write dh to dh_out)
    const float* sv_r; const float* sv_z; const float* sv_n; const float* sv_ghn; const float* sv_hprev;
    float* dgi; long ld_dgi;                  // [B,3H]
    float* dgh; long ld_dghout;               // [B,3H]
    float* dhz;                               // [B,H]
    float* db_ih; float* db_hh;               // [3H] bias gradients, accumulated with atomics (or null)
    float* dh_out; long ld_dhout;             // [B,H] (only when no pointwise part): gradient wrt the initial hidden
    int dh_out_accumulate;
    // fragment-major twins (see GruFwdProb): used when given for every problem that has a recurrent term; H % 256 == 0
    const float* dghpk_next; const float* Wpk_hhT;   // [ceil(B/16)][3H/16][64][4], [H/16][3H/16][64][4]
    float* dghpk;                                    // optional packed copy of dgh (the next step's dghpk_next)
};
struct GruBwdBatch { int H; int nprob; int tiles_per_prob; int rows_fastest; GruBwdProb p[4]; };

// host-side launchers (defined in the .hip files)
int launch_gemm(const GemmArgs& g, hipStream_t s);
constexpr int kGemmGroupMax = 4;
// n independent products: one launch when a grouped kernel applies (gemm.hip), else one after the other
int launch_gemm_group(const GemmArgs* list, int n, hipStream_t s);
void gemm_set_force(int cfg, int split);      // cfg: -1 cost model, 0..4 tile configuration; split: 0 = 1, else forced
void gemm_set_direct(int mode);                 // 0 never, 1 cost model, 2 whenever applicable (k-major x k-major products)
int launch_gru_fwd(const GruFwdBatch& b, hipStream_t s);
int launch_gru_bwd(const GruBwdBatch& b, hipStream_t s);
// hpk / Wpk: optional fragment-major twins of h and W (used when both are given and H % 256 == 0)
int launch_logits_argmax(const float* h, long ldh, int B, int H, const float* W, const float* bias, int V, float* out,
                         long ldo, long long* samples, long sstride, hipStream_t s, const float* hpk = nullptr,
                         const float* Wpk = nullptr);
