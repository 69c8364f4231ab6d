// HBM-bound helpers of the hot path: wavefront-reduced softmax / cross-entropy /
// accuracy, KL + reparameterisation, argmax with the lowest-index tie rule,
// fused Adam over the flat arena, column sums (bias gradients), one-hot
// expansion (embedding gradients as MFMA contractions instead of contended
// atomics), transposes, dropout-mask generation.  All are grid-stride,
// coalesced along the contiguous dimension, 64-lane wave reductions via DPP
// shuffles.
#include "common.h"
#include "pointwise.h"
#include "prof.h"
#include "chain.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

// out[c*ld_out + r] = in[r*ld_in + c]
__global__ void transpose_kernel(const float* __restrict__ in, long ld_in, float* __restrict__ out, long ld_out,
                                 int rows, int cols) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? in[(long)r * ld_in + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (r < rows && c < cols) out[(long)c * ld_out + r] = tile[tx][i];
    }
}

// Row-major -> fragment-major (ksplit.h): out[row/16][k/16][lane = ((k%16)/4)*16 + row%16][k%4] = Y(row, k) where
// Y(row,k) = in[row*ld + k] (transposed = 0) or in[k*ld + row] (transposed = 1).  Rows R..ceil16(R) are zero-filled.
// One thread per 16-byte lane slot; blockIdx.y = batch entry.
__global__ void pack_frag_kernel(const float* __restrict__ in, long ld, int R, int K, float* __restrict__ out,
                                 int transposed, long in_bstride, long out_bstride) {
    const int S = K >> 4;
    const long slots = (long)((R + 15) >> 4) * S * 64;
    in += blockIdx.y * in_bstride;
    out += blockIdx.y * out_bstride;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const long blk = i >> 6;
        const int sb = (int)(blk % S), rb = (int)(blk / S);
        const int row = 16 * rb + (lane & 15), k = 16 * sb + 4 * (lane >> 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < R) {
            if (!transposed) {
                const float* q = in + (long)row * ld + k;
                v = make_float4(q[0], q[1], q[2], q[3]);
            } else {
                const float* q = in + (long)k * ld + row;
                v = make_float4(q[0], q[ld], q[2 * ld], q[3 * ld]);
            }
        }
        *reinterpret_cast<float4*>(out + 4 * i) = v;
    }
}

// Up to 8 same-shaped matrices in one launch (the recurrent weights of a module): blockIdx.y = matrix.
struct PackList { const float* in[8]; float* out[8]; };
__global__ void pack_frag_multi_kernel(PackList pl, long ld, int R, int K, int transposed) {
    const int S = K >> 4;
    const long slots = (long)((R + 15) >> 4) * S * 64;
    const float* __restrict__ in = pl.in[blockIdx.y];
    float* __restrict__ out = pl.out[blockIdx.y];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const long blk = i >> 6;
        const int sb = (int)(blk % S), rb = (int)(blk / S);
        const int row = 16 * rb + (lane & 15), k = 16 * sb + 4 * (lane >> 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < R) {
            if (!transposed) {
                const float* q = in + (long)row * ld + k;
                v = make_float4(q[0], q[1], q[2], q[3]);
            } else {
                const float* q = in + (long)k * ld + row;
                v = make_float4(q[0], q[ld], q[2 * ld], q[3 * ld]);
            }
        }
        *reinterpret_cast<float4*>(out + 4 * i) = v;
    }
}

// One wavefront per row of V logits.  loss_sum += out_scale * (lse - w[target]);
// correct += out_scale * (argmax_first(w) == target); dW = (softmax - onehot) * scale.
__global__ void ce_kernel(const float* __restrict__ W, long ld_w, int rows, int V,
                          const long long* __restrict__ tgt, float* __restrict__ dW, long ld_dw, float scale,
                          float out_scale, float* __restrict__ loss_sum, float* __restrict__ correct,
                          const float* __restrict__ scale_dev, const float* __restrict__ add_term, float add_scale,
                          float* __restrict__ fwd_out, float fwd_scale) {
    // scale_dev: dW is also multiplied by this device scalar (the upstream gradient of the mean loss, read on the device:
    // no elementwise pass over dW afterwards); add_term: *loss_sum also receives add_scale * add_term[0], once (the KL
    // term of the ELBO: loss = CE + beta/B * KL leaves this kernel complete)
    // fwd_out: receives fwd_scale * scale_dev[0] (the gradient handed on to the producer of add_term, e.g. the KL sum)
    if (scale_dev) {
        if (fwd_out && blockIdx.x == 0 && threadIdx.x == 0) *fwd_out = fwd_scale * *scale_dev;
        scale *= *scale_dev;
    }
    if (add_term && loss_sum && blockIdx.x == 0 && threadIdx.x == 0) unsafeAtomicAdd(loss_sum, add_scale * *add_term);
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    float lsum = 0.f, csum = 0.f;
    for (int row = wave; row < rows; row += nwaves) {
        const float* w = W + (long)row * ld_w;
        float m = -INFINITY;
        for (int v = lane; v < V; v += 64) m = fmaxf(m, w[v]);
        m = wave_max(m);
        float se = 0.f;
        int am = 0x7fffffff;
        for (int v = lane; v < V; v += 64) {
            const float x = w[v];
            se += expf(x - m);
            if (x == m) am = min(am, v);
        }
        se = wave_sum(se);
        am = wave_min_i(am);
        const int tg = (int)tgt[row];
        const float lse = m + logf(se);
        if (lane == 0) {
            lsum += lse - w[tg];
            csum += (am == tg) ? 1.f : 0.f;
        }
        if (dW) {
            float* d = dW + (long)row * ld_dw;
            const float inv = 1.f / se;
            for (int v = lane; v < V; v += 64) d[v] = (expf(w[v] - m) * inv - (v == tg ? 1.f : 0.f)) * scale;
        }
    }
    // lane 0 of every wave holds that wave's partials: combine the 4 waves of the block, one atomic pair per block
    __shared__ float part[2][4];
    if (lane == 0) { part[0][threadIdx.x >> 6] = lsum; part[1][threadIdx.x >> 6] = csum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (loss_sum) unsafeAtomicAdd(loss_sum, (part[0][0] + part[0][1] + part[0][2] + part[0][3]) * out_scale);
        if (correct) unsafeAtomicAdd(correct, (part[1][0] + part[1][1] + part[1][2] + part[1][3]) * out_scale);
    }
}

// z = mu + eps * exp(ls);  kl_sum += sum 0.5(sigma^2 + mu^2 - 1) - ls
__global__ void reparam_kl_kernel(const float* __restrict__ mu, const float* __restrict__ ls,
                                  const float* __restrict__ eps, float* __restrict__ z, float* __restrict__ sigma,
                                  long n, float* __restrict__ kl_sum) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float l = ls[i], m = mu[i];
        const float s = expf(l);
        if (z) z[i] = m + (eps ? eps[i] : 0.f) * s;
        if (sigma) sigma[i] = s;
        acc += 0.5f * (s * s + m * m - 1.f) - l;
    }
    if (kl_sum) {                                   // one atomic per block: a thousand of them on one address cost ~10 us
        __shared__ float part[4];
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) unsafeAtomicAdd(kl_sum, (part[0] + part[1]) + (part[2] + part[3]));
    }
}

// dmu = dz + k*mu ;  dls = dz*eps*sigma + k*(sigma^2 - 1)      (k = kscale [* *kdev]: beta / B times the upstream gradient
// of the KL term, which autograd hands over as a device scalar)
__global__ void latent_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ mu,
                                  const float* __restrict__ ls, const float* __restrict__ eps, float kscale,
                                  const float* __restrict__ kdev, float* __restrict__ dmu, float* __restrict__ dls, long n) {
    if (kdev) kscale *= *kdev;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float s = expf(ls[i]);
        const float g = dz ? dz[i] : 0.f;
        dmu[i] = g + kscale * mu[i];
        dls[i] = g * (eps ? eps[i] : 0.f) * s + kscale * (s * s - 1.f);
    }
}

// torch.optim.Adam (2.x single-tensor form): denom = sqrt(v)/sqrt(bc2) + eps; p -= lr/bc1 * m/denom
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float lr_over_bc1, float inv_sqrt_bc2, float b1, float b2,
                            float eps, float gscale, const unsigned* __restrict__ abort_word,
                            const float* __restrict__ step_flag, unsigned* __restrict__ report) {
    // A chain kernel gave up waiting for its group (chain.h): the gradients of the step are not valid, so the step leaves
    // parameters and moments as they are.  Which word decides: `step_flag` when the caller passes one -- the flags of ALL
    // ranks summed with the gradients (inet_step_flag_export + the all-reduce), so that every rank of a data-parallel job
    // takes the same decision --, else this process's own device word.  `report` (four host-mapped words owned by the caller) tells
    // the host what was decided and whether a parameter left the finite range (encoder.py:111-116, decoder.py:424-429).
    const bool skip = step_flag ? (*step_flag != 0.f)
                                : (abort_word && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u);
    if (report && blockIdx.x == 0 && threadIdx.x == 0) {
        __hip_atomic_store(report + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (skip) __hip_atomic_store(report + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // word 3: SOME rank's prologue kernels met a token outside the vocabulary (step_flag[1] = the ranks' exported token words,
        // summed with the gradients): every rank raises the reference's check_index ValueError at the same step
        if (step_flag && step_flag[1] != 0.f) __hip_atomic_store(report + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (skip) return;
    bool bad = false;
    const long n4 = n >> 2;
    f32x4* p4 = reinterpret_cast<f32x4*>(p);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    f32x4* m4 = reinterpret_cast<f32x4*>(m);
    f32x4* v4 = reinterpret_cast<f32x4*>(v);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gg[e] * gscale;
            mm[e] = b1 * mm[e] + (1.f - b1) * gr;
            vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
            const float denom = sqrtf(vv[e]) * inv_sqrt_bc2 + eps;
            pp[e] -= lr_over_bc1 * (mm[e] / denom);
            bad |= !(fabsf(pp[e]) <= 3.402823466e38f);          // NaN or +-inf
        }
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    // tail (n is a multiple of 4 for arenas; kept for generality)
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gr = g[i] * gscale;
        const float mm = b1 * m[i] + (1.f - b1) * gr;
        const float vv = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mm; v[i] = vv;
        const float pn = p[i] - lr_over_bc1 * (mm / (sqrtf(vv) * inv_sqrt_bc2 + eps));
        p[i] = pn;
        bad |= !(fabsf(pn) <= 3.402823466e38f);
    }
    if (bad && report) __hip_atomic_store(report + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// dst[0] = 1 if a chain launch of this process has timed out since the last reset, else 0: the word a data-parallel caller sums
// over ranks together with the gradients (inet_step_flag_export)
// dst[1] = 1 if a prologue kernel of this process has met a token outside the vocabulary since the last reset (the host-mapped
// word behind inet_token_status): decoder.py:36-45's ValueError, raised by all ranks together
__global__ void step_flag_export_kernel(const unsigned* abort_word, const unsigned* tok_word, float* dst) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        dst[0] = (abort_word && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1.f : 0.f;
        dst[1] = (tok_word && __hip_atomic_load(tok_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) ? 1.f : 0.f;
    }
}

// out[n] (+)= sum_m X[m*ld + n].  Block = 64 columns x 4 row-lanes; grid.y splits rows; atomics combine.
__global__ void colsum_kernel(const float* __restrict__ X, long ld, int M, int N, float* __restrict__ out) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int rows_per = (M + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    float s = 0.f;
    if (c < N)
        for (int r = r0 + rl; r < r1; r += 4) s += X[(long)r * ld + c];
    part[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < N) {
        s = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        unsafeAtomicAdd(out + c, s);
    }
}

// several column sums in one launch: blockIdx.z picks the job (blocks outside its extent leave at once)
struct ColsumList { int n; PwColsumJob j[8]; };
__global__ void colsum_multi_kernel(ColsumList L) {
    __shared__ float part[4][64];
    const PwColsumJob& J = L.j[blockIdx.z];
    const int M = J.M, N = J.N;
    if ((int)blockIdx.x * 64 >= N) return;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int rows_per = (M + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    if (r0 >= M) return;
    float s = 0.f;
    if (c < N)
        for (int r = r0 + rl; r < r1; r += 4) s += J.X[(long)r * J.ld + c];
    part[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < N) {
        s = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        unsafeAtomicAdd(J.out + c, s);
    }
}

// onehot[row, idx[(row/inner)*s_outer + (row%inner)*s_inner]] = 1 on a zeroed [rows, W] buffer
__global__ void onehot_kernel(const long long* __restrict__ idx, int inner, long s_outer, long s_inner, int rows, int W,
                              float* __restrict__ out) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
        const long long v = idx[(long)(r / inner) * s_outer + (long)(r % inner) * s_inner];
        if (v >= 0 && v < W) out[(long)r * W + v] = 1.f;
    }
}

// Gradient of a gather table: out[v][c] += sum over rows r with tok(r) == v of X[r][c]   (out [W][ncols], zeroed by the
// caller).  Row r = (r / inner, r % inner) has token idx[(r/inner)*s_outer + (r%inner)*s_inner].  The one-hot product this
// replaces read X through the MFMA path; this is one pass over X at HBM rate: a workgroup owns 256 columns x `rows_per`
// rows, a thread owns one column of the LDS table [W+1][256] (plain read-add-write: nobody else touches its slots; LDS
// float atomics measured 200 cycles per wave -- 116 us for the encoder's 75 MB; row W takes out-of-range tokens) and the
// table is flushed with one global atomic per non-zero slot.  Loads go 16 rows deep before the first add.
constexpr int kSegMaxW = 128, kSegDepth = 16;        // (W + 1) KB of LDS per workgroup: 129 KB of the CU's 160
// CW columns per workgroup, 256 / CW row groups with a table each (narrow inputs -- embedding gradients with E = 20..30
// columns -- keep all 256 lanes busy); optional per-row factor (the Dropout2d scale of an embedded sequence).
template <int CW>
__global__ __launch_bounds__(256) void token_segsum_kernel(
    const float* __restrict__ X, long ld, const long long* __restrict__ idx, int inner, long s_outer, long s_inner, int rows,
    int rows_per, int W, int ncols, float* __restrict__ out, const float* __restrict__ row_scale) {
    constexpr int RG = 256 / CW;
    extern __shared__ float tab[];                            // [RG][W + 1][CW]
    const int c = threadIdx.x % CW, rg = threadIdx.x / CW;
    const int col = blockIdx.x * CW + c;
    float* const mine = tab + rg * (W + 1) * CW + c;
    for (int v = 0; v <= W; ++v) mine[v * CW] = 0.f;
    const int r0 = blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
    const int colc = min(col, ncols - 1);
    for (int rb = r0 + rg; rb < r1; rb += RG * kSegDepth) {
        int v[kSegDepth];
        float x[kSegDepth];
#pragma unroll
        for (int k = 0; k < kSegDepth; ++k) {
            const int r = min(rb + k * RG, rows - 1);
            const long long t = idx[(long)(r / inner) * s_outer + (long)(r % inner) * s_inner];
            v[k] = rb + k * RG < r1 && t >= 0 && t < W ? (int)t : W;
            x[k] = X[(long)r * ld + colc];
            if (row_scale) x[k] *= row_scale[r];
        }
#pragma unroll
        for (int k = 0; k < kSegDepth; ++k) mine[v[k] * CW] += x[k];
    }
    if (col < ncols)
        for (int v = 0; v < W; ++v) {
            const float x = mine[v * CW];
            if (x != 0.f) unsafeAtomicAdd(out + (long)v * ncols + col, x);
        }
}

// The two small products behind a gather table of input-side gate gradients dtab [W][ndir*N3] (ld = ndir*N3), E <= 16:
//   dW[d][j][e] (ld ldw) += sum_v dtab[v][d*N3 + j] * emb[v][e]          (blocks 0 .. nblk_a-1: a thread per (d, j, e))
//   demb[v][e]           += sum_d sum_j dtab[v][d*N3 + j] * Wih[d][j][e]  (then a block per (table row v, 512 columns))
struct TableGradArgs {
    const float* dtab; int W, N3, ndir, E;
    const float* emb; long ld_emb;                            // [W][E]
    const float* emb_last; float* demb_last;                  // optional: row W-1 of emb / demb lives elsewhere (the decoder's
                                                              // start symbol x_0 behind its V embedding rows)
    const float* Wih[2]; float* dW[2]; long ldw;              // [N3][E] (row stride ldw)
    float* demb; long ld_demb;                                // [W][E], accumulated (or null)
    int nblk_a, chunks;
};
__global__ __launch_bounds__(256) void table_grad_kernel(TableGradArgs a) {
    const int ncols = a.ndir * a.N3;
    __shared__ float sh[kSegMaxW * 16];
    if ((int)blockIdx.x < a.nblk_a) {
        for (int i = threadIdx.x; i < a.W * a.E; i += 256) {
            const int v = i / a.E, e = i % a.E;
            sh[i] = (v == a.W - 1 && a.emb_last) ? a.emb_last[e] : a.emb[(long)v * a.ld_emb + e];
        }
        __syncthreads();
        const int o = blockIdx.x * 256 + threadIdx.x;         // (col, e), e fastest
        if (o >= ncols * a.E) return;
        const int col = o / a.E, e = o - col * a.E;
        float acc = 0.f;
#pragma unroll 8
        for (int v = 0; v < a.W; ++v) acc += a.dtab[(long)v * ncols + col] * sh[v * a.E + e];
        const int d = col / a.N3, j = col - d * a.N3;
        a.dW[d][(long)j * a.ldw + e] += acc;
        return;
    }
    const int bb = blockIdx.x - a.nblk_a, v = bb / a.chunks, ch = bb - v * a.chunks;
    float acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int col = ch * 512 + threadIdx.x; col < min(ncols, (ch + 1) * 512); col += 256) {
        const int d = col / a.N3, j = col - d * a.N3;
        const float t = a.dtab[(long)v * ncols + col];
        const float* wr = a.Wih[d] + (long)j * a.ldw;
#pragma unroll
        for (int e = 0; e < 16; ++e) if (e < a.E) acc[e] += t * wr[e];
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float x = acc[e];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
        if (lane == 0) sh[wv * 16 + e] = x;
    }
    __syncthreads();
    if ((int)threadIdx.x < a.E) {
        float* dst = (v == a.W - 1 && a.demb_last) ? a.demb_last : a.demb + (long)v * a.ld_demb;
        unsafeAtomicAdd(dst + threadIdx.x, sh[threadIdx.x] + sh[16 + threadIdx.x] + sh[32 + threadIdx.x] + sh[48 + threadIdx.x]);
    }
}

// pw_prologue (pointwise.h): blockIdx.y picks the job, blockIdx.x strides over it
__global__ void prologue_kernel(PwPrologue a) {
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x, step = (long)gridDim.x * blockDim.x;
    const int job = blockIdx.y;
    if (job < 4) {
        if (job >= a.ntab) return;
        const PwTableJob& t = a.tab[job];
        const long n = (long)t.rows * t.N;
        for (long i = i0; i < n; i += step) {
            const int r = (int)(i / t.N), c = (int)(i - (long)r * t.N);
            const float* e = t.emb + r * t.ld_emb;
            const float* w = t.W + c * t.ldw;
            float acc = t.bias ? t.bias[c] : 0.f;
            for (int k = 0; k < t.E; ++k) acc = fmaf(e[k], w[k], acc);
            t.out[r * t.ld_out + c] = acc;
        }
    } else if (job == 4) {
        for (long i = i0; i < a.nzero; i += step) a.zero_words[i] = 0u;
        for (long i = i0; i < a.ncopy; i += step) a.copy_dst[i] = a.copy_src[i];
    } else if (job == 5) {
        if (a.axpb_n > 0) {
            const float av = *a.axpb_a;
            for (long i = i0; i < a.axpb_n; i += step) a.axpb_y[i] = av * a.axpb_x[i * a.axpb_incx] + a.axpb_b[i];
        }
        for (long i = i0; i < a.nfill; i += step) a.fill_ptr[i] = a.fill_val;
    } else {
        if (!a.tok_src) return;
        const long n = (long)a.tok_B * a.tok_T;
        bool bad = false;
        for (long i = i0; i < n; i += step) {
            const long long v = a.tok_src[i];
            bad |= a.tok_V > 0 && (v < 0 || v >= a.tok_V);
            if (a.tok_copy) a.tok_copy[i] = v;
            if (a.tok_shift) a.tok_shift[i] = (i % a.tok_T) == 0 ? a.tok_first : a.tok_src[i - 1];
        }
        if (bad && a.tok_bad) __hip_atomic_fetch_add(a.tok_bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}


__global__ void shift_tokens_kernel(const long long* __restrict__ target, int B, int T, long long first,
                                    long long* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * T; i += gridDim.x * blockDim.x)
        out[i] = i % T == 0 ? first : target[i - 1];
}

// x *= m      or      x *= selu'(a)  (a = SELU output)
__global__ void mul_kernel(float* __restrict__ x, const float* __restrict__ m, long n, int selu_grad) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        x[i] *= selu_grad ? selu_grad_from_out(m[i]) : m[i];
}

// in [A][B][K] -> out [B][A][K]
__global__ void swap01_kernel(const float* __restrict__ in, int A, int B, int K, float* __restrict__ out) {
    const long n = (long)A * B * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const long ab = i / K;
        const int b = (int)(ab % B), a = (int)(ab / B);
        out[((long)b * A + a) * K + k] = in[i];
    }
}

// samples[b*stride] = argmax_first(W[b,:]); one wavefront per row
__global__ void argmax_kernel(const float* __restrict__ W, long ld_w, int rows, int V, long long* __restrict__ out,
                              long stride) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int row = wave; row < rows; row += nwaves) {
        const float* w = W + (long)row * ld_w;
        float m = -INFINITY;
        for (int v = lane; v < V; v += 64) m = fmaxf(m, w[v]);
        m = wave_max(m);
        int am = 0x7fffffff;
        for (int v = lane; v < V; v += 64) if (w[v] == m) am = min(am, v);
        am = wave_min_i(am);
        if (lane == 0) out[(long)row * stride] = am;
    }
}

// p[r*ld + c] = 0  (plain kernels instead of hipMemsetAsync / hipMemcpyAsync: the runtime's blit path showed 20-40 us
// bubbles around its fill/copy kernels in the step's timeline)
__global__ void zero2d_kernel(float* __restrict__ p, long ld, long rows, int cols) {
    const long n = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        p[(i / cols) * ld + (i % cols)] = 0.f;
}
__global__ void copy_words_kernel(unsigned* __restrict__ dst, const unsigned* __restrict__ src, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}

__global__ void fill_i64_kernel(long long* p, long n, long long v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

// dst[r*ld_d + c] = src[r*ld_s + c] * (mask ? (mask[r*ld_m + c] > 0) : 1)   (generic strided 2D copy)
__global__ void copy2d_kernel(float* __restrict__ dst, long ld_d, const float* __restrict__ src, long ld_s,
                              const float* __restrict__ pos, long ld_p, int rows, int cols) {
    const long n = (long)rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        float v = src[(long)r * ld_s + c];
        if (pos && !(pos[(long)r * ld_p + c] > 0.f)) v = 0.f;
        dst[(long)r * ld_d + c] = v;
    }
}

// [B,T,V] batch-first d(post-ReLU logits) -> [T,B,V] time-major d(pre-ReLU logits)
__global__ void dlogits_relayout_kernel(const float* __restrict__ dW, const float* __restrict__ Wt, int B, int T, int V,
                                        float* __restrict__ out) {
    const long n = (long)B * T * V;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int v = (int)(i % V);
        const long bt = i / V;
        const int tt = (int)(bt % T), b = (int)(bt / T);
        const float g = Wt[i] > 0.f ? dW[i] : 0.f;
        out[((long)tt * B + b) * V + v] = g;
    }
}

// out[g][b][c] = sum_{j<G} in[(g*G + j)][b][c]   (sum the ticks of each beat)
__global__ void group_sum_kernel(const float* __restrict__ in, int groups, int G, long inner, float* __restrict__ out) {
    const long n = (long)groups * inner;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long g = i / inner, r = i % inner;
        float s = 0.f;
        for (int j = 0; j < G; ++j) s += in[((long)g * G + j) * inner + r];
        out[i] = s;
    }
}

// y[i] = a * x[i*incx] + (b ? b[i] : 0)
__global__ void axpb_kernel(const float* a_ptr, const float* __restrict__ x, long incx, const float* __restrict__ b,
                            float* __restrict__ y, int n) {
    const float a = *a_ptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        y[i] = a * x[(long)i * incx] + (b ? b[i] : 0.f);
}

// Beat-RNN layer-0 input gradients: gi = b0 * w + b_ih  =>  dw[i*incw] += b0 * s[i]  (s = column sums of dgi);
// db0 = sum_rows sum_i dgi[row][i] * w[i*incw] is a ~1.5 M-term sum of both signs that nearly cancels: per-wave atomics in
// launch order moved it by up to 2e-4 of itself between runs (round 3's parity tests gave this one scalar its own bound).
// Now a fixed-order two-stage sum: kB0Parts blocks each reduce their rows in a fixed order into partial[block]
// (rowdot_partials_kernel), and one thread adds the partials in index order (here): bit-identical from run to run.
constexpr int kB0Parts = 64;
__global__ __launch_bounds__(256) void rowdot_partials_kernel(const float* __restrict__ X, long ld, int M, int N,
                                                              const float* __restrict__ w, long incw, float* __restrict__ partial) {
    __shared__ float red[256];
    const int per = (M + kB0Parts - 1) / kB0Parts, r0 = blockIdx.x * per, r1 = min(M, r0 + per);
    float acc = 0.f;
    for (int c = threadIdx.x; c < N; c += 256) {
        const float wc = w[(long)c * incw];
        float col = 0.f;
        for (int r = r0; r < r1; ++r) col += X[(long)r * ld + c];
        acc = fmaf(col, wc, acc);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void beat_input_grad_kernel(const float* __restrict__ s, const float* __restrict__ w, long incw,
                                       const float* b0, float* __restrict__ dw, float* __restrict__ db0, int n,
                                       const float* __restrict__ partial) {
    const float bb = *b0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dw[(long)i * incw] += bb * s[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < kB0Parts; ++k) t += partial[k];
        *db0 += t;                                  // the only writer of this element in a backward call
    }
}

// Counter-based dropout mask: out = keep ? 1/(1-p) : 0, keep ~ Bernoulli(1-p).
// splitmix64-style hash of (seed, stream offset + element index): reproducible,
// rank-offsettable, no state.
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void dropout_mask_kernel(float* __restrict__ out, long n, float p, uint64_t seed, uint64_t offset) {
    const float scale = 1.f / (1.f - p);
    const uint32_t thr = (uint32_t)((double)p * 4294967296.0);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const uint64_t h = mix64(mix64(seed) ^ (offset + (uint64_t)i));
        out[i] = ((uint32_t)(h >> 32) >= thr) ? scale : 0.f;
    }
}

// samples[row*stride] ~ Multinomial(softmax(W[row,:]))  (decoder.py:506-509, sampling = 'multinomial'): one wavefront per
// row, inverse-CDF draw with one counter-based uniform per row (seed, offset + row): reproducible, rank-offsettable.
__global__ void sample_multinomial_kernel(const float* __restrict__ W, long ld_w, int rows, int V,
                                          long long* __restrict__ out, long stride, uint64_t seed, uint64_t offset) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int row = wave; row < rows; row += nwaves) {
        const float* w = W + (long)row * ld_w;
        float m = -INFINITY;
        for (int v = lane; v < V; v += 64) m = fmaxf(m, w[v]);
        m = wave_max(m);
        float tot = 0.f;
        for (int v = lane; v < V; v += 64) tot += expf(w[v] - m);
        tot = wave_sum(tot);
        const uint64_t h = mix64(mix64(seed) ^ (offset + (uint64_t)row));
        const float target = (float)((uint32_t)(h >> 40)) * (1.f / 16777216.f) * tot;     // u in [0, 1) times the total mass
        float base = 0.f;
        int pick = V - 1;
        bool found = false;
        for (int v0 = 0; v0 < V && !found; v0 += 64) {
            const int v = v0 + lane;
            const float e = v < V ? expf(w[v] - m) : 0.f;
            float incl = e;                                     // inclusive prefix sum over the 64 lanes
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float up = __shfl_up(incl, o, 64);
                if (lane >= o) incl += up;
            }
            const unsigned long long hit = __ballot(v < V && base + incl > target);
            if (hit) { pick = v0 + __ffsll((long long)hit) - 1; found = true; }
            base += __shfl(incl, 63, 64);
        }
        if (lane == 0) out[(long)row * stride] = pick;
    }
}

__global__ void scale_kernel(float* __restrict__ x, long n, float a) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= a;
}

// out[r, :] = table[idx[r], :]
__global__ void embedding_fwd_kernel(const float* __restrict__ table, const long long* __restrict__ idx, long rows, int E,
                                     float* __restrict__ out, const float* __restrict__ row_scale) {
    const long n = rows * E;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / E; const int e = (int)(i % E);
        out[i] = table[idx[r] * E + e] * (row_scale ? row_scale[r] : 1.f);
    }
}
// dtable[idx[r], :] += dout[r, :]
__global__ void embedding_bwd_kernel(const float* __restrict__ dout, const long long* __restrict__ idx, long rows, int E,
                                     float* __restrict__ dtable, const float* __restrict__ row_scale) {
    const long n = rows * E;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / E; const int e = (int)(i % E);
        const float sc = row_scale ? row_scale[r] : 1.f;
        if (sc != 0.f) unsafeAtomicAdd(dtable + idx[r] * E + e, dout[i] * sc);
    }
}

// Input feed: the dataset tensors are int32 (folk_dataset.py:852-861); the model wants int64 (utils/helpers.py:17-26).
// The H2D copy moves the 4-byte form, the widening happens here.
__global__ void tokens_i32_to_i64_kernel(const int* __restrict__ src, long long* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}
// score [B, M*L] int32 -> past [B, np, L], target [B, nt, L], future [B, M-np-nt, L] int64, each contiguous
// (LatentRNNTrainer.split_score, latent_rnn_trainer.py:134-160): one pass, every token read once and written once.
__global__ void split_measures_kernel(const int* __restrict__ score, int B, int M, int L, int n_past, int n_target,
                                      long long* __restrict__ past, long long* __restrict__ target,
                                      long long* __restrict__ future) {
    const long n = (long)B * M * L;
    const int n_future = M - n_past - n_target;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int l = (int)(i % L);
        const int m = (int)((i / L) % M);
        const long b = i / ((long)L * M);
        const long long v = score[i];
        if (m < n_past) past[(b * n_past + m) * L + l] = v;
        else if (m < n_past + n_target) target[(b * n_target + (m - n_past)) * L + l] = v;
        else future[(b * n_future + (m - n_past - n_target)) * L + l] = v;
    }
}

inline int grid_for(long n, int block = 256, int cap = 2048) {
    long g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}
inline int ok() { return hipGetLastError() == hipSuccess ? 0 : -2; }

}  // namespace

int pw_transpose(const float* in, long ld_in, float* out, long ld_out, int rows, int cols, hipStream_t s) {
    dim3 grid((cols + 31) / 32, (rows + 31) / 32);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, in, ld_in, out, ld_out, rows, cols);
    return ok();
}
int pw_pack_frag(const float* in, long ld, int R, int K, float* out, int transposed, int nbatch, long in_bstride,
                 long out_bstride, hipStream_t s) {
    if (K % 16 != 0 || R <= 0 || nbatch <= 0) return -1;
    const long slots = (long)((R + 15) / 16) * (K / 16) * 64;
    dim3 grid(grid_for(slots, 256, 1024), nbatch);
    hipLaunchKernelGGL(pack_frag_kernel, grid, dim3(256), 0, s, in, ld, R, K, out, transposed, in_bstride, out_bstride);
    return ok();
}
int pw_pack_frag_multi(const float* const* ins, float* const* outs, int n, long ld, int R, int K, int transposed,
                       hipStream_t s) {
    if (K % 16 != 0 || R <= 0 || n <= 0 || n > 8) return -1;
    PackList pl{};
    for (int i = 0; i < n; ++i) { pl.in[i] = ins[i]; pl.out[i] = outs[i]; }
    const long slots = (long)((R + 15) / 16) * (K / 16) * 64;
    dim3 grid(grid_for(slots, 256, 1024), n);
    hipLaunchKernelGGL(pack_frag_multi_kernel, grid, dim3(256), 0, s, pl, ld, R, K, transposed);
    return ok();
}
int pw_cross_entropy(const float* W, long ld_w, int rows, int V, const long long* tgt, float* dW, long ld_dw,
                     float scale, float out_scale, float* loss_sum, float* correct, hipStream_t s, const float* scale_dev,
                     const float* add_term, float add_scale, float* fwd_out, float fwd_scale) {
    // blocks: every block ends with one atomic pair on the same two words, and those serialise (~20 ns each): 2048 blocks (one row
    // per wave) measured 77 us for the loss glue of the B = 256 step, 512 blocks 54 us with the forward launch at 18.7 us against
    // 6.2 for the backward launch that has no sums (profiles/r06_p_timeline_full_step.txt).  With sums: 128 blocks (12 rows per
    // wave at 6144 rows); without (the backward launch): as many as there are rows to go round
    const int g = grid_for((long)rows * 64, 256, (loss_sum || correct) ? 128 : 512);
    hipLaunchKernelGGL(ce_kernel, dim3(g), dim3(256), 0, s, W, ld_w, rows, V, tgt, dW, ld_dw, scale, out_scale, loss_sum,
                       correct, scale_dev, add_term, add_scale, fwd_out, fwd_scale);
    return ok();
}
int pw_reparam_kl(const float* mu, const float* ls, const float* eps, float* z, float* sigma, long n, float* kl_sum,
                  hipStream_t s) {
    hipLaunchKernelGGL(reparam_kl_kernel, dim3(grid_for(n, 256, kl_sum ? 64 : 512)), dim3(256), 0, s, mu, ls, eps, z, sigma, n, kl_sum);
    return ok();
}
int pw_latent_bwd(const float* dz, const float* mu, const float* ls, const float* eps, float kscale, const float* kdev,
                  float* dmu, float* dls, long n, hipStream_t s) {
    hipLaunchKernelGGL(latent_bwd_kernel, dim3(grid_for(n, 256, 512)), dim3(256), 0, s, dz, mu, ls, eps, kscale, kdev, dmu,
                       dls, n);
    return ok();
}
int pw_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, int step,
            float gscale, hipStream_t s, const float* step_flag, unsigned* report) {
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    ProfScope prof(PROF_HBM, 0.0, s, "adam", 28.0 * (double)n);      // read p,g,m,v; write p,m,v
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4 + 1, 256, 4096)), dim3(256), 0, s, p, g, m, v, n,
                       (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), b1, b2, eps, gscale,
                       (const unsigned*)chain_dev_status(), step_flag, report);
    return ok();
}
int pw_step_flag_export(float* dst, hipStream_t s) {
    hipLaunchKernelGGL(step_flag_export_kernel, dim3(1), dim3(64), 0, s, (const unsigned*)chain_dev_status(),
                       (const unsigned*)token_host_status(), dst);
    return ok();
}

// sums[0..2] += (loss, accuracy, 1) unless a chain launch of this process has timed out since the last reset: the statistics of
// steps whose results are not valid (and that the optimizer kernel skipped) stay out of the epoch means
__global__ void epoch_stats_add_kernel(float* sums, const float* loss, const float* acc, const unsigned* abort_word,
                                       const float* step_flag) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // the same decision as the optimizer kernel of the step (adam_kernel): the ranks' summed flag if the caller has one
    if (step_flag ? (*step_flag != 0.f)
                  : (abort_word && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) return;
    sums[0] += loss[0];
    if (acc) sums[1] += acc[0];
    sums[2] += 1.f;
}
int pw_epoch_stats_add(float* sums, const float* loss, const float* acc, hipStream_t s, const float* step_flag) {
    hipLaunchKernelGGL(epoch_stats_add_kernel, dim3(1), dim3(64), 0, s, sums, loss, acc, (const unsigned*)chain_dev_status(),
                       step_flag);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int pw_colsum(const float* X, long ld, int M, int N, float* out, hipStream_t s) {
    int gy = (M + 255) / 256;
    if (gy > 64) gy = 64;
    if (gy < 1) gy = 1;
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, gy), dim3(256), 0, s, X, ld, M, N, out);
    return ok();
}
int pw_colsum_multi(const PwColsumJob* jobs, int n, hipStream_t s) {
    if (n <= 0) return 0;
    if (n > 8) return -1;
    ColsumList L{};
    L.n = n;
    int maxN = 0, maxM = 0;
    for (int i = 0; i < n; ++i) {
        L.j[i] = jobs[i];
        if (jobs[i].N > maxN) maxN = jobs[i].N;
        if (jobs[i].M > maxM) maxM = jobs[i].M;
    }
    int gy = (maxM + 255) / 256;
    if (gy > 64) gy = 64;
    if (gy < 1) gy = 1;
    hipLaunchKernelGGL(colsum_multi_kernel, dim3((maxN + 63) / 64, gy, n), dim3(256), 0, s, L);
    return ok();
}
int pw_onehot(const long long* idx, int inner, long s_outer, long s_inner, int rows, int W, float* out, int zero_first,
              hipStream_t s) {
    if (zero_first && pw_zero(out, (long)rows * W, s) != 0) return -2;
    hipLaunchKernelGGL(onehot_kernel, dim3(grid_for(rows)), dim3(256), 0, s, idx, inner, s_outer, s_inner, rows, W, out);
    return ok();
}
int pw_token_segsum(const float* X, long ld, const long long* idx, int inner, long s_outer, long s_inner, int rows, int W,
                    int ncols, float* out, hipStream_t s, const float* row_scale, int zero_first) {
    if (W > kSegMaxW) return -1;
    if (zero_first && pw_zero(out, (long)W * ncols, s) != 0) return -2;
    const int cw = ncols <= 32 ? 32 : (ncols <= 64 ? 64 : 256), rg = 256 / cw;
    const int col_blocks = (ncols + cw - 1) / cw;
    int row_blocks = 1;                                       // >= 384 workgroups (1.5 per CU, 3 fit), at least 64 rows per row group
    while (col_blocks * row_blocks < 384 && rows / (row_blocks * 2) >= 64 * rg) row_blocks *= 2;
    const int rows_per = (rows + row_blocks - 1) / row_blocks;
    const dim3 grid(col_blocks, row_blocks);
    const size_t lds = (size_t)(W + 1) * 256 * sizeof(float);
#define DISPATCH_SEG(C)                                                                                                        \
    do {                                                                                                                   \
        static bool attr = false;                              /* more than 64 KB of dynamic LDS needs the opt-in */        \
        if (!attr) {                                                                                                       \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&token_segsum_kernel<C>),                               \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                             \
            attr = true;                                                                                                   \
        }                                                                                                                  \
        hipLaunchKernelGGL(token_segsum_kernel<C>, grid, dim3(256), lds, s, X, ld, idx, inner, s_outer, s_inner, rows,     \
                           rows_per, W, ncols, out, row_scale);                                                            \
    } while (0)
    if (cw == 32) DISPATCH_SEG(32); else if (cw == 64) DISPATCH_SEG(64); else DISPATCH_SEG(256);
#undef DISPATCH_SEG
    return ok();
}
int pw_table_grad(const float* dtab, int W, int N3, int ndir, int E, const float* emb, long ld_emb, const float* const* Wih,
                  float* const* dW, long ldw, float* demb, long ld_demb, hipStream_t s, const float* emb_last,
                  float* demb_last) {
    if (E > 16 || W > kSegMaxW || ndir < 1 || ndir > 2) return -1;
    TableGradArgs a{};
    a.emb_last = emb_last; a.demb_last = demb_last;
    a.dtab = dtab; a.W = W; a.N3 = N3; a.ndir = ndir; a.E = E; a.emb = emb; a.ld_emb = ld_emb; a.ldw = ldw;
    for (int d = 0; d < ndir; ++d) { a.Wih[d] = Wih[d]; a.dW[d] = dW[d]; }
    a.demb = demb; a.ld_demb = ld_demb;
    a.nblk_a = (ndir * N3 * E + 255) / 256; a.chunks = (ndir * N3 + 511) / 512;
    hipLaunchKernelGGL(table_grad_kernel, dim3(a.nblk_a + (demb ? W * a.chunks : 0)), dim3(256), 0, s, a);
    return ok();
}
int pw_shift_tokens(const long long* target, int B, int T, long long first, long long* out, hipStream_t s) {
    hipLaunchKernelGGL(shift_tokens_kernel, dim3(grid_for((long)B * T)), dim3(256), 0, s, target, B, T, first, out);
    return ok();
}
int pw_mul(float* x, const float* m, long n, int selu_grad, hipStream_t s) {
    hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, m, n, selu_grad);
    return ok();
}
int pw_swap01(const float* in, int A, int B, int K, float* out, hipStream_t s) {
    hipLaunchKernelGGL(swap01_kernel, dim3(grid_for((long)A * B * K)), dim3(256), 0, s, in, A, B, K, out);
    return ok();
}
int pw_argmax(const float* W, long ld_w, int rows, int V, long long* out, long stride, hipStream_t s) {
    hipLaunchKernelGGL(argmax_kernel, dim3(grid_for((long)rows * 64, 256, 1024)), dim3(256), 0, s, W, ld_w, rows, V, out, stride);
    return ok();
}
int pw_zero2d(float* p, long ld, long rows, int cols, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(zero2d_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, s, p, ld, rows, cols);
    return ok();
}
int pw_zero(float* p, long n, hipStream_t s) { return pw_zero2d(p, n, 1, (int)n, s); }
int pw_copy_bytes(void* dst, const void* src, long nbytes, hipStream_t s) {
    if (nbytes <= 0) return 0;
    if (nbytes % 4 != 0) return -1;
    hipLaunchKernelGGL(copy_words_kernel, dim3(grid_for(nbytes / 4)), dim3(256), 0, s, (unsigned*)dst, (const unsigned*)src,
                       nbytes / 4);
    return ok();
}
int pw_fill_i64(long long* p, long n, long long v, hipStream_t s) {
    hipLaunchKernelGGL(fill_i64_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, n, v);
    return ok();
}
int pw_copy2d(float* dst, long ld_d, const float* src, long ld_s, const float* pos, long ld_p, int rows, int cols,
              hipStream_t s) {
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for((long)rows * cols)), dim3(256), 0, s, dst, ld_d, src, ld_s, pos, ld_p, rows, cols);
    return ok();
}
int pw_dlogits_relayout(const float* dW, const float* Wt, int B, int T, int V, float* out, hipStream_t s) {
    hipLaunchKernelGGL(dlogits_relayout_kernel, dim3(grid_for((long)B * T * V)), dim3(256), 0, s, dW, Wt, B, T, V, out);
    return ok();
}
int pw_group_sum(const float* in, int groups, int G, long inner, float* out, hipStream_t s) {
    hipLaunchKernelGGL(group_sum_kernel, dim3(grid_for((long)groups * inner)), dim3(256), 0, s, in, groups, G, inner, out);
    return ok();
}
int pw_axpb(const float* a, const float* x, long incx, const float* b, float* y, int n, hipStream_t s) {
    hipLaunchKernelGGL(axpb_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, x, incx, b, y, n);
    return ok();
}
int pw_beat_input_grad(const float* sv, const float* w, long incw, const float* b0, float* dw, float* db0, int n,
                       const float* X, long ld, int M, float* partial, hipStream_t s) {
    hipLaunchKernelGGL(rowdot_partials_kernel, dim3(kB0Parts), dim3(256), 0, s, X, ld, M, n, w, incw, partial);
    hipLaunchKernelGGL(beat_input_grad_kernel, dim3(grid_for(n)), dim3(256), 0, s, sv, w, incw, b0, dw, db0, n,
                       (const float*)partial);
    return ok();
}
int pw_dropout_mask(float* out, long n, float p, uint64_t seed, uint64_t offset, hipStream_t s) {
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, s, out, n, p, seed, offset);
    return ok();
}
int pw_sample_multinomial(const float* W, long ld_w, int rows, int V, long long* out, long stride, uint64_t seed,
                          uint64_t offset, hipStream_t s) {
    hipLaunchKernelGGL(sample_multinomial_kernel, dim3(grid_for((long)rows * 64, 256, 1024)), dim3(256), 0, s, W, ld_w, rows, V,
                       out, stride, seed, offset);
    return ok();
}
int pw_scale(float* x, long n, float a, hipStream_t s) {
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, n, a);
    return ok();
}

int pw_embedding_fwd(const float* table, const long long* idx, long rows, int E, float* out, const float* row_scale,
                     hipStream_t s) {
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(grid_for(rows * E)), dim3(256), 0, s, table, idx, rows, E, out, row_scale);
    return ok();
}
int pw_embedding_bwd(const float* dout, const long long* idx, long rows, int E, float* dtable, const float* row_scale,
                     hipStream_t s, int num_embeddings) {
    // a small table: thousands of rows land on a handful of table rows and the per-element atomics serialise (100-160 us for
    // the metadata embeddings of AnticipationRNN); the segment-sum kernel adds them up in LDS first
    if (num_embeddings > 0 && num_embeddings <= kSegMaxW && rows >= 1024 && rows < (1L << 31))
        return pw_token_segsum(dout, E, idx, (int)rows, 0, 1, (int)rows, num_embeddings, E, dtable, s, row_scale, 0);
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(grid_for(rows * E)), dim3(256), 0, s, dout, idx, rows, E, dtable, row_scale);
    return ok();
}
int pw_tokens_i32_to_i64(const int* src, long long* dst, long n, hipStream_t s) {
    hipLaunchKernelGGL(tokens_i32_to_i64_kernel, dim3(grid_for(n)), dim3(256), 0, s, src, dst, n);
    return ok();
}
int pw_split_measures(const int* score, int B, int M, int L, int n_past, int n_target, long long* past,
                      long long* target, long long* future, hipStream_t s) {
    hipLaunchKernelGGL(split_measures_kernel, dim3(grid_for((long)B * M * L)), dim3(256), 0, s, score, B, M, L, n_past,
                       n_target, past, target, future);
    return ok();
}
int pw_prologue(const PwPrologue& p, hipStream_t s) {
    PwPrologue q = p;
    q.tok_bad = q.tok_V > 0 ? token_host_status() : nullptr;
    // grid.x: one pass over the largest job -- the tick GRU's gather table, (V + 1) x 3H outputs of E dependent multiply-adds each, every
    // thread walking its own row of W_ih -- instead of six (48 workgroups per job until round 6: the launch took 11.8 us, a tenth of a
    // one-measure decode call); every job strides over the grid, so any width is correct
    long widest = 0;
    for (int j = 0; j < q.ntab && j < 4; ++j) widest = widest > (long)q.tab[j].rows * q.tab[j].N ? widest : (long)q.tab[j].rows * q.tab[j].N;
    widest = widest > q.nzero / 4 ? widest : q.nzero / 4;
    const int gx = (int)((widest + 255) / 256 < 48 ? 48 : ((widest + 255) / 256 > 512 ? 512 : (widest + 255) / 256));
    hipLaunchKernelGGL(prologue_kernel, dim3(gx, 7), dim3(256), 0, s, q);
    return ok();
}
