// fp32 GEMM on the f32-input matrix cores of gfx950 (v_mfma_f32_32x32x2_f32:
// exact f32 products, f32 accumulate -- the only MFMA class that keeps the
// reference's fp32 numerics; there is no TF32-like path on CDNA4).
//
//   C[M,N] (op)= epi( sum_k A(m,k) * B(n,k) + bias[n] )
//
// Used for every batched ("all time steps at once") contraction of the hot path:
// GRU input projections, output projections, the Linear+SELU heads, and in the
// backward pass the dgrad (B given k-major) and wgrad (A and B given k-major,
// split-K with hardware f32 atomics) products.  The recurrent, latency-bound
// steps live in gru.hip.
//
// Tiling: 256 threads = 4 wavefronts in a 2x2 arrangement; block tile
// BMxBNx32, each wave owns (BM/2)x(BN/2) as 32x32 MFMA tiles.  Operands are
// staged through LDS (double buffered, next tile prefetched into registers
// while the current one feeds the MFMAs).  An operand whose reduction index is
// contiguous in memory is kept [row][k] with a row pitch of 36 words (4*odd:
// ds_read_b128 conflict-free, one read feeds 4 MFMA k-steps); an operand that
// is contiguous along its row index is kept [k][row] and read with
// conflict-free ds_read_b32.  The k index owned by lane-half h in MFMA step i
// of a 32-chunk is 8*(i/4) + 4*h + i%4 for both operands.
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include "prof.h"
#include "side.h"
#include "pointwise.h"

namespace {

template <int R, bool KM>
struct OperandTile {
    static constexpr int NV = R / 32;                       // float4 per thread per 32-deep chunk
    static constexpr int PITCH = KM ? (R + 4) : 36;         // words
    static constexpr int WORDS = KM ? 32 * PITCH : R * PITCH;

    // global -> registers.  rows_total: valid extent of the row index, kend: valid extent of k.
    // FAST = this 32-deep chunk lies inside [0,kend) and (k-major operands) the tile lies inside the row range:
    // every load is then an unconditional 16-byte load.  (Per-lane "load or zero" branches make hipcc wrap each
    // load in an exec-mask branch and serialise the round trips -- measured 2x on this kernel.)  Rows past the
    // end are CLAMPED, not zeroed: they only feed output rows/columns the epilogue never stores.
    template <bool FAST>
    __device__ static __forceinline__ void load(f32x4 (&v)[NV], const float* __restrict__ P, long ld,
                                                int row0, int rows_total, int k0, int kend, int t) {
        if (!KM) {
            const int c4 = t & 7;
            const int k = k0 + c4 * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int row = min(row0 + (t >> 3) + 32 * i, rows_total - 1);
                const float* p = P + (long)row * ld + k;
                if (FAST) {
                    v[i] = ld4u(p);
                } else {
                    f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < kend) x[e] = p[e];
                    v[i] = x;
                }
            }
        } else {
            constexpr int VPR = R / 4;                       // float4 per k-row
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int e = t + 256 * i;
                const int k = k0 + e / VPR;
                const int row = row0 + (e % VPR) * 4;
                if (FAST) {
                    v[i] = ld4u(P + (long)k * ld + row);
                } else {
                    f32x4 x = {0.f, 0.f, 0.f, 0.f};
                    if (k < kend) {
                        const float* p = P + (long)k * ld + row;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (row + q < rows_total) x[q] = p[q];
                    }
                    v[i] = x;
                }
            }
        }
    }

    // registers -> LDS
    __device__ static __forceinline__ void store(const f32x4 (&v)[NV], float* lds, int t) {
        if (!KM) {
            const int c4 = t & 7;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int r = (t >> 3) + 32 * i;
                *reinterpret_cast<f32x4*>(lds + r * PITCH + c4 * 4) = v[i];
            }
        } else {
            constexpr int VPR = R / 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int e = t + 256 * i;
                *reinterpret_cast<f32x4*>(lds + (e / VPR) * PITCH + (e % VPR) * 4) = v[i];
            }
        }
    }

    // LDS -> the 4 fragment values (MFMA steps 4q..4q+3) of the 32-row tile starting at `row`
    __device__ static __forceinline__ f32x4 frag(const float* lds, int row, int q, int lane) {
        const int h = lane >> 5, l31 = lane & 31;
        if (!KM) {
            return *reinterpret_cast<const f32x4*>(lds + (row + l31) * PITCH + 8 * q + 4 * h);
        } else {
            f32x4 r;
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = lds[(8 * q + 4 * h + j) * PITCH + row + l31];
            return r;
        }
    }
};

__device__ __forceinline__ float apply_epi(float v, int epi, float aux) {
    switch (epi) {
        case EPI_SELU: return selu_f(v);
        case EPI_RELU: return v > 0.f ? v : 0.f;
        case EPI_MUL_SELU_GRAD: return v * selu_grad_from_out(aux);
        case EPI_MUL_AUX: return v * aux;
        case EPI_MUL_POS: return aux > 0.f ? v : 0.f;
        default: return v;
    }
}

// Products with a handful of rows (M <= 8: the beat-level projections and the z -> hidden linear of a b = 1 / b = 2 decode call,
// decoder.py:412-453): a tile kernel spends 9-15 us on them (one 64-row tile per workgroup, a k loop for rows that are not there).
// Here a WAVE owns one output column n: its lanes stride over k (16-byte loads of the weight row, coalesced), every lane keeps M
// partial sums, a butterfly sums the wave.  Both operands k-contiguous; all epilogues; store or accumulate.
template <int M>
__device__ __forceinline__ void gemv_rows_body(const GemmArgs& g, int n) {
    const int lane = threadIdx.x & 63;
    if (n >= g.N) return;
    const float* wrow = g.B + (long)n * g.ldb;
    float acc[M];
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = 0.f;
    const int K4 = g.K & ~3;
    for (int k = lane * 4; k < K4; k += 256) {
        const f32x4 w = ld4u(wrow + k);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const f32x4 a = ld4u(g.A + (long)m * g.lda + k);
            acc[m] += a[0] * w[0] + a[1] * w[1] + a[2] * w[2] + a[3] * w[3];
        }
    }
    for (int k = K4 + lane; k < g.K; k += 64) {
        const float w = wrow[k];
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m] += g.A[(long)m * g.lda + k] * w;
    }
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[m] += __shfl_xor(acc[m], off, 64);
    if (lane < M) {
        float v = 0.f;
#pragma unroll
        for (int m = 0; m < M; ++m) v = lane == m ? acc[m] : v;
        if (g.bias) v += g.bias[n];
        if (g.epi != EPI_NONE) v = apply_epi(v, g.epi, g.aux ? g.aux[(long)lane * g.ldaux + n] : 0.f);
        float* dst = g.C + (long)lane * g.ldc + n;
        *dst = g.acc == ACC_ADD ? *dst + v : v;
    }
}
template <int M>
__global__ __launch_bounds__(256) void gemv_rows_kernel(GemmArgs g) { gemv_rows_body<M>(g, blockIdx.x * 4 + (threadIdx.x >> 6)); }

template <int TM, int TN, bool AKM, bool BKM>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    using TA = OperandTile<BM, AKM>;
    using TB = OperandTile<BN, BKM>;
    constexpr int STAGE = TA::WORDS + TB::WORDS;
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nchunks = (kend - kbeg + 31) / 32;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[TA::NV], rb[TB::NV];
    const bool a_in = !AKM || (m0 + BM <= g.M);     // block-uniform: k-major tiles need all rows in range
    const bool b_in = !BKM || (n0 + BN <= g.N);
    auto gload = [&](int c) {
        const int k0 = kbeg + c * 32;
        const bool kfull = k0 + 32 <= kend;
        if (kfull && a_in) TA::template load<true>(ra, g.A, g.lda, m0, g.M, k0, kend, t);
        else TA::template load<false>(ra, g.A, g.lda, m0, g.M, k0, kend, t);
        if (kfull && b_in) TB::template load<true>(rb, g.B, g.ldb, n0, g.N, k0, kend, t);
        else TB::template load<false>(rb, g.B, g.ldb, n0, g.N, k0, kend, t);
    };
    if (nchunks > 0) {
        gload(0);
        TA::store(ra, lds, t);
        TB::store(rb, lds + TA::WORDS, t);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const float* As = lds + (c & 1) * STAGE;
        const float* Bs = As + TA::WORDS;
        const bool more = c + 1 < nchunks;
        if (more) gload(c + 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = TA::frag(As, wr * (BM / 2) + i * 32, q, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = TB::frag(Bs, wc * (BN / 2) + j * 32, q, lane);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
        }
        if (more) {
            float* nxt = lds + ((c + 1) & 1) * STAGE;
            TA::store(ra, nxt, t);
            TB::store(rb, nxt + TA::WORDS, t);
        }
        __syncthreads();
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int h = lane >> 5, l31 = lane & 31;
    const bool add_bias = g.bias != nullptr && blockIdx.z == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wc * (BN / 2) + j * 32 + l31;
            if (col >= g.N) continue;
            const float bv = add_bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= g.M) continue;
                float v = acc[i][j][r] + bv;
                if (g.epi != EPI_NONE) {
                    const float a = g.aux ? g.aux[(long)row * g.ldaux + col] : 0.f;
                    v = apply_epi(v, g.epi, a);
                }
                float* cp = g.C + (long)row * g.ldc + col;
                if (g.acc == ACC_STORE) *cp = v;
                else if (g.acc == ACC_ADD) *cp += v;
                else unsafeAtomicAdd(cp, v);
            }
        }
}

// ---------------------------------------------------------------------------
// Weight-gradient products (A and B both k-major: C[m][n] = sum_k A[k][m] * B[k][n], K = T*B rows) without LDS.
//
// A k-major operand already has the MFMA register image in memory: lane (c = lane % 32, h = lane / 32) of a
// v_mfma_f32_32x32x2_f32 step holds X[k + h][r0 + c], i.e. a half-wave reads one aligned 128-byte line.  Each wave
// therefore feeds its MFMAs straight from global memory, kept D k steps (~3,000 cycles) ahead in registers: no LDS, no
// barriers, one wave per SIMD that never waits for its neighbours.  The texture addresser spends 16 cycles on every
// wave-wide load whatever its width, and five dword loads per 6-MFMA step from four waves would keep it 83 % busy
// (measured: the loop then runs at 76 % of the MFMA rate), so the rows of the A strip are dealt to lanes TA at a time
// (tile i of the strip holds row TA*c + i): one 8/12-byte load feeds all TA tiles and the addresser is 50 % busy.  B stays
// in natural order, which keeps every epilogue store / atomic a contiguous 128 bytes per half-wave.  The 2 x 2 waves of a workgroup share their A / B strips through the CU's L1.
// Buffer loads past the split's last k row return 0 (the resource is sized to the split), so there is no K tail code.
// Workgroups are renumbered so that one XCD works on one k range: its L2 then streams every A / B row once.
// ---------------------------------------------------------------------------
template <int TA, int TB>
__global__ __launch_bounds__(256) void gemm_tn_direct_kernel(GemmArgs g, int tiles_n, int tiles) {
    constexpr int D = 8;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int h = lane >> 5, l31 = lane & 31;
    int v = blockIdx.x;
    if ((gridDim.x & 7) == 0) v = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int comb = v / tiles, tile = v - comb * tiles;        // comb = (problem, k range): one per XCD at 8 of them
    int split = comb;
    if (g.nbatch > 1) {
        const int splits = gridDim.x / (tiles * g.nbatch), prob = comb / splits;
        split = comb - prob * splits;
        g.A += prob * g.batchA; g.B += prob * g.batchB; g.C += prob * g.batchC;
    }
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * (64 * TA) + wr * (32 * TA), n0 = tn * (64 * TB) + wc * (32 * TB);
    const int kbeg = split * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int lda = (int)g.lda, ldb = (int)g.ldb;

    const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.A) + (long)kbeg * lda, 0, (kend - kbeg) * lda * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.B) + (long)kbeg * ldb, 0, (kend - kbeg) * ldb * 4, 0x00020000);
    int oa[D], ob[D];                                   // byte offsets of this lane's elements in k step s of the block
#pragma unroll
    for (int s = 0; s < D; ++s) {
        oa[s] = ((2 * s + h) * lda + m0 + TA * l31) * 4;
        ob[s] = ((2 * s + h) * ldb + n0 + l31) * 4;
    }
    const int adv_a = 2 * D * lda * 4, adv_b = 2 * D * ldb * 4;

    f32x16 acc[TA][TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float fa[D][TA], fb[D][TB];
    auto fetch = [&](int s) {
        if constexpr (TA == 3) {
            typedef float f32x3 __attribute__((ext_vector_type(3)));
            const f32x3 x = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(ra_rsrc, oa[s], 0, 0));
            fa[s][0] = x[0]; fa[s][1] = x[1]; fa[s][2] = x[2];
        } else {
            static_assert(TA == 2, "row strip");
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 x = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ra_rsrc, oa[s], 0, 0));
            fa[s][0] = x[0]; fa[s][1] = x[1];
        }
#pragma unroll
        for (int j = 0; j < TB; ++j)
            fb[s][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb_rsrc, ob[s] + 128 * j, 0, 0));
        oa[s] += adv_a;
        ob[s] += adv_b;
    };
#pragma unroll
    for (int s = 0; s < D; ++s) {
        fetch(s);
        __builtin_amdgcn_sched_barrier(0);              // same issue order as in the loop: the loop-head wait count is
    }                                                   // the minimum over both ways in
    const int nblocks = (kend - kbeg + 2 * D - 1) / (2 * D);
    for (int b = 0; b < nblocks; ++b) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
#pragma unroll
            for (int i = 0; i < TA; ++i)
#pragma unroll
                for (int j = 0; j < TB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][i], fb[s][j], acc[i][j], 0, 0, 0);
            fetch(s);                                   // k step s of the next block (reads 0 past the split)
            __builtin_amdgcn_sched_barrier(0);          // or hipcc gathers the block's loads behind its last MFMA
        }
    }

    const bool add_bias = g.bias != nullptr && split == 0;
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int col = n0 + j * 32 + l31;
            const float bv = add_bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + i + TA * ((r & 3) + 8 * (r >> 2) + 4 * h);
                float val = acc[i][j][r] + bv;
                if (g.epi != EPI_NONE) {
                    const float a = g.aux ? g.aux[(long)row * g.ldaux + col] : 0.f;
                    val = apply_epi(val, g.epi, a);
                }
                float* cp = g.C + (long)row * g.ldc + col;
                if (g.acc == ACC_STORE) *cp = val;
                else if (g.acc == ACC_ADD) *cp += val;
                else unsafeAtomicAdd(cp, val);
            }
        }
}

// ---------------------------------------------------------------------------
// The same idea for the forward / data-gradient products, whose A operand (activations, [M][K]) is k-contiguous:
// v_mfma_f32_16x16x4_f32 with lane (c = lane % 16, q = lane / 16).  A k-contiguous operand is read with one 16-byte
// load per (16-row tile, 16-deep k group): lane (c, q) takes row c, k = 4q..4q+3 of the group, element e feeding the
// group's MFMA e as k slot q (16 rows x 64 bytes per instruction).  B is read to match: k-contiguous the same way,
// k-major (data gradient: W as stored) as TB consecutive columns of row k = 4q + e.  Columns are dealt to lanes TB at
// a time (tile j of the strip holds column TB*c + j), so a lane's accumulators for one output row are TB adjacent
// columns and the epilogue stores 8 / 16 bytes per lane.  D = 4 groups (4 x 16 k) are kept in flight; K % 64 == 0.
// ---------------------------------------------------------------------------
template <int NV>
struct FVec { float v[NV]; };

template <int TB>
__device__ __forceinline__ FVec<TB> ld_cols(__amdgpu_buffer_rsrc_t r, int off) {
    FVec<TB> o;
    if constexpr (TB == 4) {
        const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
        o.v[0] = x[0]; o.v[1] = x[1]; o.v[2] = x[2]; o.v[3] = x[3];
    } else {
        static_assert(TB == 2 || TB == 6, "strip width");
#pragma unroll
        for (int p = 0; p < TB / 2; ++p) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 x = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off + 8 * p, 0, 0));
            o.v[2 * p] = x[0]; o.v[2 * p + 1] = x[1];
        }
    }
    return o;
}

template <int TA, int TB, bool BKM>
__global__ __launch_bounds__(256) void gemm_kc_direct_kernel(GemmArgs g, int tiles_n) {
    constexpr int D = 4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int c = lane & 15, q = lane >> 4;
    int v = blockIdx.x;
    if ((gridDim.x & 7) == 0) v = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int tm = v / tiles_n, tn = v - tm * tiles_n;
    const int m0 = tm * (32 * TA) + wr * (16 * TA), n0 = tn * (32 * TB) + wc * (16 * TB);
    const int lda = (int)g.lda, ldb = (int)g.ldb;
    const __amdgpu_buffer_rsrc_t ra_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, g.M * lda * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, (BKM ? g.K : g.N) * ldb * 4, 0x00020000);

    int oa[TA];                                          // this lane's 16 bytes of k group 0, row tile i
#pragma unroll
    for (int i = 0; i < TA; ++i) oa[i] = ((m0 + 16 * i + c) * lda + 4 * q) * 4;
    constexpr int NOB = BKM ? 4 : TB;
    int ob[NOB];                                         // k-major: row 4q + e, columns n0 + TB c ..; else row tile j
#pragma unroll
    for (int x = 0; x < NOB; ++x)
        ob[x] = BKM ? ((4 * q + x) * ldb + n0 + TB * c) * 4 : ((n0 + TB * c + x) * ldb + 4 * q) * 4;
    const int adv_b = BKM ? 16 * ldb * 4 : 64;           // one k group further

    f32x4 acc[TA][TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 fa[D][TA];
    FVec<4> fbk[D][BKM ? 1 : TB];                        // k-contiguous B: [row tile j] -> elements e
    FVec<TB> fbm[D][BKM ? 4 : 1];                        // k-major B:      [e] -> columns j
    auto fetch = [&](int s) {
#pragma unroll
        for (int i = 0; i < TA; ++i) {
            fa[s][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, oa[i], 0, 0));
            oa[i] += 64;
        }
#pragma unroll
        for (int x = 0; x < NOB; ++x) {
            if constexpr (BKM) fbm[s][x] = ld_cols<TB>(rb_rsrc, ob[x]);
            else fbk[s][x] = ld_cols<4>(rb_rsrc, ob[x]);
            ob[x] += adv_b;
        }
    };
#pragma unroll
    for (int s = 0; s < D; ++s) {
        fetch(s);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int nblocks = g.K / (16 * D);
    for (int b = 0; b < nblocks; ++b) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TA; ++i)
#pragma unroll
                    for (int j = 0; j < TB; ++j) {
                        float bv;
                        if constexpr (BKM) bv = fbm[s][e].v[j];
                        else bv = fbk[s][j].v[e];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][i][e], bv, acc[i][j], 0, 0, 0);
                    }
            fetch(s);                                    // same group of the next block; past K the values are never used
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // C/D map of the 16x16 MFMA: row = 4q + r, column slot c  ->  global column n0 + TB c + j
    const int col0 = n0 + TB * c;
    float bias[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) bias[j] = g.bias ? g.bias[col0 + j] : 0.f;
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 16 * i + 4 * q + r;
            float* cp = g.C + (long)row * g.ldc + col0;
            FVec<TB> o;
#pragma unroll
            for (int j = 0; j < TB; ++j) o.v[j] = acc[i][j][r] + bias[j];
            if (g.epi != EPI_NONE) {
                const float* ap = g.aux ? g.aux + (long)row * g.ldaux + col0 : nullptr;
#pragma unroll
                for (int j = 0; j < TB; ++j) o.v[j] = apply_epi(o.v[j], g.epi, ap ? ap[j] : 0.f);
            }
            if (g.acc == ACC_ADD) {
#pragma unroll
                for (int j = 0; j < TB; ++j) o.v[j] += cp[j];
            }
            struct __attribute__((packed, aligned(4))) Out { float v[TB]; };
            Out st;
#pragma unroll
            for (int j = 0; j < TB; ++j) st.v[j] = o.v[j];
            *reinterpret_cast<Out*>(cp) = st;
        }
}

// ---------------------------------------------------------------------------
// Medium and small products (M or N of a few hundred, K of 256..2048: the Linear heads, the decoder's batched input
// projections and their gradients): too few output tiles to give every wave its own, so the four waves of a workgroup
// split K instead.  Wave w runs the 16-deep k groups w, w+4, ... of the WHOLE workgroup tile through the same
// register-fed v_mfma_f32_16x16x4_f32 loop as above (no sharing between waves, no barriers in the loop), the four partial
// tiles meet once in LDS, and every wave finishes a quarter of the tile with the full epilogue: bias, non-linearity, aux
// operand, store or accumulate -- one launch where the LDS-tiled path needs a zero-fill, a split-K launch with f32
// atomics and (for a non-linear epilogue) a third pass.  Either operand may be k-contiguous (16-byte loads of one row) or
// k-major (the strip's rows dealt to lanes T at a time, one 4T-byte load per k row).
// ---------------------------------------------------------------------------
template <int T, bool KM>
struct KsOperand {
    // one 16-deep k group of a 16*T-row strip, as the values this lane feeds to MFMA e (k slot q): v[e][t]
    float v[4][T];
    int off[KM ? 4 : T];
    int adv;
    __device__ __forceinline__ void init(int r0, int ld, int c, int q, int g0) {
        if constexpr (KM) {
#pragma unroll
            for (int e = 0; e < 4; ++e) off[e] = ((16 * g0 + 4 * q + e) * ld + r0 + T * c) * 4;
            adv = 64 * ld * 4;                          // 4 groups (one per wave) further
        } else {
#pragma unroll
            for (int t = 0; t < T; ++t) off[t] = ((r0 + T * c + t) * ld + 16 * g0 + 4 * q) * 4;
            adv = 64 * 4;
        }
    }
    __device__ __forceinline__ void fetch(__amdgpu_buffer_rsrc_t r) {
        if constexpr (KM) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const FVec<T> x = ld_cols<T>(r, off[e]);
#pragma unroll
                for (int t = 0; t < T; ++t) v[e][t] = x.v[t];
                off[e] += adv;
            }
        } else {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const FVec<4> x = ld_cols<4>(r, off[t]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e][t] = x.v[e];
                off[t] += adv;
            }
        }
    }
};

// v = this workgroup's index among the tiles * splits workgroups of the product `g`
template <int TA, int TB, bool AKM, bool BKM>
__device__ __forceinline__ void gemm_ks_body(const GemmArgs& g, int v, int tiles_n, int tiles) {
    constexpr int D = 4, NR = TA * TB * 4;
    constexpr int RR = NR <= 64 ? NR : NR / 2;           // registers per reduction round (64 KB of LDS at most)
    constexpr int QR = RR / 4;
    static_assert(NR % RR == 0 && RR % 4 == 0, "reduction rounds");
    __shared__ float red[RR * 4 * 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    // a k-major x k-major product may also be split over the grid (k range `split`; one XCD then works on one range)
    const int split = v / tiles, tile = v - split * tiles;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * (16 * TA), n0 = tn * (16 * TB);
    const int lda = (int)g.lda, ldb = (int)g.ldb;
    const int kbeg = split * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.A) + (AKM ? (long)kbeg * lda : 0), 0, (AKM ? kend - kbeg : g.M) * lda * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.B) + (BKM ? (long)kbeg * ldb : 0), 0, (BKM ? kend - kbeg : g.N) * ldb * 4, 0x00020000);

    KsOperand<TA, AKM> a[D];
    KsOperand<TB, BKM> b[D];
    f32x4 acc[TA][TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < D; ++s) {
        a[s].init(m0, lda, c, q, w + 4 * s);
        b[s].init(n0, ldb, c, q, w + 4 * s);
        a[s].adv *= D;
        b[s].adv *= D;
        a[s].fetch(ra_rsrc);
        b[s].fetch(rb_rsrc);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int G = (kend - kbeg + 15) >> 4;               // k groups of this range; this wave owns w, w+4, ...
    for (int g0 = w; g0 < G; g0 += 4 * D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            if (g0 + 4 * s < G) {                        // wave-uniform: past K a k-contiguous operand is not zero
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TA; ++i)
#pragma unroll
                        for (int j = 0; j < TB; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].v[e][i], b[s].v[e][j], acc[i][j], 0, 0, 0);
            }
            a[s].fetch(ra_rsrc);
            b[s].fetch(rb_rsrc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // the four partial tiles meet in LDS, RR registers at a time: [register][wave][lane]; wave w finishes registers
    // [QR w, QR w + QR) of the round.  Rows and columns are dealt like the operands: MFMA row c' = 4q + r of tile i is
    // strip row TA*c' + i, column c of tile j is strip column TB*c + j.
    const bool add_bias = g.bias != nullptr && split == 0;
#pragma unroll
    for (int round = 0; round < NR / RR; ++round) {
        if (round) __syncthreads();
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int reg = (i * TB + j) * 4 + r;
                    if (reg / RR == round) red[((reg % RR) * 4 + w) * 64 + lane] = acc[i][j][r];
                }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < QR; ++x) {
            const int lr = QR * w + x, reg = round * RR + lr;
            const float* p = red + (lr * 4) * 64 + lane;
            float val = (p[0] + p[64]) + (p[128] + p[192]);
            const int r = reg & 3, j = (reg >> 2) % TB, i = (reg >> 2) / TB;
            const int row = m0 + TA * (4 * q + r) + i;
            const int col = n0 + TB * c + j;
            if (add_bias) val += g.bias[col];
            if (g.epi != EPI_NONE) val = apply_epi(val, g.epi, g.aux ? g.aux[(long)row * g.ldaux + col] : 0.f);
            float* cp = g.C + (long)row * g.ldc + col;
            if (g.acc == ACC_STORE) *cp = val;
            else if (g.acc == ACC_ADD) *cp += val;
            else unsafeAtomicAdd(cp, val);
        }
    }
}

template <int TA, int TB, bool AKM, bool BKM>
__global__ __launch_bounds__(256) void gemm_ks_kernel(GemmArgs g, int tiles_n, int tiles) {
    int v = blockIdx.x;
    if ((gridDim.x & 7) == 0) v = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    gemm_ks_body<TA, TB, AKM, BKM>(g, v, tiles_n, tiles);
}

// Up to kGemmGroupMax INDEPENDENT products of one operand layout in one launch (the two Linear heads of the encoder, the two
// beat -> tick projections of the decoder, a module's leaf weight gradients): each is too small to fill the chip, and a
// launch of its own costs >= 5 us start to start whatever it does.  Workgroup v belongs to the product whose tile range
// [first[i], first[i+1]) holds it; every product keeps its own shape, bias, epilogue and accumulation mode.
struct GemmGroupArgs { int n; int first[kGemmGroupMax + 1]; int tiles_n[kGemmGroupMax]; GemmArgs g[kGemmGroupMax]; };

// several few-row products (one M for all) in one launch: blockIdx.y = product
template <int M>
__global__ __launch_bounds__(256) void gemv_rows_group_kernel(GemmGroupArgs a) {
    gemv_rows_body<M>(a.g[blockIdx.y], blockIdx.x * 4 + (threadIdx.x >> 6));
}

template <int TA, int TB, bool AKM, bool BKM>
__global__ __launch_bounds__(256) void gemm_ks_group_kernel(GemmGroupArgs a) {
    const int v = blockIdx.x;
    int i = 0;
#pragma unroll
    for (int k = 1; k < kGemmGroupMax; ++k)
        if (k < a.n && v >= a.first[k]) i = k;
    gemm_ks_body<TA, TB, AKM, BKM>(a.g[i], v - a.first[i], a.tiles_n[i], a.first[i + 1] - a.first[i]);
}

template <int TM, int TN>
int launch_cfg(const GemmArgs& g, dim3 grid, hipStream_t s, unsigned pad) {
    if (!g.a_kmajor && !g.b_kmajor) hipLaunchKernelGGL((gemm_kernel<TM, TN, false, false>), grid, dim3(256), pad, s, g);
    else if (!g.a_kmajor && g.b_kmajor) hipLaunchKernelGGL((gemm_kernel<TM, TN, false, true>), grid, dim3(256), pad, s, g);
    else if (g.a_kmajor && g.b_kmajor) hipLaunchKernelGGL((gemm_kernel<TM, TN, true, true>), grid, dim3(256), pad, s, g);
    else hipLaunchKernelGGL((gemm_kernel<TM, TN, true, false>), grid, dim3(256), pad, s, g);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

struct TileCfg { int tm, tn, wgs_per_cu; double eff; };
// block tile = 64*tm x 64*tn; residency from the LDS footprint (2 stages); eff = relative MFMA efficiency
const TileCfg kCfgs[] = {
    {1, 1, 4, 0.70}, {2, 2, 2, 0.60}, {3, 1, 2, 0.70}, {3, 2, 1, 0.70}, {3, 3, 1, 0.85},
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

// second pass of a split-K product with a non-linear epilogue: C = epi(C, aux)   (bias was added by split 0)
__global__ void gemm_epilogue_kernel(float* __restrict__ C, long ldc, int M, int N, const float* __restrict__ aux,
                                     long ldaux, int epi) {
    const long n = (long)M * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / N), c = (int)(i % N);
        float* p = C + (long)r * ldc + c;
        *p = apply_epi(*p, epi, aux ? aux[(long)r * ldaux + c] : 0.f);
    }
}

bool gemv_enabled() { return true; }
int g_force_cfg = -2, g_force_split = 0;      // -2: not read yet (inet_set_option keys 2, 3), -1: cost model
int g_direct = 1;    // direct kernels (inet_set_option key 5): 0 never, 1 (default) by shape, 2 direct whenever applicable, 3 big shapes only, 4 split-K first

struct DirectCfg { int ta, tb; };
const DirectCfg kDirect[] = {{3, 2}, {2, 2}, {3, 3}};
inline long tiles_of(const DirectCfg& c, const GemmArgs& g) { return (long)(g.M / (64 * c.ta)) * (g.N / (64 * c.tb)); }

// Direct (LDS-free) k-major x k-major product: returns 1 when the shape does not qualify, else the launch status.
int launch_gemm_direct(const GemmArgs& gin, hipStream_t s, int force_split) {
    GemmArgs g = gin;
    const bool nonlinear = g.epi != EPI_NONE;
    const int nbt = g.nbatch > 1 ? g.nbatch : 1;
    if (!g.a_kmajor || !g.b_kmajor || (g.M & 63) || (g.N & 63) || g.K < 64) return 1;
    if ((double)g.K * g.lda * 4 >= 2.0e9 || (double)g.K * g.ldb * 4 >= 2.0e9) return 1;
    if (nbt > 1 && (g.bias || nonlinear || g.acc == ACC_STORE)) return 1;
    const int kSplits[] = {1, 2, 4, 8, 16, 32};
    double best = 1e300;
    int bi = -1, bs = 1;
    constexpr int only_cfg = -1;
    for (int ci = 0; ci < 3; ++ci) {
        const DirectCfg& c = kDirect[ci];
        if (g.M % (64 * c.ta) || g.N % (64 * c.tb)) continue;
        if (only_cfg >= 0 && ci != only_cfg) continue;
        const long tiles = (long)(g.M / (64 * c.ta)) * (g.N / (64 * c.tb));
        for (int sp : kSplits) {
            if (sp > 1 && (g.K / sp < 64 || (nonlinear && g.acc != ACC_STORE))) break;
            if (force_split > 0 && sp != force_split) continue;
            const long wgs = tiles * sp * nbt;
            const long rounds = (wgs + 255) / 256;
            const double steps = (double)((g.K + sp - 1) / sp + 1) / 2;
            double cost = rounds * (steps * c.ta * c.tb * (64.0 / 2400.0) + 4.0);
            if (sp > 1) cost += 2.0 + (double)g.M * g.N * sp * nbt / 6.0e5 + (nonlinear ? 3.0 : 0.0);
            if (cost < best) { best = cost; bi = ci; bs = sp; }
        }
    }
    if (bi < 0 || (g_direct != 2 && (g.K / bs < 768 || (long)tiles_of(kDirect[bi], g) * bs * nbt < 192))) return 1;
    const DirectCfg& c = kDirect[bi];
    int kps = (g.K + bs - 1) / bs;
    kps = (kps + 1) / 2 * 2;
    const int splits = (g.K + kps - 1) / kps;
    g.k_per_split = kps;
    const bool two_pass = splits > 1 && nonlinear;
    if (splits > 1) {
        if (g.acc == ACC_STORE && pw_zero2d(g.C, g.ldc, g.M, g.N, s) != 0) return -2;
        g.acc = ACC_ATOMIC;
        if (two_pass) g.epi = EPI_NONE;
    }
    const int tiles_n = g.N / (64 * c.tb), tiles = tiles_n * (g.M / (64 * c.ta));
    char label[96];
    if (nbt > 1 && (g.K + kps - 1) / kps * kps != g.K) return 1;   // (the kernel derives the split count from the grid)
    if (nbt > 1) std::snprintf(label, sizeof label, "M%d N%d K%d TN d%dx%d s%d e%d x%d", g.M, g.N, g.K, 64 * c.ta, 64 * c.tb,
                               splits, gin.epi, nbt);
    else std::snprintf(label, sizeof label, "M%d N%d K%d TN d%dx%d s%d e%d", g.M, g.N, g.K, 64 * c.ta, 64 * c.tb, splits, gin.epi);
    ProfScope prof(PROF_GEMM, 2.0 * g.M * g.N * g.K * nbt, s, label,
                   4.0 * nbt * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N));
    const dim3 grid(tiles * splits * nbt);
    if (bi == 0) hipLaunchKernelGGL((gemm_tn_direct_kernel<3, 2>), grid, dim3(256), 0, s, g, tiles_n, tiles);
    else if (bi == 1) hipLaunchKernelGGL((gemm_tn_direct_kernel<2, 2>), grid, dim3(256), 0, s, g, tiles_n, tiles);
    else hipLaunchKernelGGL((gemm_tn_direct_kernel<3, 3>), grid, dim3(256), 0, s, g, tiles_n, tiles);
    int rc = hipGetLastError() == hipSuccess ? 0 : -2;
    if (rc == 0 && two_pass) {
        long n = (long)g.M * g.N;
        int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        hipLaunchKernelGGL(gemm_epilogue_kernel, dim3(blocks), dim3(256), 0, s, g.C, g.ldc, g.M, g.N, gin.aux, gin.ldaux,
                           gin.epi);
        rc = hipGetLastError() == hipSuccess ? 0 : -2;
    }
    return rc;
}

}  // namespace

// inet_set_option keys 2 / 3: force tile configuration `cfg` (index into kCfgs, -1 = cost model) / split-K factor
void gemm_set_force(int cfg, int split) {
    if (cfg >= -1) g_force_cfg = cfg;
    if (split >= 0) g_force_split = split;
}

struct KcCfg { int ta, tb; };
const KcCfg kKc[] = {{6, 6}, {6, 4}, {6, 2}, {4, 4}};

template <int TA, int TB>
void launch_kc(const GemmArgs& g, dim3 grid, hipStream_t s, int tiles_n) {
    if (g.b_kmajor) hipLaunchKernelGGL((gemm_kc_direct_kernel<TA, TB, true>), grid, dim3(256), 0, s, g, tiles_n);
    else hipLaunchKernelGGL((gemm_kc_direct_kernel<TA, TB, false>), grid, dim3(256), 0, s, g, tiles_n);
}

// Direct kernel for a k-contiguous A (forward and data-gradient products): no split-K, so it needs a tile
// configuration whose grid fills the chip by itself.  Returns 1 when the shape does not qualify.
int launch_gemm_kc_direct(const GemmArgs& g, hipStream_t s) {
    if (g.a_kmajor || (g.K & 63) || g.acc == ACC_ATOMIC) return 1;
    if ((double)g.M * g.lda * 4 >= 2.0e9 || (double)(g.b_kmajor ? g.K : g.N) * g.ldb * 4 >= 2.0e9) return 1;
    double best = 1e300;
    int bi = -1;
    for (int ci = 0; ci < 4; ++ci) {
        const KcCfg& c = kKc[ci];
        if (g.M % (32 * c.ta) || g.N % (32 * c.tb)) continue;
        const long wgs = (long)(g.M / (32 * c.ta)) * (g.N / (32 * c.tb));
        if (wgs < 192 && g_direct != 2) continue;
        const long rounds = (wgs + 255) / 256;
        const double cost = rounds * ((double)g.K / 4 * c.ta * c.tb * (32.0 / 1900.0) + 8.0);
        if (cost < best) { best = cost; bi = ci; }
    }
    if (bi < 0 || (g_direct != 2 && g.K < 256)) return 1;
    const KcCfg& c = kKc[bi];
    const int tiles_n = g.N / (32 * c.tb);
    const dim3 grid(tiles_n * (g.M / (32 * c.ta)));
    char label[96];
    std::snprintf(label, sizeof label, "M%d N%d K%d N%c d%dx%d s1 e%d", g.M, g.N, g.K, g.b_kmajor ? 'N' : 'T', 32 * c.ta,
                  32 * c.tb, g.epi);
    ProfScope prof(PROF_GEMM, 2.0 * g.M * g.N * g.K, s, label,
                   4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N));
    switch (bi) {
        case 0: launch_kc<6, 6>(g, grid, s, tiles_n); break;
        case 1: launch_kc<6, 4>(g, grid, s, tiles_n); break;
        case 2: launch_kc<6, 2>(g, grid, s, tiles_n); break;
        default: launch_kc<4, 4>(g, grid, s, tiles_n); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

struct KsCfg { int ta, tb; };
const KsCfg kKs[] = {{4, 4}, {4, 2}, {2, 2}, {6, 4}, {4, 6}};

template <int TA, int TB>
void launch_ks(const GemmArgs& g, dim3 grid, hipStream_t s, int tiles_n, int tiles) {
    if (g.a_kmajor) hipLaunchKernelGGL((gemm_ks_kernel<TA, TB, true, true>), grid, dim3(256), 0, s, g, tiles_n, tiles);
    else if (g.b_kmajor) hipLaunchKernelGGL((gemm_ks_kernel<TA, TB, false, true>), grid, dim3(256), 0, s, g, tiles_n, tiles);
    else hipLaunchKernelGGL((gemm_ks_kernel<TA, TB, false, false>), grid, dim3(256), 0, s, g, tiles_n, tiles);
}

// In-workgroup split-K kernel.  Medium / small products: one workgroup per output tile, the largest tile that still
// fills the chip.  Long weight-gradient products (k-major x k-major, K = T*B): 96x64 tiles, and when those are fewer than
// the CUs the k range is also split over the grid (2 ranges for M1536 N512: 1/4 of the atomics of an 8-way split).
// Returns 1 when the shape does not qualify.
int launch_gemm_ks(const GemmArgs& gin, hipStream_t s, int force_split) {
    GemmArgs g = gin;
    if (g.a_kmajor && !g.b_kmajor) return 1;
    if (!g.a_kmajor && (g.K & 15)) return 1;             // a k-contiguous operand has no zero-returning K tail
    if (g.acc == ACC_ATOMIC || g.K < 64) return 1;
    if ((double)(g.a_kmajor ? g.K : g.M) * g.lda * 4 >= 2.0e9 || (double)(g.b_kmajor ? g.K : g.N) * g.ldb * 4 >= 2.0e9) return 1;
    const bool nonlinear = g.epi != EPI_NONE;
    // rounds of 256 workgroups x MFMAs per workgroup, the smaller tiles charged for their higher L2 traffic per MFMA
    const double kL2[] = {1.0, 1.15, 1.5, 0.95, 0.95};
    int bi = -1, bs = 1;
    double best = 1e300;
    constexpr bool wide46 = true;
    for (int ci = 0; ci < 5; ++ci) {
        const KsCfg& c = kKs[ci];
        if (g.M % (16 * c.ta) || g.N % (16 * c.tb)) continue;
        if (ci == 3 && !(g.a_kmajor && g.K >= 2048)) continue;
        // 64 x 96 tiles (k-contiguous A): products whose 64 x 64 tiling needs two rounds of workgroups and whose 64 x 32 tiling pays
        // for it in L2 traffic (M1024 N1536 K512: 256 tiles instead of 384 / 768)
        if (ci == 4 && (g.a_kmajor || !wide46)) continue;
        const long tiles = (long)(g.M / (16 * c.ta)) * (g.N / (16 * c.tb));
        for (int sp = 1; sp <= 8; sp *= 2) {
            if (sp > 1 && (!g.a_kmajor || g.K / sp < 1024 || nonlinear)) break;
            if (force_split > 0 && sp != force_split) continue;
            const long wgs = tiles * sp;
            double cost = (double)((wgs + 255) / 256) * c.ta * c.tb * kL2[ci] / sp;
            if (sp > 1) cost *= 1.0 + 0.04 * sp;         // zero-fill + atomics
            if (cost < best) { best = cost; bi = ci; bs = sp; }
        }
    }
    if (bi < 0) return 1;
    const KsCfg& c = kKs[bi];
    // many tiles (several rounds of workgroups that share nothing): the LDS-tiled kernel is the better one there
    // (M12288 N1024 K256: 133 us here, 85 us LDS-tiled)
    if ((long)(g.M / (16 * c.ta)) * (g.N / (16 * c.tb)) * bs > 1024 && g_direct != 4) return 1;
    int kps = (g.K + bs - 1) / bs;
    kps = (kps + 15) / 16 * 16;
    const int splits = (g.K + kps - 1) / kps;
    g.k_per_split = kps;
    if (splits > 1) {
        if (g.acc == ACC_STORE && pw_zero2d(g.C, g.ldc, g.M, g.N, s) != 0) return -2;
        g.acc = ACC_ATOMIC;
    }
    const int tiles_n = g.N / (16 * c.tb), tiles = tiles_n * (g.M / (16 * c.ta));
    const dim3 grid(tiles * splits);
    char label[96];
    std::snprintf(label, sizeof label, "M%d N%d K%d %c%c k%dx%d s%d e%d", g.M, g.N, g.K, g.a_kmajor ? 'T' : 'N',
                  g.b_kmajor ? 'N' : 'T', 16 * c.ta, 16 * c.tb, splits, g.epi);
    ProfScope prof(PROF_GEMM, 2.0 * g.M * g.N * g.K, s, label,
                   4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N));
    switch (bi) {
        case 0: launch_ks<4, 4>(g, grid, s, tiles_n, tiles); break;
        case 1: launch_ks<4, 2>(g, grid, s, tiles_n, tiles); break;
        case 2: launch_ks<2, 2>(g, grid, s, tiles_n, tiles); break;
        case 4: launch_ks<4, 6>(g, grid, s, tiles_n, tiles); break;
        default:
            if (!g.a_kmajor) return 1;
            hipLaunchKernelGGL((gemm_ks_kernel<6, 4, true, true>), grid, dim3(256), 0, s, g, tiles_n, tiles);
            break;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// inet_set_option key 5: 0 = LDS-tiled kernels only, 1 = cost model (default), 2 = the direct kernel whenever it applies
void gemm_set_direct(int mode) { g_direct = mode; }

template <int TA, int TB>
void launch_ks_group(const GemmGroupArgs& a, bool akm, bool bkm, dim3 grid, hipStream_t s) {
    if (akm) hipLaunchKernelGGL((gemm_ks_group_kernel<TA, TB, true, true>), grid, dim3(256), 0, s, a);
    else if (bkm) hipLaunchKernelGGL((gemm_ks_group_kernel<TA, TB, false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm_ks_group_kernel<TA, TB, false, false>), grid, dim3(256), 0, s, a);
}

// Independent products in ONE launch of the workgroup split-K kernel when all of them have the same operand layout and a
// common tile shape divides them (a switch of earlier rounds ran them one after the other); else one launch each.
int launch_gemm_group(const GemmArgs* list, int n, hipStream_t s) {
    constexpr bool grouped = true;
    {   // a group of few-row products of one M (the beat -> tick projections of a b = 1 decode call): one launch of the wave-per-column kernel
        bool gv = grouped && n >= 2 && n <= kGemmGroupMax && g_force_cfg < 0 && gemv_enabled() && list[0].M >= 1 && list[0].M <= 8;
        int maxN = 0;
        double flops = 0, bytes = 0;
        for (int i = 0; i < n && gv; ++i) {
            const GemmArgs& g = list[i];
            gv = g.M == list[0].M && g.N > 0 && g.K > 0 && !g.a_kmajor && !g.b_kmajor && g.nbatch <= 1 && g.acc != ACC_ATOMIC;
            maxN = g.N > maxN ? g.N : maxN;
            flops += 2.0 * g.M * g.N * g.K; bytes += 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N);
        }
        if (gv) {
            GemmGroupArgs a{};
            a.n = n;
            for (int i = 0; i < n; ++i) {
                if (list[i].acc == ACC_ADD && side_is(s) && side_order_dest(list[i].C, s) != 0) return -2;
                a.g[i] = list[i];
            }
            char label[64];
            std::snprintf(label, sizeof label, "group%d M%d N%d K%d NT gemv e%d", n, list[0].M, list[0].N, list[0].K, list[0].epi);
            ProfScope prof(PROF_GEMM, flops, s, label, bytes);
            const dim3 grid((maxN + 3) / 4, n);
            switch (list[0].M) {
                case 1: hipLaunchKernelGGL(gemv_rows_group_kernel<1>, grid, dim3(256), 0, s, a); break;
                case 2: hipLaunchKernelGGL(gemv_rows_group_kernel<2>, grid, dim3(256), 0, s, a); break;
                case 3: hipLaunchKernelGGL(gemv_rows_group_kernel<3>, grid, dim3(256), 0, s, a); break;
                case 4: hipLaunchKernelGGL(gemv_rows_group_kernel<4>, grid, dim3(256), 0, s, a); break;
                case 5: hipLaunchKernelGGL(gemv_rows_group_kernel<5>, grid, dim3(256), 0, s, a); break;
                case 6: hipLaunchKernelGGL(gemv_rows_group_kernel<6>, grid, dim3(256), 0, s, a); break;
                case 7: hipLaunchKernelGGL(gemv_rows_group_kernel<7>, grid, dim3(256), 0, s, a); break;
                default: hipLaunchKernelGGL(gemv_rows_group_kernel<8>, grid, dim3(256), 0, s, a); break;
            }
            return hipGetLastError() == hipSuccess ? 0 : -2;
        }
    }
    bool ok = grouped && n >= 2 && n <= kGemmGroupMax && g_direct > 0 && g_direct != 3 && g_force_cfg < 0;
    for (int i = 0; i < n && ok; ++i) {
        const GemmArgs& g = list[i];
        ok = g.M > 0 && g.N > 0 && g.K >= 64 && g.nbatch <= 1 && g.acc != ACC_ATOMIC && !(g.a_kmajor && !g.b_kmajor) &&
             (g.a_kmajor || (g.K & 15) == 0) && g.a_kmajor == list[0].a_kmajor && g.b_kmajor == list[0].b_kmajor &&
             (double)(g.a_kmajor ? g.K : g.M) * g.lda * 4 < 2.0e9 && (double)(g.b_kmajor ? g.K : g.N) * g.ldb * 4 < 2.0e9;
    }
    int bi = -1;
    long total = 0;
    if (ok) {
        // the largest tile that divides every product and still gives the launch >= 192 workgroups; else the smallest that divides
        for (int ci = 0; ci < 3; ++ci) {
            const KsCfg& c = kKs[ci];
            bool div = true;
            long wgs = 0;
            for (int i = 0; i < n; ++i) {
                if (list[i].M % (16 * c.ta) || list[i].N % (16 * c.tb)) div = false;
                else wgs += (long)(list[i].M / (16 * c.ta)) * (list[i].N / (16 * c.tb));
            }
            if (!div) continue;
            bi = ci; total = wgs;
            if (wgs >= 192) break;
        }
        if (bi < 0 || total > 2048) ok = false;
    }
    if (!ok) {
        for (int i = 0; i < n; ++i) { const int rc = launch_gemm(list[i], s); if (rc != 0) return rc; }
        return 0;
    }
    const KsCfg& c = kKs[bi];
    GemmGroupArgs a{};
    a.n = n;
    double flops = 0, bytes = 0;
    for (int i = 0; i < n; ++i) {
        GemmArgs g = list[i];
        if (g.acc == ACC_ADD && side_is(s) && side_order_dest(g.C, s) != 0) return -2;
        g.k_per_split = (g.K + 15) / 16 * 16;
        a.g[i] = g;
        a.tiles_n[i] = g.N / (16 * c.tb);
        a.first[i + 1] = a.first[i] + a.tiles_n[i] * (g.M / (16 * c.ta));
        flops += 2.0 * g.M * g.N * g.K;
        bytes += 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N);
    }
    char label[96];
    std::snprintf(label, sizeof label, "group%d M%d N%d K%d %c%c k%dx%d e%d", n, list[0].M, list[0].N, list[0].K,
                  list[0].a_kmajor ? 'T' : 'N', list[0].b_kmajor ? 'N' : 'T', 16 * c.ta, 16 * c.tb, list[0].epi);
    ProfScope prof(PROF_GEMM, flops, s, label, bytes);
    const dim3 grid(a.first[n]);
    const bool akm = list[0].a_kmajor, bkm = list[0].b_kmajor;
    switch (bi) {
        case 0: launch_ks_group<4, 4>(a, akm, bkm, grid, s); break;
        case 1: launch_ks_group<4, 2>(a, akm, bkm, grid, s); break;
        default: launch_ks_group<2, 2>(a, akm, bkm, grid, s); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Tile / split-K selection by a small cost model (microseconds), calibrated on MI355X (profiles/r01_*):
//  * a workgroup alone on a CU spends ~0.7 us per 32-deep chunk on the load -> LDS -> MFMA dependency, whatever the
//    tile; co-resident workgroups overlap that latency until the CU's MFMA pipe (tm*tn*0.43 us per chunk) is full;
//  * the grid runs in ceil(WGs / (256 * residency)) rounds;
//  * split-K adds a zero-fill, f32 atomics (~6e5 elements/us chip-wide) and, when the epilogue is non-linear, a
//    second elementwise pass.
// Constants refitted against tools/gemm_sweep.py (7 shapes x 25 forced tile/split points): the picks are within 3 % of
// the best forced configuration on every swept shape.
// Big shapes (M or K = T*B = 6144) end up on 192-wide tiles with exactly 256 workgroups; small ones on 64x64
// tiles split until every CU holds several workgroups.
int launch_gemm(const GemmArgs& gin, hipStream_t s) {
    // Accumulations issued on the rotating side streams (leaf weight gradients): when a module is applied more than once per
    // step (a per-tick free-running pass) two of them may target the same tensor from different streams -- the second one is
    // ordered behind the first (side.hip side_order_dest; no cost when every tensor has one writer per step).
    if (gin.acc == ACC_ADD && side_is(s)) {
        for (int i = 0; i < (gin.nbatch > 1 ? gin.nbatch : 1); ++i)
            if (side_order_dest(gin.C + i * gin.batchC, s) != 0) return -2;
    }
    GemmArgs g = gin;
    if (g.M <= 0 || g.N <= 0) return 0;
    if (g.K <= 0) return -1;
    if (g_force_cfg == -2) g_force_cfg = -1;
    if (g.nbatch > 1) {
        // several products of one shape: one launch of the shared-strip direct kernel when it applies (half the split-K
        // factor of a single product for the same 256 workgroups), else one product after the other
        constexpr bool batched = true;
        int rc = 1;
        if (batched && g_direct > 0 && g_direct != 4 && g_force_cfg < 0) rc = launch_gemm_direct(gin, s, g_force_split > 0 ? g_force_split : 0);
        if (rc != 1) return rc;
        for (int i = 0; i < gin.nbatch; ++i) {
            GemmArgs one = gin;
            one.nbatch = 0;
            one.A += i * gin.batchA; one.B += i * gin.batchB; one.C += i * gin.batchC;
            if ((rc = launch_gemm(one, s)) != 0) return rc;
        }
        return 0;
    }
    if (g.M <= 8 && !g.a_kmajor && !g.b_kmajor && g.nbatch <= 1 && g_force_cfg < 0 && g.acc != ACC_ATOMIC && gemv_enabled()) {
        char label[64];
        std::snprintf(label, sizeof label, "M%d N%d K%d NT gemv e%d", g.M, g.N, g.K, g.epi);
        ProfScope prof(PROF_GEMM, 2.0 * g.M * g.N * g.K, s, label, 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N));
        const dim3 grid((g.N + 3) / 4);
        switch (g.M) {
            case 1: hipLaunchKernelGGL(gemv_rows_kernel<1>, grid, dim3(256), 0, s, g); break;
            case 2: hipLaunchKernelGGL(gemv_rows_kernel<2>, grid, dim3(256), 0, s, g); break;
            case 3: hipLaunchKernelGGL(gemv_rows_kernel<3>, grid, dim3(256), 0, s, g); break;
            case 4: hipLaunchKernelGGL(gemv_rows_kernel<4>, grid, dim3(256), 0, s, g); break;
            case 5: hipLaunchKernelGGL(gemv_rows_kernel<5>, grid, dim3(256), 0, s, g); break;
            case 6: hipLaunchKernelGGL(gemv_rows_kernel<6>, grid, dim3(256), 0, s, g); break;
            case 7: hipLaunchKernelGGL(gemv_rows_kernel<7>, grid, dim3(256), 0, s, g); break;
            default: hipLaunchKernelGGL(gemv_rows_kernel<8>, grid, dim3(256), 0, s, g); break;
        }
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    const int force_cfg = g_force_cfg, force_split = g_force_split;
    const int kSplits[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
    const bool nonlinear = g.epi != EPI_NONE;
    double best = 1e300;
    int bi = 0, bs = 1;
    for (int ci = 0; ci < kNumCfgs; ++ci) {
        const TileCfg& c = kCfgs[ci];
        const long tiles = (long)((g.M + 64 * c.tm - 1) / (64 * c.tm)) * ((g.N + 64 * c.tn - 1) / (64 * c.tn));
        for (int sp : kSplits) {
            if (sp > 1 && (g.K / sp < 128 || (nonlinear && g.acc != ACC_STORE))) break;
            const long wgs = tiles * sp;
            const long slots = 256L * c.wgs_per_cu;
            const long rounds = (wgs + slots - 1) / slots;
            const double conc = (double)(wgs < slots ? (wgs + 255) / 256 : c.wgs_per_cu);
            const double chunks = (double)(((g.K + sp - 1) / sp + 31) / 32);
            const double t_mfma = c.tm * c.tn * 0.4267 / c.eff;
            const double lat = 0.7 + 0.1 * (c.tm + c.tn - 2);
            double cost = rounds * ((chunks * lat > conc * chunks * t_mfma ? chunks * lat : conc * chunks * t_mfma) + 1.5);
            if (sp > 1) cost += 2.0 + (double)g.M * g.N * sp / 6.0e5 + (nonlinear ? 3.0 : 0.0);
            if (cost < best) { best = cost; bi = ci; bs = sp; }
        }
    }
    if (g_direct > 0 && force_cfg < 0) {
        // The long weight-gradient products go to the shared-strip direct kernel first.  Workgroup split-K with 96x64 tiles
        // (g_direct = 4 tries it first) is 8-15 % faster alone (no zero-fill, 1/4 of the atomics) but 1 % slower
        // in the training step, where its doubled L2 traffic competes with the BPTT chain on the other stream.
        int rc = 1;
        if (gin.a_kmajor && g_direct != 4) rc = launch_gemm_direct(gin, s, force_split);
        else if (!gin.a_kmajor) rc = launch_gemm_kc_direct(gin, s);
        if (rc == 1 && g_direct != 3) rc = launch_gemm_ks(gin, s, force_split);
        if (rc == 1 && gin.a_kmajor) rc = launch_gemm_direct(gin, s, force_split);
        if (rc != 1) return rc;
    }
    if (force_cfg >= 0 && force_cfg < kNumCfgs) {
        bi = force_cfg;
        bs = (nonlinear && g.acc != ACC_STORE) ? 1 : (force_split > 0 ? force_split : 1);
    }
    const TileCfg& c = kCfgs[bi];
    const int BM = 64 * c.tm, BN = 64 * c.tn;
    int splits = bs;
    int kps = (g.K + splits - 1) / splits;
    kps = (kps + 31) / 32 * 32;
    splits = (g.K + kps - 1) / kps;
    g.k_per_split = kps;
    const bool two_pass = splits > 1 && nonlinear;
    if (splits > 1) {
        if (g.acc == ACC_STORE) {
            if (pw_zero2d(g.C, g.ldc, g.M, g.N, s) != 0) return -2;
        }
        g.acc = ACC_ATOMIC;
        if (two_pass) g.epi = EPI_NONE;
    }
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, splits);
    char label[96];
    std::snprintf(label, sizeof label, "M%d N%d K%d %c%c t%dx%d s%d e%d", g.M, g.N, g.K, g.a_kmajor ? 'T' : 'N',
                  g.b_kmajor ? 'N' : 'T', 64 * c.tm, 64 * c.tn, splits, gin.epi);
    int rc;
    {
        ProfScope prof(PROF_GEMM, 2.0 * g.M * g.N * g.K, s, label, 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N));
        // Leaf GEMMs on the side stream: 64x64 tiles would sit 4 to a CU and take 147 of its 160 KB of LDS, so a BPTT
        // step kernel arriving on the main stream (16-32 KB) has to wait for one of them to retire.  A few KB of unused
        // dynamic LDS caps them at 3 per CU and leaves the step kernels room to co-reside (5.22 -> 5.18 ms per step;
        // capping at 2 per CU costs the GEMMs more than it gives: 5.40).
        const unsigned pad = (bi == 0 && side_is(s)) ? 4608u : 0u;
        switch (bi) {
            case 0: rc = launch_cfg<1, 1>(g, grid, s, pad); break;
            case 1: rc = launch_cfg<2, 2>(g, grid, s, 0); break;
            case 2: rc = launch_cfg<3, 1>(g, grid, s, 0); break;
            case 3: rc = launch_cfg<3, 2>(g, grid, s, 0); break;
            default: rc = launch_cfg<3, 3>(g, grid, s, 0); break;
        }
        if (rc == 0 && two_pass) {
            long n = (long)g.M * g.N;
            int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
            hipLaunchKernelGGL(gemm_epilogue_kernel, dim3(blocks), dim3(256), 0, s, g.C, g.ldc, g.M, g.N, gin.aux,
                               gin.ldaux, gin.epi);
            rc = hipGetLastError() == hipSuccess ? 0 : -2;
        }
    }
    return rc;
}
