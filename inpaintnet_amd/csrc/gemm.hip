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
#include "prof.h"

namespace {

template <int R, bool KM>
struct OperandTile {
    static constexpr int NV = R / 32;                       // float4 per thread per 32-deep chunk
    static constexpr int PITCH = KM ? (R + 4) : 36;         // words
    static constexpr int WORDS = KM ? 32 * PITCH : R * PITCH;

    // global -> registers.  rows_total: valid extent of the row index, kend: valid extent of k
    __device__ static __forceinline__ void load(f32x4 (&v)[NV], const float* __restrict__ P, long ld,
                                                int row0, int rows_total, int k0, int kend, int t) {
        if (!KM) {
            const int c4 = t & 7;
            const int k = k0 + c4 * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int row = row0 + (t >> 3) + 32 * i;
                f32x4 x = {0.f, 0.f, 0.f, 0.f};
                if (row < rows_total) {
                    const float* p = P + (long)row * ld + k;
                    if (k + 3 < kend) {
                        x = ld4u(p);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (k + e < kend) x[e] = p[e];
                    }
                }
                v[i] = x;
            }
        } else {
            constexpr int VPR = R / 4;                       // float4 per k-row
            constexpr int KSTEP = 256 / VPR;
            const int c4 = t % VPR;
            const int row = row0 + c4 * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int k = k0 + t / VPR + KSTEP * i;
                f32x4 x = {0.f, 0.f, 0.f, 0.f};
                if (k < kend) {
                    const float* p = P + (long)k * ld + row;
                    if (row + 3 < rows_total) {
                        x = ld4u(p);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (row + e < rows_total) x[e] = p[e];
                    }
                }
                v[i] = x;
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
            constexpr int KSTEP = 256 / VPR;
            const int c4 = t % VPR;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int k = t / VPR + KSTEP * i;
                *reinterpret_cast<f32x4*>(lds + k * PITCH + c4 * 4) = v[i];
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

template <int BM, int BN, bool AKM, bool BKM>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    using TA = OperandTile<BM, AKM>;
    using TB = OperandTile<BN, BKM>;
    constexpr int TM = BM / 64, TN = BN / 64;
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
    if (nchunks > 0) {
        TA::load(ra, g.A, g.lda, m0, g.M, kbeg, kend, t);
        TB::load(rb, g.B, g.ldb, n0, g.N, kbeg, kend, t);
        TA::store(ra, lds, t);
        TB::store(rb, lds + TA::WORDS, t);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const float* As = lds + (c & 1) * STAGE;
        const float* Bs = As + TA::WORDS;
        const bool more = c + 1 < nchunks;
        if (more) {
            TA::load(ra, g.A, g.lda, m0, g.M, kbeg + (c + 1) * 32, kend, t);
            TB::load(rb, g.B, g.ldb, n0, g.N, kbeg + (c + 1) * 32, kend, t);
        }
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

template <int BM, int BN>
int launch_cfg(const GemmArgs& g, dim3 grid, hipStream_t s) {
    if (!g.a_kmajor && !g.b_kmajor) hipLaunchKernelGGL((gemm_kernel<BM, BN, false, false>), grid, dim3(256), 0, s, g);
    else if (!g.a_kmajor && g.b_kmajor) hipLaunchKernelGGL((gemm_kernel<BM, BN, false, true>), grid, dim3(256), 0, s, g);
    else if (g.a_kmajor && g.b_kmajor) hipLaunchKernelGGL((gemm_kernel<BM, BN, true, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, true, false>), grid, dim3(256), 0, s, g);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

// Picks the block tile and a split-K factor so that even the skinny (M = batch)
// and the short-and-deep (wgrad: K = T*B) products put >= ~2 workgroups on each
// of the 256 CUs.  Split-K partial sums are combined with f32 hardware atomics
// into a zeroed (ACC_STORE) or live (ACC_ADD) destination.
int launch_gemm(const GemmArgs& gin, hipStream_t s) {
    GemmArgs g = gin;
    if (g.M <= 0 || g.N <= 0) return 0;
    if (g.K <= 0) return -1;
    const long tilesL = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    const bool useL = tilesL >= 192;
    const int BM = useL ? 128 : 64, BN = BM;
    const long tiles = (long)((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    int splits = 1;
    if (g.epi == EPI_NONE && tiles < 384 && g.K >= 256) {
        splits = (int)((512 + tiles - 1) / tiles);
        const int maxs = g.K / 128;
        if (splits > maxs) splits = maxs;
        if (splits < 1) splits = 1;
    }
    int kps = (g.K + splits - 1) / splits;
    kps = (kps + 31) / 32 * 32;
    splits = (g.K + kps - 1) / kps;
    g.k_per_split = kps;
    if (splits > 1) {
        if (g.acc == ACC_STORE) {
            if (hipMemset2DAsync(g.C, g.ldc * sizeof(float), 0, (size_t)g.N * sizeof(float), g.M, s) != hipSuccess)
                return -2;
        }
        g.acc = ACC_ATOMIC;
    }
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, splits);
    ProfScope prof(PROF_GEMM, 2.0 * g.M * g.N * g.K, s);
    return useL ? launch_cfg<128, 128>(g, grid, s) : launch_cfg<64, 64>(g, grid, s);
}
