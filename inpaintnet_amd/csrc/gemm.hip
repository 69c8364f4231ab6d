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

int g_force_cfg = -2, g_force_split = 0;      // -2: not read yet (INET_GEMM_FORCE="cfg,split"), -1: cost model

}  // namespace

// inet_set_option keys 2 / 3: force tile configuration `cfg` (index into kCfgs, -1 = cost model) / split-K factor
void gemm_set_force(int cfg, int split) {
    if (cfg >= -1) g_force_cfg = cfg;
    if (split >= 0) g_force_split = split;
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
    GemmArgs g = gin;
    if (g.M <= 0 || g.N <= 0) return 0;
    if (g.K <= 0) return -1;
    if (g_force_cfg == -2) {
        g_force_cfg = -1;
        if (const char* v = std::getenv("INET_GEMM_FORCE")) std::sscanf(v, "%d,%d", &g_force_cfg, &g_force_split);
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
