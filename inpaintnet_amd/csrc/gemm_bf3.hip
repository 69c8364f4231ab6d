// Large products on the bf16 matrix cores AT FP32 ACCURACY (round 3).
// Every f32 operand value is split exactly into three bf16 pieces, x = x0 + x1 + x2 (3 x 8 mantissa bits: nothing is lost),
// and a product a*b is the sum of the piece products a_i*b_j -- each exact in the f32 accumulator of
// v_mfma_f32_16x16x32_bf16.  All nine terms (mode 9) are the products of fp32 arithmetic, only the order of the f32 summation
// differs from the f32-input MFMA kernels of gemm.hip; mode 6 drops the three terms below 2^-24 |ab|.  A bf16 MFMA issues
// 16x the MACs per cycle of the f32-input one, so nine of them cost 0.56 of the f32-input product.
//
// Operands are "piece buffers" (gemm_bf3.h): both sides of the product in the MFMA's fragment order, so that
//   * the operand stream global -> LDS is a plain copy of 1 KB fragments, done by the LDS-DMA path (global_load_lds_dwordx4:
//     no staging registers, no ds_write pass), two stages, one barrier per 32-wide k block;
//   * fragments leave LDS with lane-contiguous ds_read_b128 (conflict-free, no swizzle);
//   * NT / NN / TN products are one kernel: the layout of the source array only matters to whoever writes the pieces
//     (bf3_split here; a producer kernel can write them directly).
// Workgroup = 8 waves (two per SIMD: one wave's LDS reads hide behind the other's MFMAs), tile 192 x 192 (wave tile 96 x 48)
// or 192 x 128 (48 x 64); 256 workgroups = one per CU for the encoder's shapes (6144 x 1536 x 1024: 32 x 8 tiles).
// Measured (tools/exp_gemm_bf3.hip, profiles/r03_r_gemm_bf3.txt): 6144 x 1536 x 1024 in 108.6 us against 147 us for the
// f32-input direct kernel (80 us with six products).
#include <cstdio>
#include <cstdlib>
#include "common.h"
#include "prof.h"
#include "gemm_bf3.h"
#include "side.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Edge cases (tests/test_gpu_kernels.py::test_gemm_bf3_edge_operands): denormals split like any other value (bf16 has f32's
// exponent range); a finite |x| above the largest bf16 (3.3895e38) would ROUND to +-inf and leave inf - inf = NaN in the
// residual where the f32-input kernels return a finite product, so there the first piece is truncated instead (still exact:
// the residual just carries one more bit); +-inf and NaN stay non-finite in the first piece and poison the row / column of
// the result, as they do -- as inf or NaN, by IEEE rules -- in the f32-input kernels: finite in, finite out; non-finite in,
// non-finite out at the same positions.
__device__ __forceinline__ void split3(float x, __bf16& a0, __bf16& a1, __bf16& a2) {
    a0 = (__bf16)x;                                  // round to nearest: |x - a0| <= 2^-9 |x|
    if (__builtin_isinf((float)a0) && !__builtin_isinf(x))
        a0 = (__bf16)__builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u);
    const float r1 = x - (float)a0;                  // exact
    a1 = (__bf16)r1;
    a2 = (__bf16)(r1 - (float)a1);                   // exact, and fits 8 bits
}
__device__ __forceinline__ void split8_store(const float* v, unsigned char* dst, long piece_bytes) {
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 a, b, c;
        split3(v[j], a, b, c);
        p0[j] = a; p1[j] = b; p2[j] = c;
    }
    *reinterpret_cast<bf16x8*>(dst) = p0;
    *reinterpret_cast<bf16x8*>(dst + piece_bytes) = p1;
    *reinterpret_cast<bf16x8*>(dst + 2 * piece_bytes) = p2;
}

struct SplitArgs {
    const float* X; long ld;
    unsigned char* P; long piece_bytes; int kb_total, rb0, kb0, R, K;
    int rb_mul;                                      // source row block rb lands in row block rb0 + rb * rb_mul (rows form only)
};

// k-contiguous source: one thread = one lane of one fragment (32 contiguous bytes of a source row)
__global__ __launch_bounds__(256) void bf3_split_rows_kernel(SplitArgs a) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = id & 63;
    const long frag = id >> 6;
    const int KB = a.K / 32;
    const long rb = frag / KB; const int kb = (int)(frag - rb * KB);
    if (rb >= a.R / 16) return;
    const long off = (rb * 16 + (lane & 15)) * a.ld + kb * 32 + (lane >> 4) * 8;
    const f32x4 v0 = ld4u(a.X + off), v1 = ld4u(a.X + off + 4);
    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    split8_store(v, a.P + (((a.rb0 + rb * a.rb_mul) * a.kb_total + a.kb0 + kb) * 64 + lane) * 16, a.piece_bytes);
}

// k-major source X[k * ld + r]: a workgroup transposes one k block (32 source rows) x 64 r through LDS
__global__ __launch_bounds__(256) void bf3_split_cols_kernel(SplitArgs a) {
    __shared__ float tile[32][65];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int kb = blockIdx.x, r0 = blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int kk = (t >> 4) + 16 * i, rr = (t & 15) * 4;
        const long off = (long)(kb * 32 + kk) * a.ld + r0 + rr;
        const f32x4 v = ld4u(a.X + off);
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[kk][rr + j] = v[j];
    }
    __syncthreads();
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tile[8 * (lane >> 4) + j][16 * w + (lane & 15)];
    const long rb = a.rb0 + r0 / 16 + w;
    split8_store(v, a.P + ((rb * a.kb_total + a.kb0 + kb) * 64 + lane) * 16, a.piece_bytes);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Tile order of a launch whose workgroups are ONE tile each (no k split, one product).  Workgroup ids go round-robin over
// the 8 XCDs and every XCD has its own L2, so what an XCD's 32 CUs run AT THE SAME TIME decides what that L2 serves twice:
// XCD x = (xi, xj) of an xm x xn arrangement owns the rm x rn tiles of its region, and walks it in blocks of bm x bn = 32
// tiles (one per CU): a block reads bm A strips and bn B strips through that L2 once.  Round 3 gave an XCD a contiguous range
// of the tm-major list -- 2 x 16 tiles at a time for the 32 x 16 tiles of the layer-1 input product: 18 strips per block, every
// B strip fetched by every XCD twice (441 MB of memory-side traffic for 132 MB of operands + result); 4 x 8 blocks of an
// 8 x 8 region need 12.  bm == 0: the contiguous-range order.
struct Bf3Map { int xn, rm, rn, bm, bn; };

template <int WM, int WN, int RM, int RN, int NP>
__global__ __launch_bounds__(64 * WM * WN) void gemm_bf3_kernel(Bf3Gemm g, int tiles_m, int tiles_n, int kb_per, Bf3Map mp) {
    constexpr int NW = WM * WN, TMB = WM * RM, TNB = WN * RN;
    constexpr int STAGE = (TMB + TNB) * 3 * 1024;
    constexpr int CH = (TMB + TNB) * 3;                        // 1 KB fragments per stage
    constexpr int CPW = (CH + NW - 1) / NW;                    // fragments per wave and stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w / WN, wn = w % WN;
    // Consecutive workgroup ids go round-robin over the 8 XCDs: give each XCD a contiguous range of tiles (neighbours in
    // that range share their A strip and walk the same B strips through that XCD's L2).
    const int nb = gridDim.x, id = blockIdx.x;
    const int tid = (nb % 8 == 0) ? (id % 8) * (nb / 8) + id / 8 : id;
    const int per_prob = tiles_m * tiles_n * g.ksplit;
    const int prob = tid / per_prob, v = tid - prob * per_prob;
    const int ks = v / (tiles_m * tiles_n), tt = v - ks * (tiles_m * tiles_n);
    int tm = tt / tiles_n, tn = tt - tm * tiles_n;
    if (mp.bm > 0) {
        const int x = id & 7, slot = id >> 3, per = mp.bm * mp.bn;
        const int blk = slot / per, in = slot - blk * per, bpr = mp.rn / mp.bn;
        tm = (x / mp.xn) * mp.rm + (blk / bpr) * mp.bm + in / mp.bn;
        tn = (x % mp.xn) * mp.rn + (blk % bpr) * mp.bn + in % mp.bn;
    }
    const unsigned char* const Ap = prob ? g.A2 : g.A;
    const unsigned char* const Bp = prob ? g.B2 : g.B;
    float* const Cp = prob ? g.C2 : g.C;
    const int kb0 = ks * kb_per;
    // fragment c of a stage: c < 3 TMB: A piece c / TMB, row block c % TMB;  then B alike
    const unsigned char* gsrc[CPW]; int loff[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int c = w + i * NW;
        const int cc = c < CH ? c : CH - 1;                    // a short last round repeats the last fragment (same bytes)
        if (cc < TMB * 3) {
            const int p = cc / TMB, rbl = cc % TMB;
            int rbg = tm * TMB + rbl;
            const unsigned char* src = Ap;
            long pstride = g.a_piece;
            if (g.a_alt_from && rbg >= g.a_alt_from) { rbg -= g.a_alt_from; src = prob ? g.A2_alt : g.A_alt; pstride = g.a_alt_piece; }
            gsrc[i] = src + p * pstride + ((long)rbg * g.a_kb + kb0) * 1024 + lane * 16;
        } else {
            const int c2 = cc - TMB * 3, p = c2 / TNB, rbl = c2 % TNB;
            gsrc[i] = Bp + p * g.b_piece + ((long)(tn * TNB + rbl) * g.b_kb + kb0) * 1024 + lane * 16;
        }
        loff[i] = cc * 1024;
    }
    auto fill = [&](int kb, unsigned char* stage) {            // LDS destination: wave-uniform base + lane * 16
#pragma unroll
        for (int i = 0; i < CPW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[i] + (long)kb * 1024),
                                             (__attribute__((address_space(3))) void*)(stage + loff[i]), 16, 0, 0);
    };
    f32x4 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    fill(0, smem);
    __syncthreads();
    for (int kb = 0; kb < kb_per; ++kb) {
        const unsigned char* sa = smem + (kb & 1) * STAGE + lane * 16;
        const unsigned char* sb = sa + TMB * 3 * 1024;
        if (kb + 1 < kb_per) fill(kb + 1, smem + ((kb + 1) & 1) * STAGE);   // (everyone left that stage at the last barrier)
        bf16x8 Af[RM][3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < RM; ++i) Af[i][p] = *reinterpret_cast<const bf16x8*>(sa + (p * TMB + wm * RM + i) * 1024);
#pragma unroll
        for (int pj = 0; pj < 3; ++pj) {
            bf16x8 Bf[RN];
#pragma unroll
            for (int j = 0; j < RN; ++j) Bf[j] = *reinterpret_cast<const bf16x8*>(sb + (pj * TNB + wn * RN + j) * 1024);
#pragma unroll
            for (int pi = 0; pi < 3; ++pi) {
                if (NP == 6 && pi + pj > 2) continue;
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[i][pi], Bf[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();                                       // (carries the vmcnt(0) of this wave's fills)
    }
    // epilogue: lane (c, q) holds rows 4q + r, column c of each 16 x 16 tile
    const int c = lane & 15, q = lane >> 4;
    const bool atomic = g.ksplit > 1;
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int col = (tn * TNB + wn * RN + j) * 16 + c;
        float bv = 0.f;
        if (g.bias && ks == 0) bv = (g.bias2 && col >= g.bias2_from) ? g.bias2[col - g.bias2_from] : g.bias[col];
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            const long row = (long)(tm * TMB + wm * RM + i) * 16 + 4 * q;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float val = acc[i][j][r] + bv;
                float* dst = Cp + (row + r) * g.ldc + col;
                if (atomic) { atomicAdd(dst, val); continue; }
                if (g.epi == EPI_MUL_AUX) val *= g.aux[(row + r) * g.ldaux + col];
                *dst = g.acc == ACC_ADD ? *dst + val : val;
            }
        }
    }
}

int g_mode = -1;

template <int WM, int WN, int RM, int RN>
int launch_cfg(const Bf3Gemm& g, int kb_per, hipStream_t s) {
    constexpr int TMB = WM * RM, TNB = WN * RN;
    const int tiles_m = g.M / (TMB * 16), tiles_n = g.N / (TNB * 16);
    const dim3 grid(tiles_m * tiles_n * g.ksplit * (g.nbatch > 1 ? 2 : 1));
    const size_t lds = (size_t)2 * (TMB + TNB) * 3 * 1024;
    // XCD-aware tile order (Bf3Map): the arrangement with the fewest strip bytes per XCD, counting every block of 32 tiles as
    // fetching its strips afresh (an L2 holds ~4 MB; a strip is K * 6 bytes * 192 or 128 rows)
    constexpr int map_on = 1;
    Bf3Map mp{1, 0, 0, 0, 0};
    if (map_on && g.ksplit == 1 && g.nbatch <= 1 && (tiles_m * tiles_n) % 256 == 0) {
        double best = 0.0;
        for (int xm = 1; xm <= 8; xm *= 2) {
            const int xn = 8 / xm;
            if (tiles_m % xm || tiles_n % xn) continue;
            const int rm = tiles_m / xm, rn = tiles_n / xn;
            for (int bm = 1; bm <= 32; bm *= 2) {
                const int bn = 32 / bm;
                if (rm % bm || rn % bn) continue;
                const double cost = (double)(rm / bm) * (rn / bn) * (bm * (double)TMB + bn * (double)TNB);
                if (mp.bm == 0 || cost < best) { best = cost; mp = Bf3Map{xn, rm, rn, bm, bn}; }
            }
        }
    }
    static bool attr_set = false;
    auto kern = &gemm_bf3_kernel<WM, WN, RM, RN, 9>;
    if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, g, tiles_m, tiles_n, kb_per, mp);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

int bf3_mode() {
    if (g_mode < 0) {
        const char* v = std::getenv("INET_GEMM_BF3");
        const int m = v ? std::atoi(v) : 9;
        g_mode = m == 0 ? 0 : 9;
    }
    return g_mode;
}
void bf3_set_mode(int m) { g_mode = m == 0 ? 0 : 9; }

bool gemm_bf3_ok(int M, int N, int K) {
    return M > 0 && M % 192 == 0 && N > 0 && N % 128 == 0 && K >= 64 && K % 32 == 0;
}

int bf3_split(const float* X, long ld, int kmajor, int R, int K, unsigned char* P, long piece_bytes, int kb_total, int rb0,
              int kb0, hipStream_t s) {
    if (!X || !P || R <= 0 || K <= 0 || R % 16 || K % 32 || (kmajor && R % 64)) return -1;
    SplitArgs a{X, ld, P, piece_bytes, kb_total, rb0, kb0, R, K, 1};
    char label[64];
    std::snprintf(label, sizeof label, "bf3_split %s R%d K%d", kmajor ? "cols" : "rows", R, K);
    ProfScope prof(PROF_HBM, 0.0, s, label, 10.0 * R * K);
    if (kmajor) hipLaunchKernelGGL(bf3_split_cols_kernel, dim3(K / 32, R / 64), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(bf3_split_rows_kernel, dim3((unsigned)(((long)R / 16 * (K / 32) * 64 + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Up to 8 strided row splits of one shape in ONE launch (blockIdx.y = job): the three gates of up to two directions' W_hh for the
// big-batch step kernels (six 4 us launches in front of every layer otherwise).
struct SplitBatch { SplitArgs j[8]; int n; };
__global__ __launch_bounds__(256) void bf3_split_rows_batch_kernel(SplitBatch b) {
    if ((int)blockIdx.y >= b.n) return;
    const SplitArgs& a = b.j[blockIdx.y];
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = id & 63;
    const long frag = id >> 6;
    const int KB = a.K / 32;
    const long rb = frag / KB; const int kb = (int)(frag - rb * KB);
    if (rb >= a.R / 16) return;
    const long off = (rb * 16 + (lane & 15)) * a.ld + kb * 32 + (lane >> 4) * 8;
    const f32x4 v0 = ld4u(a.X + off), v1 = ld4u(a.X + off + 4);
    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    split8_store(v, a.P + (((a.rb0 + rb * a.rb_mul) * a.kb_total + a.kb0 + kb) * 64 + lane) * 16, a.piece_bytes);
}
int bf3_split_strided_batch(const Bf3SplitJob* jobs, int n, int R, int K, long ld, long piece_bytes, int kb_total, hipStream_t s) {
    if (n < 1 || n > 8 || R <= 0 || K <= 0 || R % 16 || K % 32) return -1;
    SplitBatch b{};
    b.n = n;
    for (int i = 0; i < n; ++i) {
        if (!jobs[i].X || !jobs[i].P || jobs[i].rb_mul < 1) return -1;
        b.j[i] = SplitArgs{jobs[i].X, ld, jobs[i].P, piece_bytes, kb_total, jobs[i].rb0, 0, R, K, jobs[i].rb_mul};
    }
    hipLaunchKernelGGL(bf3_split_rows_batch_kernel, dim3((unsigned)(((long)R / 16 * (K / 32) * 64 + 255) / 256), n), dim3(256), 0, s, b);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int bf3_split_strided(const float* X, long ld, int R, int K, unsigned char* P, long piece_bytes, int kb_total, int rb0, int rb_mul,
                      hipStream_t s) {
    if (!X || !P || R <= 0 || K <= 0 || R % 16 || K % 32 || rb_mul < 1) return -1;
    SplitArgs a{X, ld, P, piece_bytes, kb_total, rb0, 0, R, K, rb_mul};
    hipLaunchKernelGGL(bf3_split_rows_kernel, dim3((unsigned)(((long)R / 16 * (K / 32) * 64 + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_gemm_bf3(const Bf3Gemm& gin, hipStream_t s) {
    Bf3Gemm g = gin;
    if (!gemm_bf3_ok(g.M, g.N, g.K) || !g.A || !g.B || !g.C) return -1;
    if (g.nbatch > 1 && (!g.A2 || !g.B2 || !g.C2)) return -1;
    if (g.a_alt_from && (!g.A_alt || (g.nbatch > 1 && !g.A2_alt))) return -1;
    const int nbt = g.nbatch > 1 ? 2 : 1;
    // accumulations queued on different side streams into one tensor (a module applied twice in a step) are ordered, as in launch_gemm
    if (g.acc == ACC_ADD && side_is(s)) {
        if (side_order_dest(g.C, s) != 0 || (nbt > 1 && side_order_dest(g.C2, s) != 0)) return -2;
    }
    const bool wide = g.N % 192 == 0;                          // 192 x 192 tiles, else 192 x 128
    const int tiles = (g.M / 192) * (g.N / (wide ? 192 : 128)) * nbt;
    const int KB = g.K / 32;
    if (g.ksplit <= 0) {                                       // fill the 256 CUs once; a split keeps >= 16 k blocks
        int sp = 1;
        while (tiles * sp * 2 <= 256 && KB % (sp * 2) == 0 && KB / (sp * 2) >= 16) sp *= 2;
        g.ksplit = sp;
    }
    if (KB % g.ksplit) return -1;
    if (g.ksplit > 1) {
        if (g.epi != EPI_NONE) return -1;
        if (g.acc == ACC_STORE) {
            if (hipMemset2DAsync(g.C, g.ldc * sizeof(float), 0, (size_t)g.N * sizeof(float), g.M, s) != hipSuccess) return -2;
            if (nbt > 1 && hipMemset2DAsync(g.C2, g.ldc * sizeof(float), 0, (size_t)g.N * sizeof(float), g.M, s) != hipSuccess) return -2;
        }
    }
    char label[96];
    std::snprintf(label, sizeof label, "M%d N%d K%d bf3p%d t192x%d s%d e%d%s", g.M, g.N, g.K, bf3_mode(), wide ? 192 : 128,
                  g.ksplit, g.epi, nbt > 1 ? " x2" : "");
    // algorithmic bytes: the piece operands (6 bytes per element) read once, the result written once
    ProfScope prof(PROF_GEMM, 2.0 * g.M * g.N * g.K * nbt, s, label,
                   nbt * (6.0 * ((double)g.M * g.K + (double)g.N * g.K) + 4.0 * (double)g.M * g.N));
    return wide ? launch_cfg<2, 4, 6, 3>(g, KB / g.ksplit, s) : launch_cfg<4, 2, 3, 4>(g, KB / g.ksplit, s);
}
