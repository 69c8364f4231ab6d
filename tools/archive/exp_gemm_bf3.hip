// Experiment: a GEMM on the bf16 matrix cores at fp32 accuracy.  Both operands arrive as three exact bf16 pieces
// (x = x0 + x1 + x2) in MFMA-fragment order, [row/16][K/32][lane = (k%32)/8*16 + row%16][8 bf16]; the product is the
// sum of the nine (six) piece products accumulated in f32:  C[M][N] = A[M][K] . B[N][K]^T.
// Workgroup = 8 waves (two per SIMD), tile (WM*RM*16) x (WN*RN*16), operands staged through LDS in fragment order
// (ds_read_b128, lane-contiguous), two stages, one barrier per 32-wide k block.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_gemm_bf3.hip -o build/exp_gemm_bf3 && build/exp_gemm_bf3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#ifndef DMA_
#define DMA_ 1
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float x, __bf16& a0, __bf16& a1, __bf16& a2) {
    a0 = (__bf16)x;
    const float r1 = x - (float)a0;
    a1 = (__bf16)r1;
    a2 = (__bf16)(r1 - (float)a1);
}

// X[R][K] row-major f32 -> pieces P[3][R/16][K/32][64][8]
__global__ void split_rows_kernel(const float* X, unsigned char* P, int R, int K) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one lane of one fragment
    const int lane = id & 63;
    const long frag = id >> 6;
    const int KB = K / 32;
    const long rb = frag / KB; const int kb = frag % KB;
    if (rb >= R / 16) return;
    const float* src = X + (rb * 16 + (lane & 15)) * (long)K + kb * 32 + (lane >> 4) * 8;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 a, b, c;
        split3(j < 4 ? v0[j] : v1[j - 4], a, b, c);
        p0[j] = a; p1[j] = b; p2[j] = c;
    }
    const long pstride = (long)R * K * 2;
    unsigned char* dst = P + id * 16;
    *reinterpret_cast<bf16x8*>(dst) = p0;
    *reinterpret_cast<bf16x8*>(dst + pstride) = p1;
    *reinterpret_cast<bf16x8*>(dst + 2 * pstride) = p2;
}

struct Args { const unsigned char* A; const unsigned char* B; float* C; int M, N, K; long pa, pb; int ksplit; unsigned long long* clk; };

template <int WM, int WN, int RM, int RN, int NP>
__global__ __launch_bounds__(64 * WM * WN) void gemm_bf3_kernel(Args a) {
    constexpr int NW = WM * WN, TMB = WM * RM, TNB = WN * RN;
    constexpr int STAGE = (TMB + TNB) * 3 * 1024;
    constexpr int CH = (TMB + TNB) * 3;                        // 1 KB chunks per stage
    constexpr int CPW = (CH + NW - 1) / NW;                    // chunks per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w / WN, wn = w % WN;
    const int KB = a.K / 32, kb_per = KB / a.ksplit;
    const int tiles_n = a.N / (TNB * 16), tiles_m = a.M / (TMB * 16);
    // XCD-aware: consecutive ids go round-robin over the 8 XCDs; give each XCD a contiguous range of tiles (shared A strips)
    const int nb = gridDim.x, id = blockIdx.x;
    const int per = nb / 8;
    const int tid = (nb % 8 == 0) ? (id % 8) * per + id / 8 : id;
    const int ks = tid / (tiles_m * tiles_n), tt = tid % (tiles_m * tiles_n);
    const int tm = tt / tiles_n, tn = tt % tiles_n;
    const int kb0 = ks * kb_per;
    // chunk c of a stage: c < TMB*3: A piece c / TMB, row block c % TMB;  else B
    const unsigned char* gsrc[CPW]; int loff[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int c = w + i * NW;
        const int cc = c < CH ? c : CH - 1;                    // a short last round repeats the last chunk (same bytes)
        if (cc < TMB * 3) {
            const int p = cc / TMB, rbl = cc % TMB;
            gsrc[i] = a.A + p * a.pa + ((long)(tm * TMB + rbl) * KB + kb0) * 1024 + lane * 16;
        } else {
            const int c2 = cc - TMB * 3, p = c2 / TNB, rbl = c2 % TNB;
            gsrc[i] = a.B + p * a.pb + ((long)(tn * TNB + rbl) * KB + kb0) * 1024 + lane * 16;
        }
        loff[i] = cc * 1024 + lane * 16;
    }
    f32x4 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
#if DMA_
    auto fill = [&](int kbn, unsigned char* stage) {
#pragma unroll
        for (int i = 0; i < CPW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[i] + (long)kbn * 1024),
                                             (__attribute__((address_space(3))) void*)(stage + (loff[i] - lane * 16)), 16, 0, 0);
    };
    fill(0, smem);
    __syncthreads();
#else
    u32x4 st[CPW];
    // prologue: block 0 -> stage 0
#pragma unroll
    for (int i = 0; i < CPW; ++i) st[i] = *reinterpret_cast<const u32x4*>(gsrc[i]);
#pragma unroll
    for (int i = 0; i < CPW; ++i) *reinterpret_cast<u32x4*>(smem + loff[i]) = st[i];
    __syncthreads();
#endif
    for (int kb = 0; kb < kb_per; ++kb) {
        const unsigned char* sa = smem + (kb & 1) * STAGE;
        const unsigned char* sb = sa + TMB * 3 * 1024;
#if DMA_
        if (kb + 1 < kb_per) fill(kb + 1, smem + ((kb + 1) & 1) * STAGE);
#else
        if (kb + 1 < kb_per) {
#pragma unroll
            for (int i = 0; i < CPW; ++i) st[i] = *reinterpret_cast<const u32x4*>(gsrc[i] + (long)(kb + 1) * 1024);
        }
#endif
        bf16x8 Af[RM][3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < RM; ++i) Af[i][p] = *reinterpret_cast<const bf16x8*>(sa + (p * TMB + wm * RM + i) * 1024 + lane * 16);
#pragma unroll
        for (int pj = 0; pj < 3; ++pj) {
            bf16x8 Bf[RN];
#pragma unroll
            for (int j = 0; j < RN; ++j) Bf[j] = *reinterpret_cast<const bf16x8*>(sb + (pj * TNB + wn * RN + j) * 1024 + lane * 16);
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
#if !DMA_
        if (kb + 1 < kb_per) {
            unsigned char* sn = smem + ((kb + 1) & 1) * STAGE;
#pragma unroll
            for (int i = 0; i < CPW; ++i) *reinterpret_cast<u32x4*>(sn + loff[i]) = st[i];
        }
#endif
        __syncthreads();
    }
    if (a.clk && blockIdx.x == 7 && t == 0) { a.clk[0] = clock64() - c0; a.clk[1] = wall_clock64() - w0; }
    // epilogue: lane (c, q): rows 4q + r, column c of each 16 x 16 tile
    const int c = lane & 15, q = lane >> 4;
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const long row = (long)(tm * TMB + wm * RM + i) * 16 + 4 * q, col = (long)(tn * TNB + wn * RN + j) * 16 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#ifdef SKIP_EPI_
                if (acc[i][j][r] != 12345.678f) continue;
#endif
                if (a.ksplit > 1) atomicAdd(a.C + (row + r) * a.N + col, acc[i][j][r]);
                else a.C[(row + r) * a.N + col] = acc[i][j][r];
            }
        }
}

template <int WM, int WN, int RM, int RN, int NP>
void run(const char* name, int M, int N, int K, int ksplit, const float* dA, const float* dB, unsigned char* pA, unsigned char* pB,
         float* dC, const std::vector<float>& hA, const std::vector<float>& hB) {
    constexpr int TMB = WM * RM, TNB = WN * RN;
    if (M % (TMB * 16) || N % (TNB * 16) || (K / 32) % ksplit) { printf("%s: shape does not tile\n", name); return; }
    const int nb = (M / (TMB * 16)) * (N / (TNB * 16)) * ksplit;
    const size_t lds = (size_t)2 * (TMB + TNB) * 3 * 1024;
    auto kern = &gemm_bf3_kernel<WM, WN, RM, RN, NP>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    unsigned long long* dclk; (void)hipMalloc(&dclk, 16); (void)hipMemset(dclk, 0, 16);
    Args a{pA, pB, dC, M, N, K, (long)M * K * 2, (long)N * K * 2, ksplit, dclk};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto split = [&] {
        hipLaunchKernelGGL(split_rows_kernel, dim3((M / 16) * (K / 32) / 4), dim3(256), 0, 0, dA, pA, M, K);
        hipLaunchKernelGGL(split_rows_kernel, dim3((N / 16) * (K / 32) / 4), dim3(256), 0, 0, dB, pB, N, K);
    };
    split();
    (void)hipMemset(dC, 0, (size_t)M * N * 4);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * WM * WN), lds, 0, a);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s failed: %s\n", name, hipGetErrorString(hipGetLastError())); return; }
    std::vector<float> hC((size_t)M * N);
    (void)hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    unsigned x = 777u;
    for (int s = 0; s < 4000; ++s) {
        x = x * 1664525u + 1013904223u; const int i = (x >> 8) % M;
        x = x * 1664525u + 1013904223u; const int j = (x >> 8) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)i * K + k] * hB[(size_t)j * K + k];
        maxerr = std::max(maxerr, std::fabs(ref - hC[(size_t)i * N + j])); maxref = std::max(maxref, std::fabs(ref));
    }
    const int reps = 20;
    float ms = 0, ms_split = 0;
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * WM * WN), lds, 0, a);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) split();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms_split, e0, e1);
    const double us = ms * 1e3 / reps;
    unsigned long long hclk[2]; (void)hipMemcpy(hclk, dclk, 16, hipMemcpyDeviceToHost);
    printf("   k loop of one workgroup: %llu shader clocks in %.2f us = %.0f MHz; per k block %.0f clocks (MFMA floor %d)\n", hclk[0], hclk[1] / 100.0,
           hclk[0] / (hclk[1] / 100.0), (double)hclk[0] / (K / 32 / ksplit), RM * RN * 9 * 16 * 2);
    printf("%-28s M%d N%d K%d s%d  %4d wgs  %7.1f us  %6.1f TFLOP/s fp32-equivalent | split of both operands %6.1f us | max err %.2e of %.2e (%.2e rel)\n",
           name, M, N, K, ksplit, nb, us, 2.0 * M * N * K / us * 1e-6, ms_split * 1e3 / reps, maxerr, maxref, maxerr / maxref);
}

int main() {
    const int M = 6144, N = 1536, K = 6144;                     // allocate for the largest K used
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    unsigned x = 12345u;
    auto rnd = [&] { x = x * 1664525u + 1013904223u; return ((x >> 8) & 0xffff) / 32768.f - 1.f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd() * 0.05f;
    float *dA, *dB, *dC; unsigned char *pA, *pB;
    (void)hipMalloc(&dA, hA.size() * 4); (void)hipMalloc(&dB, hB.size() * 4); (void)hipMalloc(&dC, (size_t)M * N * 4);
    (void)hipMalloc(&pA, hA.size() * 6); (void)hipMalloc(&pB, hB.size() * 6);
    (void)hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    run<2, 4, 6, 3, 9>("long-K 192x192 w2x4 9", 6144, 1536, 4096, 1, dA, dB, pA, pB, dC, hA, hB);
    // forward shape of the encoder's layer-1 input products: 6144 x 1536 x 1024 (rows of A / B are contiguous with ld = K: use K = 1024 views)
    run<4, 2, 3, 6, 9>("fwd 192x192 w4x2 9", 6144, 1536, 1024, 1, dA, dB, pA, pB, dC, hA, hB);
    run<4, 2, 3, 6, 6>("fwd 192x192 w4x2 6", 6144, 1536, 1024, 1, dA, dB, pA, pB, dC, hA, hB);
    run<2, 4, 6, 3, 9>("fwd 192x192 w2x4 9", 6144, 1536, 1024, 1, dA, dB, pA, pB, dC, hA, hB);
    // data gradient: 6144 x 1024 x 3072
    run<4, 2, 3, 4, 9>("dgrad 192x128 w4x2 9", 6144, 1024, 3072, 1, dA, dB, pA, pB, dC, hA, hB);
    // weight gradient: 1536 x 512 x 6144, split-K 4 (one product: 128 workgroups; the library groups two per launch)
    run<4, 2, 3, 4, 9>("wgrad 192x128 w4x2 9 s4", 1536, 512, 6144, 4, dA, dB, pA, pB, dC, hA, hB);
    run<4, 2, 3, 4, 9>("wgrad 192x128 w4x2 9 s8", 1536, 512, 6144, 8, dA, dB, pA, pB, dC, hA, hB);
    run<4, 2, 3, 6, 9>("wgrad1 192x192 9 s8", 1536, 1536, 6144, 4, dA, dB, pA, pB, dC, hA, hB);
    return 0;
}
