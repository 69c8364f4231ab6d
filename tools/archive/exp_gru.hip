// Standalone experiment: where does the time of one fused GRU step go?  (hipcc --offload-arch=gfx950 -O3)
// Variants of the forward step at B x H: full / no epilogue / loads only / MFMA only / empty.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f4u_t { float x, y, z, w; };
__device__ __forceinline__ f32x4 ld4u(const float* p) { f4u_t v = *reinterpret_cast<const f4u_t*>(p); return f32x4{v.x, v.y, v.z, v.w}; }
#ifdef USE_LD4U
#define LD(p) ld4u(p)
#elif defined(USE_NT)
#define LD(p) __builtin_nontemporal_load((const f32x4*)(p))
#else
#define LD(p) (*(const f32x4*)(p))
#endif
#ifndef NW
#define NW 4
#endif
#ifdef USE_CLAMP
#define ROW(r) min((r), B - 1)
#else
#define ROW(r) (r)
#endif

template <int VAR>
__global__ __launch_bounds__(64 * NW) void step(const float* __restrict__ hp, const float* __restrict__ W,
                                            const float* __restrict__ gi, const float* __restrict__ bh,
                                            float* __restrict__ hn, int B, int H, float* __restrict__ sv) {
    __shared__ float red[NW * 4 * 512];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int j0 = blockIdx.x * 16, row0 = blockIdx.y * 32;
    if (VAR == 4) { if (t == 0 && hp == nullptr) hn[0] = 0.f; return; }
    f32x4 acc[2][3];
    for (int a = 0; a < 2; ++a) for (int g = 0; g < 3; ++g) acc[a][g] = f32x4{0, 0, 0, 0};
    const int Sq = (H / 16) / NW;
    const int s0 = w * Sq;
    constexpr int ND = 32 / NW;
    f32x4 fa[ND][2], fb[ND][3];
    if (VAR != 3) {
#pragma unroll
        for (int d = 0; d < ND; ++d) {
#ifdef LINE128
            const int k = 32 * ((s0 + d) >> 1) + 8 * q + 4 * ((s0 + d) & 1);   // 4 lanes of a row cover one 128-B line per pair of steps
#else
            const int k = 16 * (s0 + d) + 4 * q;
#endif
#ifdef PACKED
            // fragment-major operands: block (row/16, k-step) is 1 KB contiguous in lane order
#pragma unroll
            for (int ms = 0; ms < 2; ++ms) fa[d][ms] = LD(hp + ((long)((row0 >> 4) + ms) * (H / 16) + s0 + d) * 256 + 4 * lane);
#pragma unroll
            for (int g = 0; g < 3; ++g) fb[d][g] = LD(W + ((long)((g * H + j0) >> 4) * (H / 16) + s0 + d) * 256 + 4 * lane);
            (void)k;
#else
#pragma unroll
            for (int ms = 0; ms < 2; ++ms) fa[d][ms] = LD(hp + (long)ROW(row0 + 16 * ms + i16) * H + k);
#pragma unroll
            for (int g = 0; g < 3; ++g) fb[d][g] = LD(W + (long)(g * H + j0 + i16) * H + k);
#endif
        }
    } else {
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            for (int ms = 0; ms < 2; ++ms) fa[d][ms] = f32x4{1.f * lane, 2.f, 3.f, 4.f};
            for (int g = 0; g < 3; ++g) fb[d][g] = f32x4{1.f, 2.f * d, 3.f, 4.f};
        }
    }
    if (VAR != 2) {
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        acc[ms][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[d][ms][e], fb[d][g][e], acc[ms][g], 0, 0, 0);
    } else {
#pragma unroll
        for (int d = 0; d < ND; ++d)
            for (int ms = 0; ms < 2; ++ms)
                for (int g = 0; g < 3; ++g) acc[ms][g] += fa[d][ms] + fb[d][g];
    }
    for (int ms = 0; ms < 2; ++ms)
        for (int a = 0; a < 3; ++a)
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ms + 4 * (lane >> 4) + r;
                red[(w * 4 + a) * 512 + row * 16 + (lane & 15)] = acc[ms][a][r];
            }
    __syncthreads();
    for (int p = 0; p < 512 / (64 * NW); ++p) {
        const int pos = t + 64 * NW * p;
        const int b = row0 + (pos >> 4), j = j0 + (pos & 15);
        float v[3];
        for (int a = 0; a < 3; ++a) {
            float s = 0;
            for (int ww = 0; ww < NW; ++ww) s += red[(ww * 4 + a) * 512 + pos];
            v[a] = s;
        }
        if (VAR == 0 || VAR >= 5) {
            const float* gp = gi + (long)b * 3 * H;
            float gr = v[0] + gp[j] + bh[j], gz = v[1] + gp[H + j] + bh[H + j], gn = gp[2 * H + j];
            float r = 1.f / (1.f + expf(-gr)), z = 1.f / (1.f + expf(-gz));
            float n = tanhf(gn + r * (v[2] + bh[2 * H + j]));
#ifdef PACKED
            const long po = ((long)(b >> 4) * (H / 16) + (j >> 4)) * 256 + (((j & 15) >> 2) * 16 + (b & 15)) * 4 + (j & 3);
            const float hpv = hp[po];
            hn[po] = (1.f - z) * n + z * hpv;
#else
            const float hpv = hp[(long)b * H + j];
            hn[(long)b * H + j] = (1.f - z) * n + z * hpv;
#endif
            const long o = (long)b * H + j, BH = (long)B * H;
            if (VAR == 5) { sv[o] = r; sv[BH + o] = z; sv[2 * BH + o] = n; sv[3 * BH + o] = v[2]; sv[4 * BH + o] = hpv; }
            if (VAR == 6) { *(f32x4*)(sv + 4 * o) = f32x4{r, z, n, v[2]}; sv[4 * BH + o] = hpv; }
            if (VAR == 7) { *(f32x4*)(sv + 4 * o) = f32x4{r, z, n, v[2]}; }
        } else {
            hn[(long)b * H + j] = v[0] + v[1] + v[2];
        }
    }
}

template <int VAR>
float run(const float* hp, const float* W, const float* gi, const float* bh, float* h2, int B, int H, int iters, float* sv) {
    dim3 grid(H / 16, B / 32);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(step<VAR>, grid, dim3(64 * NW), 0, 0, hp, W, gi, bh, h2, B, H, sv);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(step<VAR>, grid, dim3(64 * NW), 0, 0, hp, W, gi, bh, h2, B, H, sv);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}

// realistic chain: h ping-pongs between launches (produced on other XCDs), gi and saves are fresh slices per step
float run_chain(float* hA, float* hB, const float* W, const float* gi_all, const float* bh, int B, int H, int T, float* sv_all, int iters) {
    dim3 grid(H / 16, B / 32);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0);
    for (int it = 0; it < iters; ++it)
        for (int t = 0; t < T; ++t) {
            float* src = (t & 1) ? hB : hA; float* dst = (t & 1) ? hA : hB;
            hipLaunchKernelGGL(step<5>, grid, dim3(64 * NW), 0, 0, src, W, gi_all + (size_t)t * B * 3 * H, bh, dst, B, H,
                               sv_all + (size_t)t * 5 * B * H);
        }
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / (iters * T);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 512;
    float *hp, *W, *gi, *bh, *h2;
    hipMalloc(&hp, (size_t)B * H * 4); hipMalloc(&W, (size_t)3 * H * H * 4); hipMalloc(&gi, (size_t)B * 3 * H * 4);
    hipMalloc(&bh, 3 * H * 4); hipMalloc(&h2, (size_t)B * H * 4);
    float* sv; hipMalloc(&sv, (size_t)5 * B * H * 4);
    std::vector<float> tmp((size_t)3 * H * H > (size_t)B * 3 * H ? (size_t)3 * H * H : (size_t)B * 3 * H);
    for (auto& x : tmp) x = (rand() % 2001 - 1000) * 1e-4f;
    hipMemcpy(hp, tmp.data(), (size_t)B * H * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, tmp.data(), (size_t)3 * H * H * 4, hipMemcpyHostToDevice);
    hipMemcpy(gi, tmp.data(), (size_t)B * 3 * H * 4, hipMemcpyHostToDevice);
    hipMemcpy(bh, tmp.data(), 3 * H * 4, hipMemcpyHostToDevice);
    const int it = 200;
    printf("B=%d H=%d  (us per launch, %d back-to-back launches)\n", B, H, it);
    printf("  full           %7.2f\n", run<0>(hp, W, gi, bh, h2, B, H, it, sv));
    printf("  no epilogue    %7.2f\n", run<1>(hp, W, gi, bh, h2, B, H, it, sv));
    printf("  loads only     %7.2f\n", run<2>(hp, W, gi, bh, h2, B, H, it, sv));
    printf("  mfma only      %7.2f\n", run<3>(hp, W, gi, bh, h2, B, H, it, sv));
    printf("  full+5 saves   %7.2f\n", run<5>(hp, W, gi, bh, h2, B, H, it, sv));
    printf("  full+packed sv %7.2f\n", run<6>(hp, W, gi, bh, h2, B, H, it, sv));
    printf("  full+packed4   %7.2f\n", run<7>(hp, W, gi, bh, h2, B, H, it, sv));
    {
        const int T = 24;
        float *giT, *svT, *hB2;
        hipMalloc(&giT, (size_t)T * B * 3 * H * 4); hipMalloc(&svT, (size_t)T * 5 * B * H * 4); hipMalloc(&hB2, (size_t)B * H * 4);
        hipMemset(giT, 0, (size_t)T * B * 3 * H * 4); hipMemset(hB2, 0, (size_t)B * H * 4);
        run_chain(hp, hB2, W, giT, bh, B, H, T, svT, 2);
        printf("  chain (cold h, fresh gi/saves per step) %7.2f\n", run_chain(hp, hB2, W, giT, bh, B, H, T, svT, 10));
    }
    printf("  empty kernel   %7.2f\n", run<4>(hp, W, gi, bh, h2, B, H, it, sv));
    return 0;
}
