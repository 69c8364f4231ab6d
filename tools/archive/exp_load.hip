// Standalone experiment: L2 -> CU load bandwidth of the GRU step's operand traffic under different lane->address
// patterns (hipcc --offload-arch=gfx950 -O3 tools/exp_load.hip -o exp_load).
// Each 256-thread workgroup reads what one fused step tile reads: 64 rows of h_prev and 48 rows of W_hh, K = 512 fp32
// (224 KB).  grid = (32 column tiles, NY row tiles); row tiles share W, column tiles share h_prev.
//   P0 fragment : the MFMA-fragment shape the step kernels use (per wave instruction: 16 rows x 64 B)
//   P1 quarter  : the wave keeps its K quarter but reads it row-contiguously (2 rows x 512 B per instruction)
//   P2 linear   : whole rows, 1 KB contiguous per wave instruction
//   P3 lds      : P2 through global_load_lds_dwordx4 (no VGPR return)
//   P4 lds-q    : P1 through global_load_lds_dwordx4
// Also records which CU every workgroup ran on and its start/end wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LD(p) (*(const f32x4*)(p))
constexpr int K = 512, RA = 64, RW = 48;

struct Rec { unsigned xcc, hwid; unsigned long long t0, t1; };

template <int P>
__global__ __launch_bounds__(256) void loadk(const float* __restrict__ hp, const float* __restrict__ W, float* __restrict__ sink,
                                             Rec* __restrict__ rec) {
    __shared__ __attribute__((aligned(16))) float lds[16 * 1024];   // 64 KB staging ring for P3/P4
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int j0 = blockIdx.x * 16, row0 = blockIdx.y * RA;
    unsigned long long t0 = 0;
    if (rec && t == 0) t0 = wall_clock64();
    f32x4 s = f32x4{0, 0, 0, 0};
    if (P == 0) {
        const int i16 = lane & 15, q = lane >> 4;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const int k = 16 * (8 * w + d) + 4 * q;
#pragma unroll
            for (int ms = 0; ms < 4; ++ms) s += LD(hp + (long)(row0 + 16 * ms + i16) * K + k);
#pragma unroll
            for (int g = 0; g < 3; ++g) s += LD(W + (long)(g * 512 + j0 + i16) * K + k);
        }
    } else if (P == 1) {
        const int par = lane >> 5, c = lane & 31;
#pragma unroll
        for (int i = 0; i < RA / 2; ++i) s += LD(hp + (long)(row0 + 2 * i + par) * K + 128 * w + 4 * c);
#pragma unroll
        for (int i = 0; i < RW / 2; ++i) {
            const int r = 2 * i + par;
            s += LD(W + (long)((r >> 4) * 512 + j0 + (r & 15)) * K + 128 * w + 4 * c);
        }
    } else if (P == 2) {
#pragma unroll
        for (int i = 0; i < RA / 2; ++i) {
            const int ch = 4 * i + w;
            s += LD(hp + (long)(row0 + (ch >> 1)) * K + 256 * (ch & 1) + 4 * lane);
        }
#pragma unroll
        for (int i = 0; i < RW / 2; ++i) {
            const int ch = 4 * i + w, r = ch >> 1;
            s += LD(W + (long)((r >> 4) * 512 + j0 + (r & 15)) * K + 256 * (ch & 1) + 4 * lane);
        }
    } else if (P == 3) {
#pragma unroll
        for (int i = 0; i < RA / 2; ++i) {
            const int ch = 4 * i + w;
            __builtin_amdgcn_global_load_lds(hp + (long)(row0 + (ch >> 1)) * K + 256 * (ch & 1) + 4 * lane,
                                             (__attribute__((address_space(3))) void*)(lds + ((ch & 63) * 256)), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < RW / 2; ++i) {
            const int ch = 4 * i + w, r = ch >> 1;
            __builtin_amdgcn_global_load_lds(W + (long)((r >> 4) * 512 + j0 + (r & 15)) * K + 256 * (ch & 1) + 4 * lane,
                                             (__attribute__((address_space(3))) void*)(lds + ((ch & 63) * 256)), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        s = *(const f32x4*)(lds + 4 * t);
    } else if (P == 4) {
        const int par = lane >> 5, c = lane & 31;
#pragma unroll
        for (int i = 0; i < RA / 2; ++i)
            __builtin_amdgcn_global_load_lds(hp + (long)(row0 + 2 * i + par) * K + 128 * w + 4 * c,
                                             (__attribute__((address_space(3))) void*)(lds + (((4 * i + w) & 63) * 256)), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < RW / 2; ++i) {
            const int r = 2 * i + par;
            __builtin_amdgcn_global_load_lds(W + (long)((r >> 4) * 512 + j0 + (r & 15)) * K + 128 * w + 4 * c,
                                             (__attribute__((address_space(3))) void*)(lds + (((4 * i + w) & 63) * 256)), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        s = *(const f32x4*)(lds + 4 * t);
    }
    if (s.x + s.y + s.z + s.w == 123.456f) sink[t] = s.x;
    if (rec && t == 0) {
        Rec r;
        r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        r.t0 = t0; r.t1 = wall_clock64();
        rec[blockIdx.y * gridDim.x + blockIdx.x] = r;
    }
}

template <int P>
void run(const char* name, const float* hp, const float* W, float* sink, Rec* rec, int ny) {
    dim3 grid(32, ny);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(loadk<P>, grid, dim3(256), 0, 0, hp, W, sink, (Rec*)nullptr);
    const int iters = 200;
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(loadk<P>, grid, dim3(256), 0, 0, hp, W, sink, (Rec*)nullptr);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / iters;
    hipLaunchKernelGGL(loadk<P>, grid, dim3(256), 0, 0, hp, W, sink, rec);
    hipDeviceSynchronize();
    const int n = 32 * ny;
    std::vector<Rec> h(n);
    hipMemcpy(h.data(), rec, n * sizeof(Rec), hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu;
    unsigned long long tmin = ~0ull, tmax = 0; double dsum = 0, dmax = 0;
    for (auto& r : h) {
        per_cu[((r.xcc & 0xf) << 8) | ((r.hwid >> 8) & 0xff)]++;
        tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1);
        const double d = (double)(r.t1 - r.t0); dsum += d; dmax = std::max(dmax, d);
    }
    int mx = 0; for (auto& kv : per_cu) mx = std::max(mx, kv.second);
    // wall_clock64 ticks at 100 MHz
    printf("  %-10s wgs=%4d  %7.2f us/launch  %6.2f TB/s | CUs used %3zu, max WG/CU %d | per-WG avg %.2f us, max %.2f us, span %.2f us\n",
           name, n, us, n * 224.0 * 1024 / us * 1e-6, per_cu.size(), mx, dsum / n / 100.0, dmax / 100.0, (tmax - tmin) / 100.0);
}

int main() {
    const int NYMAX = 16;
    float *hp, *W, *sink; Rec* rec;
    hipMalloc(&hp, (size_t)NYMAX * RA * K * 4); hipMalloc(&W, (size_t)3 * 512 * K * 4); hipMalloc(&sink, 4096);
    hipMalloc(&rec, 32 * NYMAX * sizeof(Rec));
    hipMemset(hp, 0, (size_t)NYMAX * RA * K * 4); hipMemset(W, 0, (size_t)3 * 512 * K * 4);
    for (int ny : {4, 8, 16}) {
        printf("grid 32 x %d\n", ny);
        run<0>("fragment", hp, W, sink, rec, ny);
        run<1>("quarter", hp, W, sink, rec, ny);
        run<2>("linear", hp, W, sink, rec, ny);
        run<3>("lds", hp, W, sink, rec, ny);
        run<4>("lds-q", hp, W, sink, rec, ny);
    }
    return 0;
}
