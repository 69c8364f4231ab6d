// Feasibility: a long-running kernel raises a counter in signal memory half-way; a second stream waits for the value with
// hipStreamWaitValue32 and then runs a consumer kernel that reads what the first kernel wrote (write-through stores) before the signal.
//   hipcc --offload-arch=gfx950 -O3 tools/exp_waitvalue.hip -o build/exp_waitvalue
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void producer(float* data, unsigned* sig, unsigned long long* stamps, int n, int spin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // first half: written with write-through stores, then the signal
    __hip_atomic_store(data + i, 1.0f + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(sig, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (blockIdx.x == 0) stamps[0] = wall_clock64();
    }
    // second half: a long tail
    float x = i;
    for (int k = 0; k < spin; ++k) x = x * 1.0000001f + 1e-7f;
    data[n + i] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[1] = wall_clock64();
}
__global__ void consumer(const float* data, float* out, unsigned long long* stamps, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    out[i] = data[i];
    if (i == 0) stamps[2] = wall_clock64();
}
int main() {
    const int nb = 256, n = nb * 256;
    float *data, *out; unsigned* sig; unsigned long long* stamps;
    CK(hipMalloc(&data, 2 * n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMemset(data, 0, 2 * n * 4)); CK(hipMemset(out, 0, n * 4));
    CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory));
    CK(hipMemset(sig, 0, 8));
    CK(hipHostMalloc(&stamps, 64)); stamps[0] = stamps[1] = stamps[2] = 0;
    hipStream_t a, b; CK(hipStreamCreate(&a)); CK(hipStreamCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(out, 0, n * 4, b));
        CK(hipStreamSynchronize(b));
        const unsigned target = (rep + 1) * nb;
        hipLaunchKernelGGL(producer, dim3(nb), dim3(256), 0, a, data, sig, stamps, n, 400000);
        CK(hipStreamWaitValue32(b, sig, target, hipStreamWaitValueGte, 0xffffffffu));
        hipLaunchKernelGGL(consumer, dim3(nb), dim3(256), 0, b, data, out, stamps, n);
        CK(hipDeviceSynchronize());
        float h[4]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        float last; CK(hipMemcpy(&last, out + n - 1, 4, hipMemcpyDeviceToHost));
        printf("rep %d: signal at %.1f us, consumer at %.1f us, producer end at %.1f us (from signal) | out[0..1] = %.1f %.1f  out[n-1] = %.1f (want %.1f)\n",
               rep, 0.0, (double)(stamps[2] - stamps[0]) / 100.0, (double)(stamps[1] - stamps[0]) / 100.0, h[0], h[1], last, 1.0f + (n - 1));
    }
    return 0;
}
