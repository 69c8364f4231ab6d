// First-touch of every kernel of the library, once per device, BEFORE any step is timed.
//
// The HIP runtime loads a translation unit's code object on a device when the first kernel of that unit is launched there, and
// builds each kernel's function object at ITS first launch (hip::Function::getStatFunc): host work on the launch path -- a code
// object is mapped, relocated and copied to the device; a function object needs a symbol lookup and the kernel's metadata.  For
// this library that is 17 code objects and ~250 kernels, and which of them a step meets for the first time depends on the
// branch it takes: the first FREE-RUNNING training step of a process (the teacher-forcing coin's other side: decoder.py:432)
// launches the decode chain, its BPTT and their gradient products for the first time.  Round 5 measured that first pass at +1.3 ms
// of host time typically and at 11 and 16 ms in two of ~25 fresh processes; on the empty queue the benchmark's fence leaves
// behind, that is GPU idle time inside the timed region, and bench.py ran two un-timed "priming" steps to keep it out
// (moved, not removed).  inet_preload() removes it: hipFuncGetAttributes on a kernel's host handle does
// exactly the lazy part of a first launch (code object + function object for the current device) without launching anything.
//
// The handles: hipcc emits, per translation unit, a constructor that hands every kernel's host-side handle to the runtime's
// __hipRegisterFunction.  The library defines that symbol itself (it is linked -Bsymbolic-functions, so its own constructors
// bind here), files the handle and passes the call on to the runtime's definition: a complete list, template instantiations
// included, with nothing to keep in step by hand.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace {
using RegisterFn = void (*)(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*);

std::vector<const void*>& handles() {
    static std::vector<const void*>* v = new std::vector<const void*>();      // (alive during static constructors AND destructors)
    return *v;
}
RegisterFn runtime_register() {
    static RegisterFn fn = [] {
        void* p = dlsym(RTLD_NEXT, "__hipRegisterFunction");
        if (!p) {
            if (void* h = dlopen("libamdhip64.so", RTLD_NOLOAD | RTLD_LAZY)) p = dlsym(h, "__hipRegisterFunction");
        }
        if (!p) {
            std::fprintf(stderr, "libinpaintnet_hip: the HIP runtime's __hipRegisterFunction was not found: %s\n", dlerror());
            std::abort();                                      // no kernel of the library could ever be launched
        }
        return reinterpret_cast<RegisterFn>(p);
    }();
    return fn;
}
}  // namespace

extern "C" __attribute__((visibility("default"))) void __hipRegisterFunction(void** modules, const void* hostFunction, char* deviceFunction,
                                                                             const char* deviceName, unsigned threadLimit, void* tid,
                                                                             void* bid, void* blockDim, void* gridDim, int* wSize) {
    handles().push_back(hostFunction);
    runtime_register()(modules, hostFunction, deviceFunction, deviceName, threadLimit, tid, bid, blockDim, gridDim, wSize);
}

int preload_kernels() {
    static std::mutex mu;
    static bool done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -2;        // no device: nothing to load kernels on
    std::lock_guard<std::mutex> lock(mu);
    if (done[dev]) return 0;
    int n = 0;
    for (const void* h : handles()) {
        hipFuncAttributes at;
        if (hipFuncGetAttributes(&at, h) != hipSuccess) { (void)hipGetLastError(); return -2; }
        ++n;
    }
    done[dev] = true;
    return n;
}
int preload_kernel_count() { return (int)handles().size(); }
