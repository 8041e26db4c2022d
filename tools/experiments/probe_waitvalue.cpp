// does this device take hipStreamWaitValue32 (signal memory)?  build: hipcc -o probe tools/probe_waitvalue.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void bump(unsigned* p) { atomicAdd(p, 1u); }
int main()
{
    int can = -1; hipError_t e = hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("attr err %d can %d\n", (int)e, can);
    unsigned* sig = nullptr; e = hipExtMallocWithFlags((void**)&sig, 256, hipMallocSignalMemory);
    printf("signal alloc err %d ptr %p\n", (int)e, (void*)sig);
    unsigned* plain = nullptr; e = hipMalloc((void**)&plain, 256);
    hipStream_t a, b; hipStreamCreate(&a); hipStreamCreate(&b);
    for (int which = 0; which < 2; which++) {
        unsigned* p = which == 0 ? sig : plain;
        if (!p) continue;
        hipMemset(p, 0, 256);
        e = hipStreamWaitValue32(b, p, 1, hipStreamWaitValueGte, 0xFFFFFFFFu);
        printf("%s: wait err %d\n", which == 0 ? "signal" : "plain", (int)e);
        if (e != hipSuccess) { (void)hipGetLastError(); continue; }
        hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, a, p);
        e = hipStreamSynchronize(b);
        printf("%s: sync after bump err %d\n", which == 0 ? "signal" : "plain", (int)e);
    }
    return 0;
}
