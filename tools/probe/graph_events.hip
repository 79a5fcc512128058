// Do HIP events recorded INSIDE a captured graph carry usable timestamps after a replay?
// (decides how bench.py times the dominant kernel while mgx_solve replays its cycle from a hipGraph)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(double* p, int n)
{
    double a = p[threadIdx.x];
    for (int i = 0; i < n; ++i) a = a * 1.0000001 + 1e-9;
    p[threadIdx.x] = a;
}
int main()
{
    double* d; CK(hipMalloc(&d, 256 * sizeof(double))); CK(hipMemset(d, 0, 256 * sizeof(double)));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b, c; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c));
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(a, st));
    hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, st, d, 200000);
    CK(hipEventRecord(b, st));
    hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, st, d, 400000);
    CK(hipEventRecord(c, st));
    hipGraph_t g; CK(hipStreamEndCapture(st, &g));
    hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        float ab = -1, bc = -1;
        hipError_t e1 = hipEventElapsedTime(&ab, a, b), e2 = hipEventElapsedTime(&bc, b, c);
        std::printf("replay %d: a->b %.3f ms (%s), b->c %.3f ms (%s)\n", rep, ab, hipGetErrorString(e1), bc, hipGetErrorString(e2));
    }
    // eager reference
    CK(hipEventRecord(a, st));
    hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, st, d, 200000);
    CK(hipEventRecord(b, st));
    hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, st, d, 400000);
    CK(hipEventRecord(c, st));
    CK(hipStreamSynchronize(st));
    float ab, bc; CK(hipEventElapsedTime(&ab, a, b)); CK(hipEventElapsedTime(&bc, b, c));
    std::printf("eager: a->b %.3f ms, b->c %.3f ms\n", ab, bc);
    return 0;
}
