// Does a wave64 VALU instruction cost less when only the low 16 / 32 lanes are active?  (one wave per SIMD, dependent FMA chains)
// hipcc --offload-arch=gfx950 -O3 -o exec_skip exec_skip.hip && ./exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, int active, int iters) {
    const int lane = threadIdx.x & 63;
    float a = lane * 0.001f, b = 1.0001f, c = 0.5f, d = 0.25f;
    if (lane < active) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) { a = fmaf(a, b, c); d = fmaf(d, b, a); c = fmaf(c, b, d); b = fmaf(b, 1.0000001f, 1e-9f); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
int main() {
    float* o; hipMalloc(&o, 1024 * 256 * 4 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves_per_simd : {1, 4}) {
        for (int active : {64, 32, 16, 8, 1}) {
            const int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = one per SIMD of a CU
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, o, active, 1000);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, o, active, 20000);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("waves/SIMD %d  active lanes %2d : %.3f ms\n", waves_per_simd, active, ms);
        }
    }
    return 0;
}
