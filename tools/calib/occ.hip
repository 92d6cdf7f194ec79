// Calibration: workgroups per CU the runtime reports for a 64-thread kernel as a function of its LDS size (gfx950's LDS
// allocation granularity decides whether shaving a few hundred bytes off k_uscore's 6.2 KB buys a 25th wave).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) k(float* p) { extern __shared__ float s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[63 - threadIdx.x]; }
int main() {
    for (int lds : {4096, 5120, 5376, 5440, 5632, 5888, 6144, 6224, 6400, 6656, 6912, 7168, 7760, 8192}) {
        int n = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, lds);
        printf("LDS %5d B -> %d workgroups per CU (%s)\n", lds, n, hipGetErrorString(e));
    }
    return 0;
}
