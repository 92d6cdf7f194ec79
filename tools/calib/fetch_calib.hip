// Calibration of rocprofv3 FETCH_SIZE on gfx950 for THIS repo's access widths (MI355X_MICROARCH.md
// "HBM": FETCH_SIZE reads 1/2 of the bytes of a 16 B/lane stream; other widths must be calibrated).
// Streams a 2 GiB buffer (far beyond the 256 MiB Infinity Cache) once per kernel with 4, 8 and
// 16 bytes per lane; compare FETCH_SIZE per dispatch with the known 2 GiB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <typename T>
__global__ void k_stream(const T* __restrict__ p, size_t n, unsigned long long* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; i < n; i += stride) {
        T v = p[i];
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc += w[k];
    }
    if (acc == 0x123456789ull) *out = acc;
}
int main() {
    const size_t bytes = 2ull << 30;
    void* buf; unsigned long long* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_stream<uint32_t>, dim3(2048), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 4, out);
        hipLaunchKernelGGL(k_stream<uint2>, dim3(2048), dim3(256), 0, 0, (const uint2*)buf, bytes / 8, out);
        hipLaunchKernelGGL(k_stream<uint4>, dim3(2048), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
    }
    hipDeviceSynchronize();
    printf("streamed %zu bytes per kernel\n", bytes);
    return 0;
}
