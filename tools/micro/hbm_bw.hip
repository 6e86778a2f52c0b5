// Measured HBM ceilings on one MI355X: read-only stream (sum reduction), device-to-device copy kernel, hipMemcpyDtoD.
// 1 GiB buffers (beyond the 256 MB Infinity Cache), float4 per lane, grid-stride with 8 loads in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_read(const f32x4* __restrict__ a, size_t n, float* out) {
    f32x4 s = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = a[i + j * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; i < n; i += stride) s += a[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_copy(const f32x4* __restrict__ a, f32x4* __restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = a[i + j * stride];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[i + j * stride] = v[j];
    }
    for (; i < n; i += stride) b[i] = a[i];
}
int main() {
    const size_t bytes = 1ull << 30, n = bytes / 16;
    f32x4 *a, *b; float* out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto fn, const char* name, double moved) {
        fn(); hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 10; ++r) { hipEventRecord(e0); fn(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
        printf("%-28s %8.3f ms  %6.2f TB/s\n", name, best, moved / best / 1e9);
    };
    for (int wgs : {1024, 2048, 4096, 8192})
        time([&] { hipLaunchKernelGGL(k_read, dim3(wgs), dim3(256), 0, 0, a, n, out); }, (std::string("read, ") + std::to_string(wgs) + " WGs").c_str(), (double)bytes);
    for (int wgs : {2048, 8192})
        time([&] { hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, 0, a, b, n); }, (std::string("copy kernel (r+w), ") + std::to_string(wgs)).c_str(), 2.0 * bytes);
    time([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, "hipMemcpy DtoD (r+w)", 2.0 * bytes);
    return 0;
}
