// Streaming yardsticks with hand-written kernels (not part of the product): what does this GPU
// deliver for a pure read, a pure write and a copy, by access width, unroll and grid size?
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/probe/stream_probe tools/probe/stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void read_kernel(const f4 *__restrict__ x, size_t n4, float *sink) {
    f4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n4; i += stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = i + u * 256 < n4 ? __builtin_nontemporal_load(x + i + u * 256) : acc;
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}
template <int U>
__global__ __launch_bounds__(256) void write_kernel(f4 *__restrict__ y, size_t n4) {
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n4; i += stride)
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < n4) __builtin_nontemporal_store(v, y + i + u * 256);
}
template <int U>
__global__ __launch_bounds__(256) void copy_kernel(const f4 *__restrict__ x, f4 *__restrict__ y, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n4; i += stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < n4) v[u] = __builtin_nontemporal_load(x + i + u * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < n4) __builtin_nontemporal_store(v[u], y + i + u * 256);
    }
}
template <class F> static float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) f();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main(int argc, char **argv) {
    const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 4.0) * (1ull << 30);
    const size_t n4 = bytes / 16;
    f4 *x, *y; float *sink;
    CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(x, 0, bytes)); CK(hipMemset(y, 0, bytes));
    const double gb = bytes / 1e9;
    for (int wg : {1024, 2048, 4096, 8192, 16384, 65536}) {
        float r1 = timeit([&] { hipLaunchKernelGGL(read_kernel<1>, dim3(wg), dim3(256), 0, 0, x, n4, sink); }, 5);
        float r4 = timeit([&] { hipLaunchKernelGGL(read_kernel<4>, dim3(wg), dim3(256), 0, 0, x, n4, sink); }, 5);
        float r8 = timeit([&] { hipLaunchKernelGGL(read_kernel<8>, dim3(wg), dim3(256), 0, 0, x, n4, sink); }, 5);
        float w1 = timeit([&] { hipLaunchKernelGGL(write_kernel<1>, dim3(wg), dim3(256), 0, 0, y, n4); }, 5);
        float w4 = timeit([&] { hipLaunchKernelGGL(write_kernel<4>, dim3(wg), dim3(256), 0, 0, y, n4); }, 5);
        float c4 = timeit([&] { hipLaunchKernelGGL(copy_kernel<4>, dim3(wg), dim3(256), 0, 0, x, y, n4); }, 5);
        float c8 = timeit([&] { hipLaunchKernelGGL(copy_kernel<8>, dim3(wg), dim3(256), 0, 0, x, y, n4); }, 5);
        printf("blocks %6d  read U1 %.2f U4 %.2f U8 %.2f TB/s | write U1 %.2f U4 %.2f | copy(r+w) U4 %.2f U8 %.2f\n", wg,
               gb / r1, gb / r4, gb / r8, gb / w1, gb / w4, 2 * gb / c4, 2 * gb / c8);
        fflush(stdout);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
