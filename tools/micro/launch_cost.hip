// What a stream operation costs the caller who waits for the whole sequence (the single-query path of
// dint_and_queries is a handful of dependent launches and copies): N dependent tiny kernels, a small pinned
// host-to-device copy in front, a device-to-host copy behind — against a kernel that reads / writes the pinned
// memory itself.   hipcc -O2 --offload-arch=gfx950 -o launch_cost launch_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <functional>

__global__ void touch(uint32_t* p) { if (threadIdx.x == 0) atomicAdd(p, 1u); }
__global__ void from_host(const uint32_t* h, uint32_t* d, int n) { for (int i = threadIdx.x; i < n; i += blockDim.x) d[i] = h[i]; }
__global__ void to_host(const uint32_t* d, uint32_t* h, int n) { for (int i = threadIdx.x; i < n; i += blockDim.x) h[i] = d[i]; }

static double time_us(const std::function<void()>& f, int reps = 2000) {
    for (int i = 0; i < 50; ++i) f();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) f();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}

int main() {
    hipStream_t s;
    hipStreamCreate(&s);
    uint32_t *d, *h, *hd;
    hipMalloc(&d, 1 << 20);
    hipMemset(d, 0, 1 << 20);
    hipHostMalloc(&h, 1 << 20, hipHostMallocDefault);
    hipHostGetDevicePointer(reinterpret_cast<void**>(&hd), h, 0);
    for (int n : {1, 2, 4, 6, 8, 12})
        std::printf("%2d dependent kernels + sync: %.1f us\n", n, time_us([&] {
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, d);
            hipStreamSynchronize(s);
        }));
    std::printf("H2D 4 KB + sync: %.1f us\n", time_us([&] { hipMemcpyAsync(d, h, 4096, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); }));
    std::printf("D2H 8 B + sync: %.1f us\n", time_us([&] { hipMemcpyAsync(h, d, 8, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }));
    std::printf("H2D 4 KB + 4 kernels + D2H 8 B + sync: %.1f us\n", time_us([&] {
        hipMemcpyAsync(d, h, 4096, hipMemcpyHostToDevice, s);
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, d);
        hipMemcpyAsync(h, d, 8, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
    }));
    std::printf("H2D 4 KB + 4 kernels (last writes pinned) + sync: %.1f us\n", time_us([&] {
        hipMemcpyAsync(d, h, 4096, hipMemcpyHostToDevice, s);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, d);
        hipLaunchKernelGGL(to_host, dim3(1), dim3(64), 0, s, d, hd, 2);
        hipStreamSynchronize(s);
    }));
    std::printf("kernel reading 4 KB pinned + 3 kernels (last writes pinned) + sync: %.1f us\n", time_us([&] {
        hipLaunchKernelGGL(from_host, dim3(1), dim3(1024), 0, s, hd, d, 1024);
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, d);
        hipLaunchKernelGGL(to_host, dim3(1), dim3(64), 0, s, d, hd, 2);
        hipStreamSynchronize(s);
    }));
    std::printf("the null stream instead, 4 kernels + sync: %.1f us\n", time_us([&] {
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, nullptr, d);
        hipStreamSynchronize(nullptr);
    }));
    return 0;
}
