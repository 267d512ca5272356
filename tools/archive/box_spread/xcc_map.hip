// Which XCD does workgroup i of a launch land on? The decode kernels hand each XCD a contiguous part of the work by
// blockIdx % 8 (round-robin placement); this prints the hardware's XCC_ID per workgroup for a launch shaped like
// theirs (1024 threads, 160 KB of LDS: one workgroup per CU, 256 workgroups).
// hipcc -O2 --offload-arch=gfx950 -o xcc_map xcc_map.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void where(uint32_t* xcc, uint32_t* hwid) {
    extern __shared__ uint32_t lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t x, h;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
        xcc[blockIdx.x] = x + (lds[5] - 5);
        hwid[blockIdx.x] = h;
    }
}

int main() {
    const int grid = 256;
    uint32_t *d_x, *d_h;
    hipMalloc(&d_x, grid * 4);
    hipMalloc(&d_h, grid * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(where), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<uint32_t> x(grid), h(grid);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(where, dim3(grid), dim3(1024), 160 * 1024 - 64, nullptr, d_x, d_h);
        hipDeviceSynchronize();
        hipMemcpy(x.data(), d_x, grid * 4, hipMemcpyDeviceToHost);
        hipMemcpy(h.data(), d_h, grid * 4, hipMemcpyDeviceToHost);
    }
    int round_robin = 0;
    int per_xcc[16] = {0};
    for (int i = 0; i < grid; ++i) {
        round_robin += int((x[i] & 15u) == uint32_t(i % 8));
        per_xcc[x[i] & 15u]++;
    }
    std::printf("xcc of workgroups 0..23:");
    for (int i = 0; i < 24; ++i) std::printf(" %u", x[i] & 15u);
    std::printf("\nworkgroups with xcc == blockIdx %% 8: %d of %d; per xcc:", round_robin, grid);
    for (int i = 0; i < 8; ++i) std::printf(" %d", per_xcc[i]);
    std::printf("\n");
    return 0;
}
