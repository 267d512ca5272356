// Microbenchmark (development aid): what does a CU's vector-memory path take per stored byte, by store width?
// 16 waves per CU, every CU; each wave rewrites its own window (4 KB: stays in L2 — or `span` bytes: streams to HBM)
// with fully coalesced stores: mode 0 = dword (256 B per instruction), 1 = dwordx2 (512 B), 2 = dwordx4 (1 KB),
// 3 = dwordx4 non-temporal; `skew`: the window starts that many ints off a 256-byte boundary (a decoded unit's
// output starts wherever its list does). Prints bytes per CU-cycle.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned* out, unsigned iters, unsigned span_bytes, unsigned skew) {
    const unsigned lane = threadIdx.x & 63;
    const size_t wave = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    unsigned* base = out + size_t(unsigned(__builtin_amdgcn_readfirstlane(int(wave)))) * (span_bytes / 4 + 64) + skew;  // skew: ints off the 256-byte boundary
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, int(span_bytes), 0x00020000);
    const unsigned mask = span_bytes - 1;
    for (unsigned it = 0; it != iters; ++it) {
        const unsigned kb = (it * 1024u) & mask;  // this iteration's 1 KB
        if (MODE == 0) {
            for (unsigned j = 0; j != 4; ++j) __builtin_amdgcn_raw_buffer_store_b32(it + j, rs, 4 * lane + 256 * j, kb, 0);
        } else if (MODE == 1) {
            u32x2 v = {it, lane};
            for (unsigned j = 0; j != 2; ++j) __builtin_amdgcn_raw_buffer_store_b64(v, rs, 8 * lane + 512 * j, kb, 0);
        } else {
            u32x4 v = {it, lane, 2, 3};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, 16 * lane, kb, MODE == 3 ? 2 : 0);
        }
    }
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned* d; (void)hipMalloc(&d, size_t(cus) * 16 * ((1u << 20) + 256) + 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[4] = {"dword   ", "dwordx2 ", "dwordx4 ", "dwordx4 nt"};
    for (unsigned skew : {0u, 1u, 4u, 16u})
    for (unsigned span : {4096u, 1u << 20}) {
        const unsigned iters = span == 4096 ? 20000 : 1024;
        for (int mode = 0; mode != 4; ++mode) {
            float best = 1e9;
            for (int r = 0; r < 3; ++r) {
                (void)hipEventRecord(e0);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(cus), dim3(1024), 0, 0, d, iters, span, skew); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(cus), dim3(1024), 0, 0, d, iters, span, skew); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(cus), dim3(1024), 0, 0, d, iters, span, skew); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(cus), dim3(1024), 0, 0, d, iters, span, skew); break;
                }
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float m; (void)hipEventElapsedTime(&m, e0, e1); if (m < best) best = m;
            }
            const double bytes = double(cus) * 16 * iters * 1024;
            printf("skew %2u ints  window %7u B per wave  %s  %8.3f ms  %7.1f GB/s  %.2f bytes per CU-cycle (%.2f GHz)\n", skew, span, names[mode], best,
                   bytes / best / 1e6, bytes / cus / (best * 1e-3 * p.clockRate * 1e3), p.clockRate / 1e6);
        }
    }
    return 0;
}
