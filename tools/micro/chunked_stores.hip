// Microbenchmark (development aid): the decode kernel's output pattern without the decoding — every wave
// draws 32 KB chunks of one big buffer from a shared counter and writes them with 1 KB non-temporal (or
// default) store instructions. Does the write rate depend on the TOTAL size of the buffer?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(1024) void k(unsigned* out, unsigned long long chunks, unsigned* counter, int pace) {
    const unsigned lane = threadIdx.x & 63;
    for (;;) {
        unsigned c = 0;
        if (lane == 0) c = atomicAdd(counter, 1u);
        c = __builtin_amdgcn_readfirstlane(c);
        if (c >= chunks) return;
        unsigned* base = out + size_t(c) * 8192;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 32768, 0x00020000);
        for (unsigned i = 0; i != 32; ++i) {
            u32x4 v = {c, i, lane, 7};
            for (int p = 0; p < pace; ++p) asm volatile("v_add_u32 %0, %0, 1" : "+v"(v.x));  // compute between stores
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, 16 * lane, 1024 * i, NT ? 2 : 0);
        }
    }
}
int main() {
    unsigned* d; hipMalloc(&d, size_t(6) << 30);
    unsigned* cnt; hipMalloc(&cnt, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pace : {0, 200})
    for (double gb : {0.8, 1.6, 2.4, 3.2, 4.0, 5.6}) {
        const unsigned long long chunks = (unsigned long long)(gb * 1e9 / 32768);
        for (int nt = 0; nt < 2; ++nt) {
            float best = 1e9;
            for (int r = 0; r < 4; ++r) {
                hipMemset(cnt, 0, 4);
                hipEventRecord(e0);
                if (nt) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, d, chunks, cnt, pace);
                else hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, d, chunks, cnt, pace);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("pace %3d  %.1f GB  %s  %.3f ms  %.0f GB/s\n", pace, gb, nt ? "nt     " : "default", best, chunks * 32768.0 / best / 1e6);
        }
    }
    return 0;
}
