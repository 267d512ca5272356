// Microtest (development aid): does a raw-buffer dwordx4 store that straddles num_records write its
// in-range dwords (per-dword range check) or nothing? And what does a straddling x4 load return?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned* buf, unsigned n_in_range, unsigned* loaded) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, int(n_in_range * 4), 0x00020000);
    const unsigned lane = threadIdx.x;
    u32x4 v = {100 + 4 * lane, 101 + 4 * lane, 102 + 4 * lane, 103 + 4 * lane};
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, 16 * lane, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * lane, 0, 0);
    loaded[4 * lane + 0] = r.x; loaded[4 * lane + 1] = r.y; loaded[4 * lane + 2] = r.z; loaded[4 * lane + 3] = r.w;
}
int main() {
    unsigned *d, *l; hipMalloc(&d, 4096); hipMalloc(&l, 4096);
    for (unsigned n : {10u, 13u, 7u}) {
        hipMemset(d, 0xFF, 4096); hipMemset(l, 0, 4096);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n, l);
        unsigned h[24], hl[24]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hl, l, sizeof hl, hipMemcpyDeviceToHost);
        printf("num_records = %u dwords\n  memory :", n); for (int i = 0; i < 20; ++i) printf(" %d", int(h[i])); 
        printf("\n  loaded :"); for (int i = 0; i < 20; ++i) printf(" %d", int(hl[i])); printf("\n");
    }
    return 0;
}
