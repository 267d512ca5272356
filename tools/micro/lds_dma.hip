// Microtest (development aid): LDS-DMA gather on gfx950 — buffer_load_dwordx4 ... lds with a per-lane source
// offset under an exec mask. Checks (1) lane l's 16 bytes land at base + 16 l, (2) masked-off lanes leave their
// cell untouched, (3) out-of-range offsets (bounds-checked by the descriptor) write zeros or nothing, and
// (4) unaligned ds_read_b64 / ds_read_u16 of the landed bytes return what is expected.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(2))) q2 { u32x2 v; };
__global__ void k(const unsigned* tab, unsigned tab_bytes, const unsigned* off, unsigned* cells, unsigned* misc) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const unsigned lane = threadIdx.x;
    for (unsigned i = lane; i < 1024; i += 64) lds[i] = 0xAAAA0000u + i;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(tab), 0, int(tab_bytes), 0x00020000);
    const unsigned o = off[lane];
    if (o != 0xFFFFFFFFu)  // masked-off lanes
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 256), 16, o, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (unsigned i = lane; i < 1024; i += 64) cells[i] = lds[i];
    // unaligned reads: 8 bytes from byte address 1024 + 16 lane + 2, and a u16 from byte 1024 + 16 lane + 6
    const char* b = reinterpret_cast<const char*>(lds) + 1024 + 16 * lane;
    const u32x2 r = reinterpret_cast<const q2*>(b + 2)->v;
    misc[3 * lane] = r.x; misc[3 * lane + 1] = r.y;
    misc[3 * lane + 2] = *reinterpret_cast<const unsigned short*>(b + 6);
}
int main() {
    const unsigned rows = 1000;
    std::vector<unsigned> tab(rows * 4), off(64);
    for (unsigned i = 0; i < rows * 4; ++i) tab[i] = 0x10000u * (i / 4) + (i % 4);
    for (unsigned l = 0; l < 64; ++l) off[l] = (l % 5 == 3) ? 0xFFFFFFFFu : (l == 7 ? 16 * rows + 32 : 16 * ((l * 37 + 11) % rows));
    off[9] = 16 * rows - 8;  // straddles the end of the table
    unsigned *dt, *doff, *dc, *dm;
    hipMalloc(&dt, tab.size() * 4); hipMalloc(&doff, 256); hipMalloc(&dc, 4096); hipMalloc(&dm, 1024);
    hipMemcpy(dt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice); hipMemcpy(doff, off.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, dt, rows * 16, doff, dc, dm);
    std::vector<unsigned> c(1024), m(192);
    hipMemcpy(c.data(), dc, 4096, hipMemcpyDeviceToHost); hipMemcpy(m.data(), dm, 768, hipMemcpyDeviceToHost);
    int bad = 0;
    for (unsigned i = 0; i < 1024; ++i) {
        unsigned want = 0xAAAA0000u + i;
        if (i >= 256 && i < 512) {
            unsigned l = (i - 256) / 4, w = i % 4;
            if (off[l] != 0xFFFFFFFFu && l != 7 && l != 9) want = tab[off[l] / 4 + w];
            if (l == 7 || l == 9) { printf("lane %u word %u (offset %s): 0x%08x\n", l, w, l == 7 ? "past the end" : "straddling", c[i]); continue; }
        }
        if (c[i] != want) { if (bad < 10) printf("MISMATCH word %u: got 0x%08x want 0x%08x\n", i, c[i], want); ++bad; }
    }
    printf("cells: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
    int bad2 = 0;
    for (unsigned l = 0; l < 64; ++l) {
        const unsigned* cell = &c[256 + 4 * l];
        unsigned w0 = (cell[0] >> 16) | (cell[1] << 16), w1 = (cell[1] >> 16) | (cell[2] << 16), h = cell[1] >> 16;
        if (m[3 * l] != w0 || m[3 * l + 1] != w1 || m[3 * l + 2] != h) ++bad2;
    }
    printf("unaligned ds_read_b64 / ds_read_u16: %s\n", bad2 ? "FAIL" : "ok");
    return bad || bad2;
}
