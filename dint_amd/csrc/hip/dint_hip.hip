// C ABI of the device decode path (include/dint_hip.h): dictionary staging, the host indexing pre-pass, and the kernel
// launches. ONE translation unit (the kernels are templates in headers; the handles are shared): its parts live under
// host/ by subsystem and are included here in order — common helpers and handles first, then the extern "C" block
// (opened in host/hip_api_common.inc, closed at the end of this file): dictionary, vroom decode, in-index decode,
// queries, statistics, host-pointer calls, list cache.
#include "dint_hip.h"

#include <hip/hip_runtime.h>

#include <cstring>  // (before rocPRIM: one of its headers calls the host memset without including it)
#include <rocprim/device/device_merge_sort.hpp>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "dint_kernels.hpp"
#include "dint_query_kernels.hpp"
#include "dint_stats_kernels.hpp"

#include "host/hip_common.inc"
#include "host/hip_handles.inc"
#include "host/hip_dictionary.inc"
#include "host/hip_query_handle.inc"
#include "host/hip_api_common.inc"
#include "host/hip_api_dictionary.inc"
#include "host/hip_api_vroom.inc"
#include "host/hip_api_index.inc"
#include "host/hip_api_query.inc"
#include "host/hip_api_stats.inc"
#include "host/hip_api_host_calls.inc"
#include "host/hip_api_list_cache.inc"

int dint_debug_alloc_count(uint64_t* count) {
    if (!count) return DINT_ERR_ARG;
    *count = g_alloc_count.load();
    return DINT_OK;
}

#ifdef DINT_PROFILE
}  // extern "C"
#include "dint_profile_host.inc"
extern "C" {
#endif

// Test hook (not part of the decode ABI): inclusive wave prefix sum of 64 host words.
int dint_debug_wave_scan(const uint32_t* in64, uint32_t* out64) {
    if (!in64 || !out64) return DINT_ERR_ARG;
    uint32_t *d_in = nullptr, *d_out = nullptr;
    HIP_TRY(counted_malloc(&d_in, 256));
    HIP_TRY(counted_malloc(&d_out, 256));
    HIP_TRY(hipMemcpy(d_in, in64, 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(debug_wave_scan_kernel, dim3(1), dim3(64), 0, nullptr, d_in, d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out64, d_out, 256, hipMemcpyDeviceToHost));
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return DINT_OK;
}

}  // extern "C"

