// Error plumbing, version and device queries of the C ABI.
#include "dq_common.hpp"
#include <string.h>
#include <stddef.h>

namespace dq {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return DQ_ERR_LAUNCH;
    }
    return DQ_OK;
}

int validate_bits(int n, const int* targets, int k, const int* controls, int nc) {
    if (n < 1 || n > 40) {
        set_error("n=%d out of range [1, 40]", n);
        return DQ_ERR_ARG;
    }
    if (k < 0 || nc < 0 || (k > 0 && !targets) || (nc > 0 && !controls)) {
        set_error("bad target/control list (k=%d, nc=%d)", k, nc);
        return DQ_ERR_ARG;
    }
    if (k + nc > n) {
        set_error("k+nc=%d exceeds n=%d", k + nc, n);
        return DQ_ERR_ARG;
    }
    uint64_t seen = 0;
    for (int i = 0; i < k + nc; ++i) {
        const int p = i < k ? targets[i] : controls[i - k];
        if (p < 0 || p >= n) {
            set_error("bit position %d out of range [0, %d)", p, n);
            return DQ_ERR_ARG;
        }
        if ((seen >> p) & 1ull) {
            set_error("bit position %d used twice among targets/controls", p);
            return DQ_ERR_ARG;
        }
        seen |= 1ull << p;
    }
    return DQ_OK;
}

}  // namespace dq

extern "C" int dq_abi_version(void) { return DQ_ABI_VERSION; }

extern "C" int dq_struct_layout(int* out, int max) {
    const int v[] = {(int)sizeof(DqFusedGate), (int)sizeof(DqFusedRound), (int)sizeof(DqFusedPass),
                     (int)offsetof(DqFusedPass, rounds), (int)offsetof(DqFusedPass, gates), (int)offsetof(DqFusedPass, load_slot_off),
                     (int)offsetof(DqFusedPass, store_high_pos), (int)offsetof(DqFusedPass, store_tb),
                     (int)offsetof(DqFusedPass, slots)};
    const int n = (int)(sizeof(v) / sizeof(v[0]));
    for (int i = 0; i < n && i < max && out; ++i) out[i] = v[i];
    return n;
}

extern "C" const char* dq_last_error(void) { return dq::g_err; }

extern "C" int dq_device_info(int* cu_count, int64_t* lds_per_block, int64_t* global_mem) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) {
        dq::set_error("dq_device_info: %s", hipGetErrorString(e));
        return DQ_ERR_LAUNCH;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_per_block) *lds_per_block = (int64_t)prop.sharedMemPerBlock;
    if (global_mem) *global_mem = (int64_t)prop.totalGlobalMem;
    return DQ_OK;
}
