#include "common.hpp"

#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

namespace mrcnn {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

namespace {
std::mutex g_dev_mutex;
std::map<std::tuple<int, const void*>, size_t> g_lds_attr;  // (device, kernel) -> largest size already granted
std::map<int, int> g_cu_count;
}  // namespace

int ensure_dynamic_lds(const void* kernel, size_t lds_bytes, const char* who) {
    if (lds_bytes <= 64 * 1024) return MRCNN_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(MRCNN_ERR_LAUNCH, "%s: hipGetDevice failed", who);
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    size_t& granted = g_lds_attr[std::make_tuple(dev, kernel)];
    if (granted >= lds_bytes) return MRCNN_OK;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
    if (e != hipSuccess) return fail(MRCNN_ERR_LAUNCH, "%s: hipFuncSetAttribute(%zu B of LDS): %s", who, lds_bytes,
                                     hipGetErrorString(e));
    granted = lds_bytes;
    return MRCNN_OK;
}

int device_cu_count() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    int& n = g_cu_count[dev];
    if (n == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        n = prop.multiProcessorCount;
    }
    return n;
}

}  // namespace mrcnn

extern "C" {
int mrcnn_abi_version(void) { return MRCNN_ABI_VERSION; }
const char* mrcnn_last_error(void) { return mrcnn::error_buffer(); }
const char* mrcnn_arch(void) { return "gfx950"; }
}
