#include "common.hpp"

#include <cstring>

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

}  // namespace mrcnn

extern "C" {
int mrcnn_abi_version(void) { return MRCNN_ABI_VERSION; }
const char* mrcnn_last_error(void) { return mrcnn::error_buffer(); }
const char* mrcnn_arch(void) { return "gfx950"; }
}
