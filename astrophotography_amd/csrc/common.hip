#include "common.h"

namespace apgpu {

char *err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(APGPU_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return APGPU_OK;
}

}  // namespace apgpu

extern "C" const char *apgpu_last_error(void) { return apgpu::err_buf(); }
extern "C" int apgpu_version(void) { return APGPU_VERSION; }
