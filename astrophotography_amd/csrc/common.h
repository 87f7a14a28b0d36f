// Shared helpers for the libapgpu.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "apgpu.h"

namespace apgpu {

// Thread-local message behind apgpu_last_error().
char *err_buf();
int fail(int code, const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Call after every kernel launch: turns a launch failure into APGPU_ELAUNCH.
int check_launch(const char *what);

constexpr int kWave = 64;       // gfx950 wavefront
constexpr int kNumCU = 256;     // MI355X

}  // namespace apgpu
