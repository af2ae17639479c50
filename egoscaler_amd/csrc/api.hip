// Library identity + error strings.
#include "common.h"

thread_local int egomi_launch_err_ = 0;
thread_local int egomi_last_hip_error_ = 0;

extern "C" int egomi_version(void) { return 100; }

extern "C" const char* egomi_last_launch_error(void) { return hipGetErrorString((hipError_t)egomi_last_hip_error_); }

extern "C" const char* egomi_strerror(int code) {
    switch (code) {
        case EGOMI_OK: return "ok";
        case EGOMI_E_BADARG: return "bad argument (null pointer, unknown dtype/enum)";
        case EGOMI_E_SHAPE: return "shape/stride/workspace check failed";
        case EGOMI_E_LAUNCH: return "HIP launch failed";
        case EGOMI_E_UNSUPPORTED: return "size or mode not supported by this kernel";
        default: return "unknown egomi error";
    }
}
