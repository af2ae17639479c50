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

// ---- measurement hooks (include/egomi.h)
thread_local hipEvent_t egomi_time_start_ = nullptr, egomi_time_stop_ = nullptr;      // consumed by egomi_gemm_fast_try (gemm_fast.hip)

extern "C" int egomi_event_create(void** event) {
    if (!event) return EGOMI_E_BADARG;
    hipEvent_t e;
    const hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) { egomi_last_hip_error_ = (int)rc; return EGOMI_E_LAUNCH; }
    *event = (void*)e;
    return EGOMI_OK;
}
extern "C" int egomi_event_destroy(void* event) {
    if (!event) return EGOMI_E_BADARG;
    return hipEventDestroy((hipEvent_t)event) == hipSuccess ? EGOMI_OK : EGOMI_E_LAUNCH;
}
extern "C" int egomi_event_elapsed_ms(void* start, void* stop, float* ms) {
    if (!start || !stop || !ms) return EGOMI_E_BADARG;
    hipError_t rc = hipEventSynchronize((hipEvent_t)stop);
    if (rc == hipSuccess) rc = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
    if (rc != hipSuccess) {                                          // e.g. an event that was never recorded
        egomi_last_hip_error_ = (int)rc;
        (void)hipGetLastError();                                     // do not leave the runtime's sticky error for the next caller to trip over
        return EGOMI_E_LAUNCH;
    }
    return EGOMI_OK;
}
extern "C" int egomi_gemm_time_next(void* start, void* stop) {
    if ((start == nullptr) != (stop == nullptr)) return EGOMI_E_BADARG;
    egomi_time_start_ = (hipEvent_t)start; egomi_time_stop_ = (hipEvent_t)stop;
    return EGOMI_OK;
}
