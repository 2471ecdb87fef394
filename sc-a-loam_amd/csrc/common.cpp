// Error reporting and device selection for libscaloam_hip.so.
#include "common.hpp"

namespace scal {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int select_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s): libscaloam_hip has no CPU fallback", e == hipSuccess ? "count 0" : hipGetErrorString(e));
        return SCAL_E_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device ordinal %d out of range (%d devices)", device, n);
        return SCAL_E_ARG;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        set_error("hipGetDeviceProperties failed: %s", hipGetErrorString(e));
        return SCAL_E_HIP;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library carries gfx950 (MI355X) code objects only", device, prop.gcnArchName);
        return SCAL_E_NO_DEVICE;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) {
        set_error("hipSetDevice(%d) failed: %s", device, hipGetErrorString(e));
        return SCAL_E_HIP;
    }
    return SCAL_OK;
}

}  // namespace scal

extern "C" const char* scal_last_error(void) { return scal::g_err; }
extern "C" int scal_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}
extern "C" const char* scal_version(void) { return "scaloam_hip 0.1 (gfx950)"; }
