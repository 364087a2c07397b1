// libideas_hip.so: version / error strings of the C ABI declared in include/ideas_hip.h.
#include "common.hpp"

extern "C" int ideas_abi_version(void) { return IDEAS_ABI_VERSION; }

extern "C" int ideas_sizeof_conv_params(void) { return (int)sizeof(ideas_conv_params); }

extern "C" const char* ideas_strerror(int code) {
    switch (code) {
        case IDEAS_OK: return "ok";
        case IDEAS_E_NULL: return "required pointer is NULL";
        case IDEAS_E_SHAPE: return "bad or inconsistent dimension";
        case IDEAS_E_UNSUPPORTED: return "unsupported dtype / layout / mode";
        case IDEAS_E_ALIGN: return "pointer or channel count not 16-byte aligned";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown ideas_hip error";
}
