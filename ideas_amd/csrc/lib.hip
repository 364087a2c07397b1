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

// A HIP stream of the lowest (prio < 0), default (0) or highest (prio > 0) priority the device offers.  The gradient sink
// (ideas_amd/op/conv.py) runs the weight-gradient kernels on a LOW-priority stream: they are throughput work that should fill what
// the critical path (the input-gradient chain on the caller's stream) leaves free, not compete with it block for block -- with
// equal priorities a 20-us elementwise launch of the main stream waited up to 2 ms for a CU behind a weight-gradient grid.
extern "C" int ideas_stream_create(void** out, int prio) {
    if (!out) return IDEAS_E_NULL;
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return (int)e;
    hipStream_t s = nullptr;
    e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio < 0 ? least : prio > 0 ? greatest : 0);
    if (e != hipSuccess) return (int)e;
    *out = (void*)s;
    return IDEAS_OK;
}

extern "C" int ideas_stream_destroy(void* stream) {
    if (!stream) return IDEAS_E_NULL;
    return (int)hipStreamDestroy((hipStream_t)stream);
}

extern "C" int ideas_sizeof_prep_desc(void) { return (int)sizeof(ideas_prep_desc); }

extern "C" int ideas_weight_prep_batched(const ideas_prep_desc* table, int n, int op, int total_blocks, void* stream) {
    if (!table) return IDEAS_E_NULL;
    if (n <= 0 || total_blocks <= 0) return IDEAS_E_SHAPE;
    switch (op) {
        case IDEAS_PREP_B3_SPLIT: ideas_b3_split_batched(table, n, total_blocks, (hipStream_t)stream); break;
        case IDEAS_PREP_B3_WINO: ideas_b3_wino_split_batched(table, n, total_blocks, (hipStream_t)stream); break;
        case IDEAS_PREP_BF16_PACK: ideas_bf16_pack_batched(table, n, total_blocks, (hipStream_t)stream); break;
        default: return IDEAS_E_UNSUPPORTED;
    }
    return ideas_launch_status();
}
