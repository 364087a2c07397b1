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
