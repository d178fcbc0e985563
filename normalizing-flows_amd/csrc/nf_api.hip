// Version / diagnostics entry points of libnf_hip.so (see include/nf_hip.h).
#include "nf_common.h"

extern "C" int nf_version(void) { return 100; }

extern "C" const char* nf_strerror(int code) {
    switch (code) {
        case NF_OK: return "ok";
        case NF_ERR_NULL: return "a required pointer is NULL";
        case NF_ERR_SHAPE: return "sizes inconsistent or outside the supported range";
        case NF_ERR_LAYOUT: return "stride or alignment not accepted";
        case NF_ERR_BINS: return "minimal bin width/height too large for the number of bins";
        case NF_ERR_DTYPE: return "unknown dtype or layout";
        case NF_ERR_SCRATCH: return "workspace too small";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}
