// ABI bookkeeping of librgda_hip.so.
#include "common.h"

extern "C" int rgda_abi_version(void) { return RGDA_ABI_VERSION; }

extern "C" const char* rgda_strerror(int status) {
    switch (status) {
        case RGDA_OK: return "ok";
        case RGDA_ERR_ARG: return "invalid argument (null pointer, bad shape or unsupported parameter value)";
        case RGDA_ERR_WORKSPACE: return "workspace too small";
        case RGDA_ERR_LAUNCH: return "HIP launch failed";
        case RGDA_ERR_UNSUPPORTED: return "configuration not supported by this build";
        default: return "unknown status";
    }
}
