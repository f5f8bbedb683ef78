// Library-level entry points of libsuperscreen_hip.so (version, error strings, device info).
#include <string.h>

#include "gemm_profile.hpp"

namespace ssa {
int chol_shutdown();
int lu_shutdown();
int chain_streams_shutdown();
}

extern "C" int ssa_abi_version(void) { return SSA_ABI_VERSION; }

extern "C" const char *ssa_error_string(int status) {
    switch (status) {
        case SSA_OK: return "ok";
        case SSA_ERR_INVALID_ARGUMENT: return "invalid argument";
        case SSA_ERR_HIP: return "HIP runtime error (launch or API call failed)";
        case SSA_ERR_WORKSPACE_TOO_SMALL: return "workspace missing or too small";
        case SSA_ERR_UNSUPPORTED_SIZE: return "problem size not supported by this build";
        case SSA_ERR_RCCL: return "RCCL not available (librccl.so could not be opened) or an RCCL call failed";
        default: return "unknown status";
    }
}

extern "C" int ssa_device_info(int *num_cus, size_t *hbm_bytes, char *arch_name,
                               int arch_name_len) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SSA_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return SSA_ERR_HIP;
    if (num_cus) *num_cus = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    if (arch_name && arch_name_len > 0) {
        strncpy(arch_name, prop.gcnArchName, static_cast<size_t>(arch_name_len) - 1);
        arch_name[arch_name_len - 1] = '\0';
    }
    return SSA_OK;
}

extern "C" int ssa_shutdown(void) {
    const int a = ssa::chol_shutdown();
    const int c = ssa::lu_shutdown();
    const int d = ssa::chain_streams_shutdown();   // after the schedules that use the chain streams
    const int b = ssa::profile_shutdown();
    return a != SSA_OK ? a : (c != SSA_OK ? c : (d != SSA_OK ? d : b));
}
