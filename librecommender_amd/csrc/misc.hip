// Error strings / ABI version.
#include "common.hpp"

extern "C" const char* lr_strerror(int code) {
  switch (code) {
    case LR_OK: return "ok";
    case LR_EINVAL: return "invalid argument";
    case LR_ESHAPE: return "shape not supported by the compiled gfx950 kernels";
    case LR_EWORKSPACE: return "workspace too small";
    default: break;
  }
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  return "unknown error";
}

extern "C" int lr_abi_version(void) { return 8; }
