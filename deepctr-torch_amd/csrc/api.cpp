// api.cpp -- ABI bookkeeping entry points of libdctr_hip.so.
#include <hip/hip_runtime.h>

#include "dctr.h"

extern "C" int dctr_abi_version(void) { return DCTR_ABI_VERSION; }

extern "C" size_t dctr_sizeof_field(void) { return sizeof(dctr_field_t); }
extern "C" size_t dctr_sizeof_plan(void) { return sizeof(dctr_plan_t); }
extern "C" size_t dctr_sizeof_uslot(void) { return sizeof(dctr_uslot_t); }
extern "C" size_t dctr_sizeof_vunit(void) { return sizeof(dctr_vunit_t); }
extern "C" size_t dctr_sizeof_plan_ext(void) { return sizeof(dctr_plan_ext_t); }
extern "C" size_t dctr_sizeof_mlp(void) { return sizeof(dctr_mlp_t); }
extern "C" size_t dctr_sizeof_dense_step(void) { return sizeof(dctr_dense_step_t); }
extern "C" size_t dctr_sizeof_dense_item(void) { return sizeof(dctr_dense_item_t); }

extern "C" const char* dctr_strerror(int code) {
  switch (code) {
    case DCTR_OK: return "ok";
    case DCTR_EINVAL: return "dctr: invalid argument (null / negative / inconsistent)";
    case DCTR_ENOSUP: return "dctr: shape not supported by the gfx950 kernels";
    case DCTR_EALIGN: return "dctr: pointer or leading dimension is not aligned for vector access";
    default: break;
  }
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  return "dctr: unknown error code";
}
