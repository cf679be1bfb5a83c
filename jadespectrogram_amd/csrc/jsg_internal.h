// Internal helpers shared by the translation units of libjsg.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime_api.h>

#include <string>

#include "../../include/jsg.h"

namespace jsg {

// last error text of the calling thread (jsg_last_error(NULL)); engines keep their own copy too
std::string& tls_error();
int jsg_fail(int code, const char* what);
int jsg_fail_hip(hipError_t err, const char* where);

}  // namespace jsg
