// RAII scope used by every launcher: records a pair of HIP events on the launch stream when the
// opt-in profiler (ctk_profile_enable) is on.  `flops` / `bytes` are the ALGORITHMIC work of the launch.
#pragma once
#include <hip/hip_runtime.h>

bool ctk_profile_is_on();

class CtkProfScope {
 public:
  CtkProfScope(const char* name, double flops, double bytes, hipStream_t s);
  ~CtkProfScope();

 private:
  long idx_;
  hipStream_t s_;
};
