// TEST INFRASTRUCTURE: a stand-in for <hip/hip_runtime.h> that lets the *unmodified* kernel source
// (claxon_amd/csrc/clx_kernels.hip) be compiled by g++ and executed by a wave64 lock-step simulator
// (wavesim.h), so kernel logic can be checked on a machine without a GPU.  Never part of the product.
#ifndef WAVESIM_FAKE_HIP_RUNTIME_H
#define WAVESIM_FAKE_HIP_RUNTIME_H
#include "../../wavesim.h"
#endif
