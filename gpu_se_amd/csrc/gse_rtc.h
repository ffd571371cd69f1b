// gse_rtc.h -- user models compiled at run time (gse_rtc.hip).  A UserModel holds the generated model header (the
// restricted-Python f / g translated to two HIP device functions, gpu_se_amd/usermodel.py) and the kernels that have been
// compiled for it so far: the particle-filter sources (gse_pf_kernels.h, gse_rows_kernels.h) are compiled IN THE PROCESS by
// hiprtc with the model swapped in -- one template instance at a time, the first time a launch asks for it (about a second
// each), no compiler binary, no files -- loaded as a code-object module and launched through hipModuleLaunchKernel with the
// argument list the built-in launch uses.  What Numba's JIT does for the reference (filter/particle.py:176-208).
#pragma once

#include "gse_host.h"

namespace gse {

struct UserModel;

// The kernel `name_expression` (a C++ name such as "gse::pf_step_kernel<true, true, false, true, 2, 3>") compiled for the
// model; null + gse_last_error() on failure.
hipFunction_t user_kernel(UserModel* model, const char* name_expression);

// hipModuleLaunchKernel with error reporting in the style of GSE_LAUNCH_CHECK.
int launch_user_kernel(UserModel* model, const char* name_expression, dim3 grid, dim3 block, size_t dynamic_lds,
                       hipStream_t stream, void** params);

int user_model_nx(const UserModel* model);

}  // namespace gse
