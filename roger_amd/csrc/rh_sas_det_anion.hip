// The deterministic SAS kernels for the anion tracers (bromide, chloride, virtual tracer): one translation unit (rh_sas_kernels.h).
#include <hip/hip_runtime.h>

#include "roger_hip.h"
#include "roger_hip_sas.h"
#define RH_SAS_DET_ANION 1
#define RH_SAS_DET_NAME rh_sas_launch_det_anion
#include "rh_sas_kernels.h"
