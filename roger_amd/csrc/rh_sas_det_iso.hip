// The deterministic SAS kernels for the isotopes (oxygen-18, deuterium): one translation unit (rh_sas_kernels.h).
#include <hip/hip_runtime.h>

#include "roger_hip.h"
#include "roger_hip_sas.h"
#define RH_SAS_DET_ANION 0
#define RH_SAS_DET_NAME rh_sas_launch_det_iso
#include "rh_sas_kernels.h"
