// The explicit RK4 solver of the SAS transport step for the anion tracers: one translation unit (rh_sas_solvers_impl.h).
#include <hip/hip_runtime.h>

#include "roger_hip.h"
#include "roger_hip_sas.h"
#define RH_SOLVER_RK4 1
#define RH_SOLVER_ANION 1
#define RH_SOLVER_NAME rh_sas_launch_rk4_anion
#include "rh_sas_solvers_impl.h"
