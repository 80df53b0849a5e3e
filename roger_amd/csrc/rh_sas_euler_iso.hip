// The explicit Euler solver of the SAS transport step for the isotopes: one translation unit (rh_sas_solvers_impl.h).
#include <hip/hip_runtime.h>

#include "roger_hip.h"
#include "roger_hip_sas.h"
#define RH_SOLVER_RK4 0
#define RH_SOLVER_ANION 0
#define RH_SOLVER_NAME rh_sas_launch_euler_iso
#include "rh_sas_solvers_impl.h"
