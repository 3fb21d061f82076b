// ikfom_kernels.h -- Mode-23 (IKFoM) device block and kernels. (filled in below)
#pragma once
#include "fl_device.h"
#include "fl_math.h"

struct FlDev23 {
    double x[27];        // pos(3) rot(4 xyzw) offset_R_L_I(4) offset_T_L_I(3) vel bg ba grav(3)
    double xprop[27];
    double P[529];       // P_ (working / result)
    double Pprop[529];   // P_propagated
    double limit[23];
    double solution[23];
    double sums[FL_SUMS23];
    double total_residual;
    double meas_cov;
    int32_t iter_i;      // loop index i of esekfom.hpp:1633 (starts at -1)
    int32_t t_count;     // t
    int32_t need_search; // dyn_share.converge
    int32_t stop;
    int32_t converged;
    int32_t neff;
    int32_t status;
    int32_t iters_run;
    int32_t max_iter;
    int32_t pad;
};
