// tests/host_emul/esekf_mock.hpp -- TEST INFRASTRUCTURE ONLY.
//
// A stand-in for esekfom::esekf (include/IKFoM_toolkit/esekfom/esekfom.hpp) with the members FAST-LIVO's Mode-23 path touches,
// spelled as in the reference so that the registration line of laserMapping.cpp:1233-1235 compiles against it verbatim:
//   typedef void measurementModel_dyn_share(state &, dyn_share_datastruct<scalar_type> &);                      esekfom.hpp:129
//   void init_dyn_share(processModel f_in, processMatrix1 f_x_in, processMatrix2 f_w_in,
//                       measurementModel_dyn_share h_dyn_share_in, int maximum_iteration, scalar_type limit_vector[n]);  :238-254
//   void update_iterated_dyn_share_modified(double R, double &solve_time);                                       :1619-1928
//   get_x / get_P / change_x / change_P                                                                          :1940-1958
// The updater's arithmetic is NOT re-implemented here: update_iterated_dyn_share_modified drives the CPU oracle's line-by-line
// restatement of esekfom.hpp:1619-1928 (oracle/orc_ikfom.c: orc_ikfom_update_dyn_share), calling the registered function pointer
// exactly where the reference does (:1636).  The real header needs Eigen + Boost, which this image does not have.
#pragma once

#include "../../fast-livo_amd/host/fastlivo_types.hpp"
extern "C" {
#include "../../oracle/fastlivo_oracle.h"
}

#include <cstring>

namespace fastlivo_host {
namespace esekfom {

template <typename state, int process_noise_dof, typename input = state, typename measurement = state, int measurement_noise_dof = 0>
class esekf {
public:
    typedef double scalar_type;
    enum { n = state::DOF };
    struct cov { double d[n * n]; double &operator()(int r, int c) { return d[r * n + c]; } double operator()(int r, int c) const { return d[r * n + c]; } };
    typedef void *processModel;       // get_f / df_dx / df_dw are not on the measurement path; the mock only stores them
    typedef void *processMatrix1;
    typedef void *processMatrix2;
    typedef void measurementModel_dyn_share(state &, dyn_share_datastruct<scalar_type> &);

    void init_dyn_share(processModel f_in, processMatrix1 f_x_in, processMatrix2 f_w_in, measurementModel_dyn_share h_dyn_share_in,
                        int maximum_iteration, scalar_type limit_vector[n])
    {
        f = f_in; f_x = f_x_in; f_w = f_w_in;
        h_dyn_share = h_dyn_share_in;
        maximum_iter = maximum_iteration;
        for (int i = 0; i < n; i++) limit[i] = limit_vector[i];
    }

    void update_iterated_dyn_share_modified(double R, double &solve_time)
    {
        orc_state23 xo;
        to_orc(x_, xo);
        dyn_share_datastruct<scalar_type> dyn_share;          // stack object of esekfom.hpp:1621
        dyn_share.valid = true; dyn_share.converge = true;
        Tramp tr{this, &dyn_share};
        orc_ikfom_out out;
        last_status = orc_ikfom_update_dyn_share(&xo, P_.d, R, maximum_iter, limit, &esekf::trampoline, &tr, &out);
        from_orc(xo, x_);
        iterations = out.iterations;
        solve_time = 0.0;
    }

    const state &get_x() const { return x_; }
    const cov &get_P() const { return P_; }
    void change_x(const state &s) { x_ = s; }
    void change_P(const cov &P) { P_ = P; }
    int iterations = 0, last_status = 0;

private:
    struct Tramp { esekf *self; dyn_share_datastruct<scalar_type> *dyn; };
    static void to_orc(const state &s, orc_state23 &o)
    {
        std::memcpy(o.pos, s.pos.v, sizeof o.pos);
        o.rot[0] = s.rot.x; o.rot[1] = s.rot.y; o.rot[2] = s.rot.z; o.rot[3] = s.rot.w;
        o.offset_R_L_I[0] = s.offset_R_L_I.x; o.offset_R_L_I[1] = s.offset_R_L_I.y; o.offset_R_L_I[2] = s.offset_R_L_I.z; o.offset_R_L_I[3] = s.offset_R_L_I.w;
        std::memcpy(o.offset_T_L_I, s.offset_T_L_I.v, sizeof o.offset_T_L_I);
        std::memcpy(o.vel, s.vel.v, sizeof o.vel); std::memcpy(o.bg, s.bg.v, sizeof o.bg);
        std::memcpy(o.ba, s.ba.v, sizeof o.ba); std::memcpy(o.grav, s.grav.v, sizeof o.grav);
    }
    static void from_orc(const orc_state23 &o, state &s)
    {
        std::memcpy(s.pos.v, o.pos, sizeof o.pos);
        s.rot.x = o.rot[0]; s.rot.y = o.rot[1]; s.rot.z = o.rot[2]; s.rot.w = o.rot[3];
        s.offset_R_L_I.x = o.offset_R_L_I[0]; s.offset_R_L_I.y = o.offset_R_L_I[1]; s.offset_R_L_I.z = o.offset_R_L_I[2]; s.offset_R_L_I.w = o.offset_R_L_I[3];
        std::memcpy(s.offset_T_L_I.v, o.offset_T_L_I, sizeof o.offset_T_L_I);
        std::memcpy(s.vel.v, o.vel, sizeof o.vel); std::memcpy(s.bg.v, o.bg, sizeof o.bg);
        std::memcpy(s.ba.v, o.ba, sizeof o.ba); std::memcpy(s.grav.v, o.grav, sizeof o.grav);
    }
    // esekfom.hpp:1636: h_dyn_share(x_, dyn_share) -- through the stored function pointer, on the real types
    static void trampoline(void *ctx, orc_state23 *x, int *valid, int *converge, int *rows, const double **h_x, const double **h)
    {
        Tramp *t = (Tramp *)ctx;
        state s;
        from_orc(*x, s);
        t->dyn->valid = (*valid != 0);
        t->dyn->converge = (*converge != 0);
        t->self->h_dyn_share(s, *t->dyn);
        *valid = t->dyn->valid ? 1 : 0;
        *rows = t->dyn->h_x.rows();
        *h_x = t->dyn->h_x.d.data();        // row-major rows x 12
        *h = t->dyn->h.d.data();
    }
    processModel f = nullptr; processMatrix1 f_x = nullptr; processMatrix2 f_w = nullptr;
    measurementModel_dyn_share *h_dyn_share = nullptr;
    int maximum_iter = 0;
    scalar_type limit[n] = {};
    state x_;
    cov P_{};
};
}  // namespace esekfom
}  // namespace fastlivo_host
