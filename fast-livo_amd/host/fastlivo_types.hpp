// fastlivo_types.hpp -- Eigen-free stand-ins for the reference types that cross the hot-path boundary.
//
// The reference types are Eigen/MTK classes (not available in this image). These mirrors keep the
// SAME member names and meaning so that the shim code in fastlivo_shim.hpp reads like the code a
// maintainer pastes into the reference -- there the real types are used and the
// element-wise copies below become Eigen accessors.
//   StatesGroup                         include/common_lib.h:296-381
//   state_ikfom                         include/use-ikfom.hpp:12-21
//   esekfom::dyn_share_datastruct<T>    include/IKFoM_toolkit/esekfom/esekfom.hpp:79-89
//   SubSparseMap (fields the update reads) include/common_lib.h:263-292
#pragma once

#include <array>
#include <cstddef>
#include <cstring>
#include <vector>

namespace fastlivo_host {

struct M3D { double m[9]; double &operator()(int r, int c) { return m[r * 3 + c]; } double operator()(int r, int c) const { return m[r * 3 + c]; } };
struct V3D { double v[3]; double &operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; } };

constexpr int DIM_STATE = 18;   // common_lib.h:34

struct StatesGroup {
    M3D rot_end{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    V3D pos_end{}, vel_end{}, bias_g{}, bias_a{}, gravity{};
    double cov[DIM_STATE * DIM_STATE] = {};   // row-major
};

struct Quat { double x = 0, y = 0, z = 0, w = 1; };   // Eigen coeffs order

struct state_ikfom {
    V3D pos{};
    Quat rot, offset_R_L_I;
    V3D offset_T_L_I{}, vel{}, bg{}, ba{};
    V3D grav{{9.809, 0, 0}};
    static constexpr int DOF = 23;
};

// Dynamic matrix just big enough for dyn_share_datastruct (row-major storage, Eigen-like accessors).
struct DynMat {
    int r = 0, c = 0;
    std::vector<double> d;
    void resize(int rows, int cols) { r = rows; c = cols; d.assign((size_t)rows * cols, 0.0); }
    int rows() const { return r; }
    int cols() const { return c; }
    double &operator()(int i, int j) { return d[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return d[(size_t)i * c + j]; }
};
struct DynVec {
    std::vector<double> d;
    void resize(int n) { d.assign((size_t)n, 0.0); }
    int size() const { return (int)d.size(); }
    double &operator()(int i) { return d[(size_t)i]; }
    double operator()(int i) const { return d[(size_t)i]; }
};

namespace esekfom {
template <typename T>
struct dyn_share_datastruct {
    bool valid = true;
    bool converge = true;
    DynVec z, h;
    DynMat h_v, h_x, R;
};
}  // namespace esekfom

struct SubSparseMapView {          // what UpdateState reads of sub_sparse_map
    std::vector<float> patch;      // m x 3 x 64
    std::vector<double> pos;       // m x 3 (voxel_points[i]->pos_)
    std::vector<int> search_levels;
    std::vector<float> errors;
};

}  // namespace fastlivo_host
