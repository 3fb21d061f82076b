// oracle/ref_ikdtree/stubs/pcl/point_types.h -- TEST INFRASTRUCTURE ONLY.
// Stand-in for the one PCL type the reference's ikd-Tree uses (include/ikd-Tree/ikd_Tree.h:2,22): pcl::PointXYZINormal.
// Same member names and the 48-byte, 16-aligned layout PCL gives it (xyz+pad, normal+pad, intensity, curvature, pad);
// the tree itself only reads/writes x, y, z.  PCL is not installed in this image, this header lets the reference's own
// ikd_Tree.cpp compile unmodified.
#pragma once
#include <cmath>
#include <cstring>
namespace pcl {
struct alignas(16) PointXYZINormal {
    float x, y, z, _pad0;
    float normal_x, normal_y, normal_z, _pad1;
    float intensity, curvature, _pad2, _pad3;
    PointXYZINormal() : x(0.f), y(0.f), z(0.f), _pad0(1.f), normal_x(0.f), normal_y(0.f), normal_z(0.f), _pad1(0.f),
                        intensity(0.f), curvature(0.f), _pad2(0.f), _pad3(0.f) {}
};
}  // namespace pcl
