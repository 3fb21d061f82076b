// TEST INFRASTRUCTURE ONLY (oracle/ref_eigen): stand-in so that the reference's include/common_lib.h compiles without PCL.
// common_lib.h:52-53 names pcl::PointXYZINormal (PointType) and pcl::PointXYZRGB; esti_plane (:448-493) reads x, y, z of a PointVector.
// Member names and the 48-byte, 16-aligned layout are PCL's.
#pragma once
#include <cstdint>
namespace pcl {
struct alignas(16) PointXYZINormal {
    float x, y, z, _pad0;
    float normal_x, normal_y, normal_z, _pad1;
    float intensity, curvature, _pad2, _pad3;
    PointXYZINormal() : x(0.f), y(0.f), z(0.f), _pad0(1.f), normal_x(0.f), normal_y(0.f), normal_z(0.f), _pad1(0.f),
                        intensity(0.f), curvature(0.f), _pad2(0.f), _pad3(0.f) {}
};
struct alignas(16) PointXYZRGB {
    float x, y, z, _pad0;
    std::uint8_t b, g, r, a;
    float _pad1, _pad2, _pad3;
    PointXYZRGB() : x(0.f), y(0.f), z(0.f), _pad0(1.f), b(0), g(0), r(0), a(255), _pad1(0.f), _pad2(0.f), _pad3(0.f) {}
};
}  // namespace pcl
