#!/bin/bash
# oracle/ref_eigen/ref_text.sh -- TEST INFRASTRUCTURE ONLY.
# Writes ONE translation unit to stdout: hand-written pieces from text/ with line ranges of the REFERENCE's sources between them,
# read from $REF at build time and piped straight into the compiler by the Makefile (no reference text is stored in this repository
# or in oracle/_ref).  The ranges are functions / statements the reference only has inside main() or inside a class whose header
# needs OpenCV, PCL, vikit and Sophus; taking the text is the only way to get THOSE LINES through a compiler here.
#   ref_text.sh lio   -> both pointBodyToWorld (laserMapping.cpp:272-301), `rematch_num` / `nearest_search_en` (:1472-1473), the Mode-18
#                        loop (:1506-1732); DET_RANGE / MOV_THRESHOLD (:83, :90), points_cache_collect (:324-330), lasermap_fov_segment
#                        (:361-421), map_incremental (:692-706); h_share_model (:961-1093, the Mode-23 measurement model; its state type is a stand-in with
#                        Eigen::Quaternion members, text/lio_1b.inc)
#   ref_text.sh vio   -> everything of LidarSelector / Feature / Point that is pinned, one unit: struct Feature (feature.h:27-63), class Point
#                        (point.h:27-107) with its member functions (point.cpp:23-98, 110-247), Frame::w2c / w2f / pos (frame.h:89,98,107);
#                        lidar_selection.cpp: set_extrinsic :35-39, init :41-71 + :73, reset_grid :81-90, dpi :92-103, getpatch :119-140,
#                        addSparseMap :142-202, AddPoint :204-230, getWarpMatrixAffine :232-256, warpAffine :258-296, NCC :298-315,
#                        getBestSearchLevel :317-331, addFromSparseMap :346-587 (and its loop :476-582 once more on its own),
#                        UpdateState :743-902, updateFrameState :904-911, addObservation :913-965, ComputeJ :967-983
#   ref_text.sh imu   -> ImuProcess::UndistortPcl (IMU_Processing.cpp:611-809)
#   ref_text.sh ikf   -> the IKFoM toolkit as far as the Mode-23 update uses it: vect::boxplus / boxminus (mtk/types/vect.hpp:117-122), pi /
#                        tolerance / cos_sinc_sqrt / hat / A_matrix / exp / log (mtk/src/mtkmath.hpp:117-122, 142-183, 235-256, 268-288), struct SO3
#                        (mtk/types/SOn.hpp:177-298), struct S2 (mtk/types/S2.hpp:97-310), dyn_share_datastruct (esekfom.hpp:79-89),
#                        esekf::update_iterated_dyn_share_modified (:1619-1928); stood in for: vectview, the Matrix base of vect, the
#                        Boost-generated compound state (text/ikf_1.inc, ikf_1c.inc)
# Every range is anchored: the first and the last line must look as they did in the snapshot the line numbers were taken from
# (reference of 2024-11-08), otherwise the script fails and nothing is built.
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
LM=$REF/src/laserMapping.cpp
LS=$REF/src/lidar_selection.cpp
IP=$REF/src/IMU_Processing.cpp
EK=$REF/include/IKFoM_toolkit/esekfom/esekfom.hpp
MV=$REF/include/IKFoM_toolkit/mtk/types/vect.hpp
MM=$REF/include/IKFoM_toolkit/mtk/src/mtkmath.hpp
MS=$REF/include/IKFoM_toolkit/mtk/types/SOn.hpp
M2=$REF/include/IKFoM_toolkit/mtk/types/S2.hpp
FH=$REF/include/frame.h
FE=$REF/include/feature.h
PH=$REF/include/point.h
PC=$REF/src/point.cpp

anchor() {   # file line regex
    sed -n "$2p" "$1" | grep -Eq "$3" || { echo "ref_text.sh: $1:$2 does not match /$3/ -- the reference differs from the snapshot" >&2; exit 1; }
}
range() {    # file first last
    echo "// ---- reference text $(basename "$1"):$2-$3 (not stored; read at build time)"
    echo "#line $2 \"$1\""
    sed -n "$2,$3p" "$1"
    echo "// ---- end of reference text"
    echo "#line 1 \"hand-written piece after $(basename "$1"):$3\""
}

case "$1" in
lio)
    anchor "$LM" 272 '^void pointBodyToWorld\(PointType const'
    anchor "$LM" 286 '^\}'
    anchor "$LM" 1472 'int +rematch_num = 0;'
    anchor "$LM" 1473 'bool nearest_search_en = true;'
    anchor "$LM" 1506 'for \(iterCount = -1; iterCount < NUM_MAX_ITERATIONS && flg_EKF_inited; iterCount\+\+\)'
    anchor "$LM" 1731 'if \(EKF_stop_flg\) +break;'
    anchor "$LM" 1732 '^ {12}\}'
    anchor "$LM" 83 '^float DET_RANGE = 300.0f;'
    anchor "$LM" 90 'const float MOV_THRESHOLD = 1.5f;'
    anchor "$LM" 288 '^template<typename T>'
    anchor "$LM" 301 '^\}'
    anchor "$LM" 324 '^int points_cache_size = 0;'
    anchor "$LM" 330 '^\}'
    anchor "$LM" 361 '^BoxPointType LocalMap_Points;'
    anchor "$LM" 363 '^void lasermap_fov_segment\(\)'
    anchor "$LM" 421 '^\}'
    anchor "$LM" 692 '^void map_incremental\(\)'
    anchor "$LM" 706 '^\}'
    cat "$HERE/text/lio_1.inc"
    range "$LM" 83 83
    range "$LM" 90 90
    range "$LM" 272 301
    range "$LM" 324 330
    range "$LM" 361 421
    range "$LM" 692 706
    anchor "$EK" 79 '^template<typename T>'
    anchor "$EK" 80 '^struct dyn_share_datastruct'
    anchor "$EK" 89 '^\};'
    anchor "$LM" 960 '^#ifdef USE_IKFOM'
    anchor "$LM" 961 '^void h_share_model\(state_ikfom &s, esekfom::dyn_share_datastruct<double> &ekfom_data\)'
    anchor "$LM" 1093 '^\}'
    anchor "$LM" 1094 '^#endif'
    cat "$HERE/text/lio_1b.inc"
    range "$EK" 79 89
    cat "$HERE/text/lio_1c.inc"
    range "$LM" 961 1093
    cat "$HERE/text/lio_2.inc"
    range "$LM" 1472 1473
    cat "$HERE/text/lio_3.inc"
    range "$LM" 1506 1732
    cat "$HERE/text/lio_4.inc"
    ;;
vio)
    anchor "$LS" 35 '^void LidarSelector::set_extrinsic\('
    anchor "$LS" 39 '^\}'
    anchor "$LS" 41 '^void LidarSelector::init\(\)'
    anchor "$LS" 52 'Jdp_dR = -Rci \* tmp;'
    anchor "$LS" 58 'fx = cam->errorMultiplier2\(\);'
    anchor "$LS" 59 'fy = cam->errorMultiplier\(\) / \(4\. \* fx\);'
    anchor "$LS" 92 '^void LidarSelector::dpi\('
    anchor "$LS" 103 '^\}'
    anchor "$LS" 743 '^float LidarSelector::UpdateState\('
    anchor "$LS" 902 '^\}'
    anchor "$LS" 904 '^void LidarSelector::updateFrameState\('
    anchor "$LS" 911 '^\}'
    anchor "$LS" 967 '^void LidarSelector::ComputeJ\('
    anchor "$LS" 983 '^\}'
    anchor "$FH" 89 'inline Vector2d w2c\(const Vector3d& xyz_w\) const'
    anchor "$FH" 98 'inline Vector3d w2f\(const Vector3d& xyz_w\) const'
    anchor "$FH" 107 'inline Vector3d pos\(\) const'
    anchor "$FE" 27 '^struct Feature'
    anchor "$FE" 63 '^\};'
    anchor "$PH" 27 '^typedef Matrix<double, 2, 3> Matrix23d;'
    anchor "$PH" 30 '^class Point : boost::noncopyable'
    anchor "$PH" 107 '^\};'
    anchor "$PC" 23 '^int Point::point_counter_ = 0;'
    anchor "$PC" 88 '^void Point::deleteFeatureRef\(FeaturePtr ftr\)'
    anchor "$PC" 98 '^\}'
    anchor "$PC" 110 '^bool Point::getClosePose\('
    anchor "$PC" 141 '^bool Point::getCloseViewObs\('
    anchor "$PC" 219 '^void Point::getFurthestViewObs\('
    anchor "$PC" 247 '^\}'
    anchor "$LS" 71 'patch_cache.resize\(patch_size_total\);'
    anchor "$LS" 73 'pg_down.reset\(new PointCloudXYZI\(\)\);'
    anchor "$LS" 81 '^void LidarSelector::reset_grid\(\)'
    anchor "$LS" 90 '^\}'
    anchor "$LS" 119 '^void LidarSelector::getpatch\(cv::Mat img, V2D pc, float\* patch_tmp, int level\)'
    anchor "$LS" 140 '^\}'
    anchor "$LS" 142 '^void LidarSelector::addSparseMap\('
    anchor "$LS" 202 '^\}'
    anchor "$LS" 204 '^void LidarSelector::AddPoint\(PointPtr pt_new\)'
    anchor "$LS" 230 '^\}'
    anchor "$LS" 232 '^void LidarSelector::getWarpMatrixAffine\('
    anchor "$LS" 256 '^\}'
    anchor "$LS" 258 '^void LidarSelector::warpAffine\('
    anchor "$LS" 296 '^\}'
    anchor "$LS" 298 '^double LidarSelector::NCC\('
    anchor "$LS" 315 '^\}'
    anchor "$LS" 317 '^int LidarSelector::getBestSearchLevel\('
    anchor "$LS" 331 '^\}'
    anchor "$LS" 346 '^void LidarSelector::addFromSparseMap\(cv::Mat img, PointCloudXYZI::Ptr pg\)'
    anchor "$LS" 476 '^    for \(int i=0; i<length; i\+\+\)'
    anchor "$LS" 478 'if \(grid_num\[i\]==TYPE_MAP\)'
    anchor "$LS" 582 '^    \}'
    anchor "$LS" 587 '^\}'
    anchor "$LS" 913 '^void LidarSelector::addObservation\(cv::Mat img\)'
    anchor "$LS" 965 '^\}'
    cat "$HERE/text/vio_1.inc"
    range "$FH" 89 89
    range "$FH" 98 98
    range "$FH" 107 107
    cat "$HERE/text/vio_1b.inc"
    range "$FE" 27 63
    range "$PH" 27 107
    range "$PC" 23 98
    range "$PC" 110 247
    cat "$HERE/text/vio_1c.inc"
    range "$LS" 35 39
    range "$LS" 41 71
    range "$LS" 73 73
    cat "$HERE/text/vio_2.inc"
    range "$LS" 81 90
    range "$LS" 92 103
    range "$LS" 119 140
    range "$LS" 142 202
    range "$LS" 204 230
    range "$LS" 232 256
    range "$LS" 258 296
    range "$LS" 298 315
    range "$LS" 317 331
    cat "$HERE/text/vio_2b.inc"
    range "$LS" 476 582
    cat "$HERE/text/vio_2c.inc"
    range "$LS" 346 587
    range "$LS" 743 902
    range "$LS" 904 911
    range "$LS" 913 965
    range "$LS" 967 983
    cat "$HERE/text/vio_3.inc"
    ;;
imu)
    anchor "$IP" 611 '^void ImuProcess::UndistortPcl\(LidarMeasureGroup &lidar_meas, StatesGroup &state_inout, PointCloudXYZI &pcl_out\)'
    anchor "$IP" 809 '^\}'
    anchor "$IP" 811 '^void ImuProcess::Process2\('
    cat "$HERE/text/imu_1.inc"
    range "$IP" 611 809
    cat "$HERE/text/imu_2.inc"
    ;;
ikf)
    anchor "$EK" 79 '^template<typename T>'
    anchor "$EK" 80 '^struct dyn_share_datastruct'
    anchor "$EK" 89 '^\};'
    anchor "$EK" 1619 'void update_iterated_dyn_share_modified\(double R, double &solve_time\) \{'
    anchor "$EK" 1926 'solve_time \+= omp_get_wtime\(\) - solve_start;'
    anchor "$EK" 1928 '^[[:space:]]\}'
    anchor "$EK" 1930 'void change_x\(state &input_state\)'
    anchor "$MV" 117 'void boxplus\(MTK::vectview<const scalar, D> vec, scalar scale=1\) \{'
    anchor "$MV" 122 '^\t\}|^[[:space:]]\}'
    anchor "$MM" 117 '^const double pi = M_PI;'
    anchor "$MM" 122 'tolerance<double>\(\) \{ return 1e-11; \}'
    anchor "$MM" 143 '^std::pair<scalar, scalar> cos_sinc_sqrt\(const scalar &x2\)\{'
    anchor "$MM" 183 '^\}'
    anchor "$MM" 236 'A_matrix\(const Base & v\)\{'
    anchor "$MM" 250 '^scalar exp\(vectview<scalar, n> result, vectview<const scalar, n> vec, const scalar& scale = 1\) \{'
    anchor "$MM" 256 '^\}'
    anchor "$MM" 269 '^void log\(vectview<scalar, n> result,'
    anchor "$MM" 288 '^\}'
    anchor "$MS" 178 '^struct SO3 : public Eigen::Quaternion<_scalar, Options> \{'
    anchor "$MS" 298 '^\};'
    anchor "$M2" 98 '^struct S2 \{'
    anchor "$M2" 310 '^\};'
    cat "$HERE/text/ikf_1.inc"
    range "$MV" 117 122
    cat "$HERE/text/ikf_1b.inc"
    range "$MM" 117 122
    range "$MM" 142 183
    range "$MM" 235 256
    range "$MM" 268 288
    range "$MS" 177 298
    range "$M2" 97 310
    cat "$HERE/text/ikf_1c.inc"
    range "$EK" 79 89
    cat "$HERE/text/ikf_2.inc"
    range "$EK" 1619 1928
    cat "$HERE/text/ikf_3.inc"
    ;;
*)
    echo "usage: ref_text.sh lio|vio|imu|ikf" >&2; exit 2;;
esac
