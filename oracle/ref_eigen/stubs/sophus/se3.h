// TEST INFRASTRUCTURE ONLY (oracle/ref_eigen): stand-in so that the reference's include/common_lib.h compiles without Sophus (common_lib.h:17,23: `using namespace Sophus;`, no Sophus type used in the header).
#pragma once
namespace Sophus {}
