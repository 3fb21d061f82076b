// TEST INFRASTRUCTURE ONLY (oracle/ref_eigen): stand-in so that the reference's include/common_lib.h compiles without Boost
// (only on the include path when no real Boost was found): common_lib.h:18,304 `typedef boost::shared_ptr<SparseMap> SparseMapPtr;`.
#pragma once
#include <memory>
namespace boost { template <typename T> using shared_ptr = std::shared_ptr<T>; }
