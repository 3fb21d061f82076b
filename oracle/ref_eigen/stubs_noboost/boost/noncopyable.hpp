// TEST INFRASTRUCTURE ONLY (oracle/ref_eigen): stand-in for boost/noncopyable.hpp where Boost is not installed
// (point.h:30 `class Point : boost::noncopyable`).
#pragma once
namespace boost {
class noncopyable {
protected:
    noncopyable() {}
    ~noncopyable() {}
private:
    noncopyable(const noncopyable &);
    const noncopyable &operator=(const noncopyable &);
};
}  // namespace boost
