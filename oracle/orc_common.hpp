// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// Dependency-free C++17 CPU restatement of the SC-A-LOAM hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this code.
//
// Shared small types and Eigen-equivalent arithmetic helpers.  Each helper cites the
// third-party semantic it restates.  Build with
//   g++ -O3 -std=c++17 -ffp-contract=off      (reference flags: CMakeLists.txt:6-7, no FMA)
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include <limits>

namespace orc {

// pcl::PointXYZI as the reference uses it (include/aloam_velodyne/common.h:43): 4 floats, 16 B.
struct P4 {
    float x, y, z, i;
};

struct V3 {
    double x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }

// Eigen::Quaterniond, coefficient storage order (x, y, z, w).
struct Quat {
    double x, y, z, w;
};
// Eigen `q * v` (QuaternionBase::_transformVector): uv = q.vec x v; uv += uv;
// v + w*uv + q.vec x uv.  Not normalised.
inline V3 rotate(const Quat& q, V3 v) {
    V3 u{q.x, q.y, q.z};
    V3 uv = cross(u, v);
    uv = uv + uv;
    V3 t = cross(u, uv);
    return {(v.x + q.w * uv.x) + t.x, (v.y + q.w * uv.y) + t.y, (v.z + q.w * uv.z) + t.z};
}
// Hamilton product a*b (Eigen quat_product), storage (x,y,z,w).
inline Quat qmul(const Quat& a, const Quat& b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x,
            a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
// Eigen q.inverse(): conjugate / squaredNorm.
inline Quat qinv(const Quat& q) {
    double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    return {-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2};
}

// splitmix64 PRNG for the seeded synthetic inputs.
struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return (next() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
    double uniform(double a, double b) { return a + (b - a) * uniform(); }
    double normal() {
        double u1 = uniform();
        if (u1 < 1e-300) u1 = 1e-300;
        double u2 = uniform();
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

}  // namespace orc
