// Synthetic input generator (NOT part of the oracle and NOT part of the product path).
//
// Seeded procedural world + ray-cast LiDAR sensor models: no dataset exists
// offline (KITTI / MulRan are not in the container), so every workload is generated here, bit-identically
// for a given seed.  World frame: ground plane z = 0, sensor at height h, x forward, y left.
// Points are emitted firing by firing with clockwise azimuth, which matches the reference's
// `ori = -atan2(y, x)` convention (scanRegistration.cpp:143-146, :221).
#include "synth.h"
#include <omp.h>
#include <cmath>
#include <cstdint>
#include <vector>
#include <algorithm>

namespace synth {

struct V3 {
    double x, y, z;
};
struct Quat {
    double x, y, z, w;
};
static inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline V3 rotate(const Quat& q, V3 v) {
    V3 u{q.x, q.y, q.z};
    V3 uv = cross(u, v);
    uv = {uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};
    V3 t = cross(u, uv);
    return {(v.x + q.w * uv.x) + t.x, (v.y + q.w * uv.y) + t.y, (v.z + q.w * uv.z) + t.z};
}
static inline Quat qmul(const Quat& a, const Quat& b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
// splitmix64
struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
    double uniform(double a, double b) { return a + (b - a) * uniform(); }
    double normal() {
        double u1 = uniform();
        if (u1 < 1e-300) u1 = 1e-300;
        double u2 = uniform();
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};


struct Box {
    double x0, x1, y0, y1, h;
};
struct Cyl {
    double cx, cy, r, h;
};

struct World {
    SynthConfig cfg;
    std::vector<Box> boxes;
    std::vector<Cyl> cyls;
    std::vector<double> elev;  // beam elevations [rad], firing order
    int n_az;
    double max_range, height;

    void pose(int k, Quat& q, V3& t) const {
        // 10 Hz, 10 m/s, yaw rate 0.1 rad/s (radius 100 m), +-1 deg roll/pitch wobble
        const double tm = 0.1 * k, w = 0.1, R = 100.0;
        const double yaw = w * tm;
        const double roll = (M_PI / 180.0) * std::sin(0.7 * tm);
        const double pitch = (M_PI / 180.0) * std::sin(1.1 * tm + 0.5);
        t = {R * std::sin(yaw), R * (1.0 - std::cos(yaw)), height};
        // q = Rz(yaw) * Ry(pitch) * Rx(roll)
        Quat qz{0, 0, std::sin(yaw / 2), std::cos(yaw / 2)};
        Quat qy{0, std::sin(pitch / 2), 0, std::cos(pitch / 2)};
        Quat qx{std::sin(roll / 2), 0, 0, std::cos(roll / 2)};
        q = qmul(qmul(qz, qy), qx);
    }
};

static double dist_to_path(const World& w, double x, double y) {
    double best = 1e30;
    for (int k = 0; k <= 4000; k += 5) {  // path sampled every 0.5 m for 400 m
        Quat q;
        V3 t;
        w.pose(k, q, t);
        double d = std::hypot(x - t.x, y - t.y);
        best = std::min(best, d);
    }
    return best;
}

static World* make_world(const SynthConfig& c) {
    World* w = new World();
    w->cfg = c;
    switch (c.sensor) {
        case SYN_VLP16:
            for (int i = 0; i < 16; ++i) w->elev.push_back((-15.0 + 2.0 * i) * M_PI / 180.0);
            w->n_az = 1800, w->max_range = 100.0, w->height = 1.0;
            break;
        case SYN_HDL32:
            for (int i = 0; i < 32; ++i) w->elev.push_back((10.0 + 2.0 / 3.0 - (4.0 / 3.0) * i) * M_PI / 180.0);
            w->n_az = 1800, w->max_range = 100.0, w->height = 1.7;
            break;
        case SYN_HDL64:
            for (int i = 0; i < 32; ++i) w->elev.push_back((2.0 - i / 3.0) * M_PI / 180.0);
            for (int i = 0; i < 32; ++i) w->elev.push_back((-8.83 - 0.5 * i) * M_PI / 180.0);
            w->n_az = 1900, w->max_range = 120.0, w->height = 1.73;
            break;
        default:  // OS1-64
            for (int i = 0; i < 64; ++i) w->elev.push_back((16.6 - 33.2 * i / 63.0) * M_PI / 180.0);
            w->n_az = 1024, w->max_range = 120.0, w->height = 1.8;
            break;
    }
    SplitMix64 rng(c.seed);
    const double xmin = c.region[0], xmax = c.region[1], ymin = c.region[2], ymax = c.region[3];
    for (int i = 0; i < c.n_boxes; ++i) {
        for (int tries = 0; tries < 100; ++tries) {
            double sx = rng.uniform(4, 20), sy = rng.uniform(4, 20), h = rng.uniform(3, 15);
            double cx = rng.uniform(xmin, xmax), cy = rng.uniform(ymin, ymax);
            double rad = 0.5 * std::hypot(sx, sy);
            if (dist_to_path(*w, cx, cy) < rad + 3.0) continue;  // 6 m corridor
            w->boxes.push_back({cx - sx / 2, cx + sx / 2, cy - sy / 2, cy + sy / 2, h});
            break;
        }
    }
    for (int i = 0; i < c.n_cyl; ++i) {
        for (int tries = 0; tries < 100; ++tries) {
            double r = rng.uniform(0.1, 0.4), h = rng.uniform(2, 8);
            double cx = rng.uniform(xmin, xmax), cy = rng.uniform(ymin, ymax);
            if (dist_to_path(*w, cx, cy) < r + 3.0) continue;
            w->cyls.push_back({cx, cy, r, h});
            break;
        }
    }
    return w;
}

static int cast_scan(const World& w, const Quat& q, const V3& t, uint64_t noise_seed, float* out) {
    const int B = static_cast<int>(w.elev.size());
    const int n_rays = w.n_az * B;
    // cull objects by range
    std::vector<Box> boxes;
    std::vector<Cyl> cyls;
    for (const Box& b : w.boxes) {
        double dx = std::max({b.x0 - t.x, 0.0, t.x - b.x1}), dy = std::max({b.y0 - t.y, 0.0, t.y - b.y1});
        if (dx * dx + dy * dy < w.max_range * w.max_range) boxes.push_back(b);
    }
    for (const Cyl& c : w.cyls)
        if (std::hypot(c.cx - t.x, c.cy - t.y) < w.max_range + c.r) cyls.push_back(c);
    std::vector<float> rng_out(static_cast<size_t>(n_rays), -1.f);
    const int threads = w.cfg.threads > 0 ? w.cfg.threads : omp_get_max_threads();
    // Azimuth culling (pure speed-up, outputs unchanged): every object is a vertical prism over its footprint, so a ray can only
    // meet it when the ray's world azimuth lies inside the azimuth interval the footprint subtends from the sensor.  Per object:
    // centre and half-width of that interval (pi = always tested); per azimuth column: the interval its beams cover after the
    // rotation by q.  Objects are still tested in their original order, with the original arithmetic.
    auto wrap = [](double x) {
        while (x > M_PI) x -= 2.0 * M_PI;
        while (x <= -M_PI) x += 2.0 * M_PI;
        return x;
    };
    const double cull_eps = 1e-6;
    std::vector<double> box_c(boxes.size()), box_hw(boxes.size()), cyl_c(cyls.size()), cyl_hw(cyls.size());
    for (size_t i = 0; i < boxes.size(); ++i) {
        const Box& b = boxes[i];
        const bool near = t.x > b.x0 - 0.01 && t.x < b.x1 + 0.01 && t.y > b.y0 - 0.01 && t.y < b.y1 + 0.01;
        box_c[i] = std::atan2(0.5 * (b.y0 + b.y1) - t.y, 0.5 * (b.x0 + b.x1) - t.x);
        double hw = 0.0;
        const double cx[4] = {b.x0, b.x1, b.x0, b.x1}, cy[4] = {b.y0, b.y0, b.y1, b.y1};
        for (int k = 0; k < 4; ++k) hw = std::max(hw, std::fabs(wrap(std::atan2(cy[k] - t.y, cx[k] - t.x) - box_c[i])));
        box_hw[i] = near ? M_PI : hw + cull_eps;
    }
    for (size_t i = 0; i < cyls.size(); ++i) {
        const Cyl& c = cyls[i];
        const double dist = std::hypot(c.cx - t.x, c.cy - t.y);
        cyl_c[i] = std::atan2(c.cy - t.y, c.cx - t.x);
        cyl_hw[i] = dist <= 1.001 * c.r + 0.01 ? M_PI : std::asin(std::min(1.0, c.r / dist)) + cull_eps;
    }
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int a = 0; a < w.n_az; ++a) {
        const double phi = -2.0 * M_PI * a / w.n_az;  // clockwise
        // azimuth interval of this column's beams: reference + [lo, hi]; `all` = a beam points (almost) straight up or down
        double ref = 0.0, lo = 0.0, hi = 0.0;
        bool all = false;
        for (int b = 0; b < B; ++b) {
            const double e = w.elev[b];
            V3 ds{std::cos(e) * std::cos(phi), std::cos(e) * std::sin(phi), std::sin(e)};
            V3 d = rotate(q, ds);
            if (d.x * d.x + d.y * d.y < 1e-6) all = true;
            const double az = std::atan2(d.y, d.x);
            if (b == 0) ref = az;
            const double df = wrap(az - ref);
            lo = std::min(lo, df), hi = std::max(hi, df);
        }
        if (hi - lo > M_PI) all = true;
        const double mid = ref + 0.5 * (lo + hi), half = 0.5 * (hi - lo) + cull_eps;
        std::vector<int> bsel, csel;
        for (size_t i = 0; i < boxes.size(); ++i)
            if (all || box_hw[i] >= M_PI || std::fabs(wrap(box_c[i] - mid)) <= box_hw[i] + half) bsel.push_back(static_cast<int>(i));
        for (size_t i = 0; i < cyls.size(); ++i)
            if (all || cyl_hw[i] >= M_PI || std::fabs(wrap(cyl_c[i] - mid)) <= cyl_hw[i] + half) csel.push_back(static_cast<int>(i));
        for (int b = 0; b < B; ++b) {
            const double e = w.elev[b];
            V3 ds{std::cos(e) * std::cos(phi), std::cos(e) * std::sin(phi), std::sin(e)};
            V3 d = rotate(q, ds);
            double best = 1e30;
            if (d.z < -1e-9) best = -t.z / d.z;  // ground z = 0
            for (const int bi : bsel) {
                const Box& bx = boxes[bi];
                double t0 = 0.0, t1 = best;
                const double lo[3] = {bx.x0, bx.y0, 0.0}, hi[3] = {bx.x1, bx.y1, bx.h};
                const double o[3] = {t.x, t.y, t.z}, dd[3] = {d.x, d.y, d.z};
                bool hit = true;
                for (int ax = 0; ax < 3 && hit; ++ax) {
                    if (std::fabs(dd[ax]) < 1e-12) {
                        if (o[ax] < lo[ax] || o[ax] > hi[ax]) hit = false;
                    } else {
                        double ta = (lo[ax] - o[ax]) / dd[ax], tb = (hi[ax] - o[ax]) / dd[ax];
                        if (ta > tb) std::swap(ta, tb);
                        t0 = std::max(t0, ta);
                        t1 = std::min(t1, tb);
                        if (t0 > t1) hit = false;
                    }
                }
                if (hit && t0 > 0.05 && t0 < best) best = t0;
            }
            for (const int ci : csel) {
                const Cyl& c = cyls[ci];
                const double ox = t.x - c.cx, oy = t.y - c.cy;
                const double A = d.x * d.x + d.y * d.y;
                if (A < 1e-12) continue;
                const double Bq = ox * d.x + oy * d.y, Cq = ox * ox + oy * oy - c.r * c.r;
                const double disc = Bq * Bq - A * Cq;
                if (disc < 0) continue;
                const double tc = (-Bq - std::sqrt(disc)) / A;
                if (tc <= 0.05 || tc >= best) continue;
                const double z = t.z + tc * d.z;
                if (z < 0 || z > c.h) continue;
                best = tc;
            }
            if (best > 1e29) continue;
            SplitMix64 nr(noise_seed * 0x9E3779B97F4A7C15ull + static_cast<uint64_t>(a) * 64 + b + 1);
            nr.next();
            const double r = best + w.cfg.noise_sigma * nr.normal();
            if (r > w.max_range || r < 0.05) continue;
            rng_out[static_cast<size_t>(a) * B + b] = static_cast<float>(r);
        }
    }
    int m = 0;
    for (int a = 0; a < w.n_az; ++a) {
        const double phi = -2.0 * M_PI * a / w.n_az;
        for (int b = 0; b < B; ++b) {
            const float r = rng_out[static_cast<size_t>(a) * B + b];
            if (r < 0) continue;
            const double e = w.elev[b];
            out[3 * m + 0] = static_cast<float>(r * std::cos(e) * std::cos(phi));
            out[3 * m + 1] = static_cast<float>(r * std::cos(e) * std::sin(phi));
            out[3 * m + 2] = static_cast<float>(r * std::sin(e));
            ++m;
        }
    }
    return m;
}

}  // namespace synth

extern "C" {
void* syn_world_create(const SynthConfig* cfg) { return synth::make_world(*cfg); }
void syn_world_destroy(void* w) { delete static_cast<synth::World*>(w); }
void syn_world_pose(void* w, int k, double* q, double* t) {
    synth::Quat qq;
    synth::V3 tt;
    static_cast<synth::World*>(w)->pose(k, qq, tt);
    q[0] = qq.x, q[1] = qq.y, q[2] = qq.z, q[3] = qq.w;
    t[0] = tt.x, t[1] = tt.y, t[2] = tt.z;
}
int syn_world_max_points(void* w) {
    auto* W = static_cast<synth::World*>(w);
    return W->n_az * static_cast<int>(W->elev.size());
}
int syn_world_scan(void* w, int k, float* out_xyz) {
    auto* W = static_cast<synth::World*>(w);
    synth::Quat q;
    synth::V3 t;
    W->pose(k, q, t);
    return synth::cast_scan(*W, q, t, W->cfg.seed + 1000003ull * (k + 1), out_xyz);
}
int syn_world_scan_pose(void* w, const double* q, const double* t, uint64_t noise_seed, float* out_xyz) {
    auto* W = static_cast<synth::World*>(w);
    return synth::cast_scan(*W, {q[0], q[1], q[2], q[3]}, {t[0], t[1], t[2]}, noise_seed, out_xyz);
}
}
