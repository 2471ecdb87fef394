// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.hpp).
//
// Stage D: literal CPU restatement of SCManager, /root/reference/include/scancontext/Scancontext.cpp
// (:23-66 helpers, :69-148 distances, :151-260 descriptor/keys/insert, :336-427 detectLoopClosureID)
// with the constants of Scancontext.h:83-104.  Descriptors are 20x60 doubles, COLUMN-MAJOR
// (Eigen::MatrixXd default): desc[r + 20*c].
//
// Third-party pieces restated here (not under /root/reference except nanoflann):
//  * Eigen reductions (mean/norm/dot, Scancontext.cpp:78-81, :103, :207, :223): Eigen 3.3's default
//    linear-vectorised redux on x86-64/SSE2 sums four interleaved partial sums s_j = sum_{i%4==j} x_i and
//    returns (s0+s2)+(s1+s3); restated by eigen_sum() below.  Parity unpinned by any reference test.
//  * nanoflann KNN (vendored, include/scancontext/nanoflann.hpp:375-408 L2_Adaptor::evalMetric,
//    :142-205 KNNResultSet): exact 3-NN in f32; restated as brute force with the same f32 accumulation.
//    oracle/_ref/ compiles the vendored nanoflann itself and tests/ compares the two (pins D5).
#include "orc_common.hpp"
#include "oracle.h"
#include <memory>

namespace orc {

static const int NR = 20, NS = 60;  // PC_NUM_RING, PC_NUM_SECTOR (Scancontext.h:85-86)

// Eigen 3.3 redux_impl<LinearVectorizedTraversal, NoUnrolling>, packet size 2 (SSE2), aligned start 0.
template <class F>
static double eigen_sum(int size, F at) {
    const int packetSize = 2;
    const int alignedSize2 = (size / (2 * packetSize)) * (2 * packetSize);
    const int alignedSize = (size / packetSize) * packetSize;
    double res;
    if (alignedSize) {
        double p0[2] = {at(0), at(1)};
        if (alignedSize > packetSize) {
            double p1[2] = {at(2), at(3)};
            for (int index = 2 * packetSize; index < alignedSize2; index += 2 * packetSize) {
                p0[0] += at(index), p0[1] += at(index + 1);
                p1[0] += at(index + 2), p1[1] += at(index + 3);
            }
            p0[0] += p1[0], p0[1] += p1[1];
            if (alignedSize > alignedSize2) p0[0] += at(alignedSize2), p0[1] += at(alignedSize2 + 1);
        }
        res = p0[0] + p0[1];
        for (int index = alignedSize; index < size; ++index) res += at(index);
    } else {
        res = at(0);
        for (int index = 1; index < size; ++index) res += at(index);
    }
    return res;
}

struct SCMath {
    int float_math, cr_libm;
    // xy2theta, Scancontext.cpp:23-36 (returns float)
    float xy2theta(float x, float y) const {
        auto at = [&](float v) -> double {
            if (!float_math) return std::atan(static_cast<double>(v));  // ::atan(double)
            return cr_libm ? static_cast<double>(static_cast<float>(std::atan(static_cast<double>(v)))) : static_cast<double>(atanf(v));
        };
        if ((x >= 0) & (y >= 0)) return static_cast<float>((180 / M_PI) * at(y / x));
        if ((x < 0) & (y >= 0)) return static_cast<float>(180 - ((180 / M_PI) * at(y / (-x))));
        if ((x < 0) & (y < 0)) return static_cast<float>(180 + ((180 / M_PI) * at(y / x)));
        if ((x >= 0) & (y < 0)) return static_cast<float>(360 - ((180 / M_PI) * at((-y) / x)));
        return std::numeric_limits<float>::quiet_NaN();  // NaN input: the reference falls off the end (UB)
    }
};

static inline int ceil_to_int(double v) {  // int(ceil(v)); NaN -> INT_MIN as cvttsd2si does on x86-64
    double c = std::ceil(v);
    if (!(c == c)) return std::numeric_limits<int>::min();
    return static_cast<int>(c);
}

// makeScancontext, Scancontext.cpp:151-195
static void make_sc(const SCMath& m, double max_radius, const P4* pts, int n, double* desc) {
    const int NO_POINT = -1000;
    for (int i = 0; i < NR * NS; ++i) desc[i] = NO_POINT;
    const double LIDAR_HEIGHT = 2.0;  // Scancontext.h:83
    for (int k = 0; k < n; ++k) {
        const float px = pts[k].x, py = pts[k].y;
        const float pz = static_cast<float>(pts[k].z + LIDAR_HEIGHT);  // :168
        // :171  sqrt of the float sum; double-rounding through ::sqrt(double) is exact for sqrt, so both
        // overloads give the same float.
        const float azim_range = sqrtf(px * px + py * py);
        const float azim_angle = m.xy2theta(px, py);
        if (azim_range > max_radius) continue;  // :175
        int ring_idx = std::max(std::min(NR, ceil_to_int((azim_range / max_radius) * NR)), 1);
        int sctor_idx = std::max(std::min(NS, ceil_to_int((azim_angle / 360.0) * NS)), 1);
        double& cell = desc[(ring_idx - 1) + NR * (sctor_idx - 1)];
        if (cell < pz) cell = pz;  // :182-183
    }
    for (int i = 0; i < NR * NS; ++i)
        if (desc[i] == NO_POINT) desc[i] = 0;  // :187-190
}

// makeRingkeyFromScancontext :198-211 (row means), makeSectorkeyFromScancontext :214-227 (column means)
static void make_keys(const double* desc, double* ringkey, double* sectorkey) {
    for (int r = 0; r < NR; ++r) ringkey[r] = eigen_sum(NS, [&](int c) { return desc[r + NR * c]; }) / NS;
    for (int c = 0; c < NS; ++c) sectorkey[c] = eigen_sum(NR, [&](int r) { return desc[r + NR * c]; }) / NR;
}

// distDirectSC :69-90 on (sc1, circshift(sc2, shift)) without materialising the shifted copy:
// circshift :39-59 moves column c of sc2 to (c + shift) % 60, so shifted.col(j) = sc2.col((j - shift) mod 60).
static double dist_direct(const double* sc1, const double* sc2, int shift) {
    int num_eff_cols = 0;
    double sum_sector_similarity = 0;
    for (int col = 0; col < NS; ++col) {
        const double* a = sc1 + NR * col;
        const double* b = sc2 + NR * ((col - shift + NS) % NS);
        const double na = std::sqrt(eigen_sum(NR, [&](int i) { return a[i] * a[i]; }));
        const double nb = std::sqrt(eigen_sum(NR, [&](int i) { return b[i] * b[i]; }));
        if ((na == 0) | (nb == 0)) continue;
        const double sim = eigen_sum(NR, [&](int i) { return a[i] * b[i]; }) / (na * nb);
        sum_sector_similarity = sum_sector_similarity + sim;
        num_eff_cols = num_eff_cols + 1;
    }
    const double sc_sim = sum_sector_similarity / num_eff_cols;
    return 1.0 - sc_sim;
}

// fastAlignUsingVkey :93-113
static int fast_align(const double* vkey1, const double* vkey2) {
    int argmin = 0;
    double min_norm = 10000000;
    for (int shift = 0; shift < NS; ++shift) {
        const double nrm = std::sqrt(eigen_sum(NS, [&](int j) {
            const double d = vkey1[j] - vkey2[(j - shift + NS) % NS];
            return d * d;
        }));
        if (nrm < min_norm) {
            argmin = shift;
            min_norm = nrm;
        }
    }
    return argmin;
}

// distanceBtnScanContext :116-148
static void sc_distance(const double* sc1, const double* sc2, double* dist, int* shift_out) {
    double rk[NR], v1[NS], v2[NS];
    make_keys(sc1, rk, v1);
    make_keys(sc2, rk, v2);
    const int argmin_vkey_shift = fast_align(v1, v2);
    const double SEARCH_RATIO = 0.1;                                          // Scancontext.h:96
    const int SEARCH_RADIUS = static_cast<int>(std::round(0.5 * SEARCH_RATIO * NS));  // :123
    std::vector<int> space{argmin_vkey_shift};
    for (int ii = 1; ii < SEARCH_RADIUS + 1; ii++) {
        space.push_back((argmin_vkey_shift + ii + NS) % NS);
        space.push_back((argmin_vkey_shift - ii + NS) % NS);
    }
    std::sort(space.begin(), space.end());
    int argmin_shift = 0;
    double min_sc_dist = 10000000;
    for (int s : space) {
        const double d = dist_direct(sc1, sc2, s);
        if (d < min_sc_dist) {
            argmin_shift = s;
            min_sc_dist = d;
        }
    }
    *dist = min_sc_dist;
    *shift_out = argmin_shift;
}

struct SCManager {
    OrcSCConfig cfg;
    SCMath math;
    // Scancontext.h:112-117
    std::vector<std::vector<double>> polarcontexts_;
    std::vector<std::vector<float>> invkeys_mat_;
    std::vector<std::vector<float>> invkeys_to_search_;
    bool tree_made = false;
    int tree_making_period_conter = 0;
    const int NUM_EXCLUDE_RECENT = 30, NUM_CANDIDATES_FROM_TREE = 3, TREE_MAKING_PERIOD_ = 30;

    void insert_desc(const double* desc) {  // :236-260
        double rk[NR], sk[NS];
        make_keys(desc, rk, sk);
        polarcontexts_.emplace_back(desc, desc + NR * NS);
        std::vector<float> kf(NR);
        for (int i = 0; i < NR; ++i) kf[i] = static_cast<float>(rk[i]);  // eig2stdvec :62-66
        invkeys_mat_.push_back(kf);
    }

    // nanoflann L2_Adaptor<float>::evalMetric, nanoflann.hpp:383-408 (dim 20: five groups of four)
    static float key_dist(const float* a, const float* b) {
        float result = 0.f;
        for (int g = 0; g < NR; g += 4) {
            const float d0 = a[g] - b[g], d1 = a[g + 1] - b[g + 1], d2 = a[g + 2] - b[g + 2], d3 = a[g + 3] - b[g + 3];
            result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
        return result;
    }

    int detect(float* yaw, double* min_dist_out, int* nn_idx_out, int* cand_out, float* cand_d) {  // :336-427
        int loop_id = -1;
        *yaw = 0.f;
        *min_dist_out = 10000000;
        *nn_idx_out = 0;
        for (int i = 0; i < 3; ++i) cand_out[i] = 0, cand_d[i] = 0.f;
        const std::vector<float>& curr_key = invkeys_mat_.back();
        const std::vector<double>& curr_desc = polarcontexts_.back();
        if (static_cast<int>(invkeys_mat_.size()) < NUM_EXCLUDE_RECENT + 1) return -1;  // :346-350
        if (tree_making_period_conter % TREE_MAKING_PERIOD_ == 0) {                     // :353-364
            invkeys_to_search_.assign(invkeys_mat_.begin(), invkeys_mat_.end() - NUM_EXCLUDE_RECENT);
            tree_made = true;
        }
        tree_making_period_conter = tree_making_period_conter + 1;

        double min_dist = 10000000;
        int nn_align = 0, nn_idx = 0;
        // KNNResultSet<float> (nanoflann.hpp:142-205): capacity 3, ascending, strict '>' insertion;
        // index/dist vectors are zero-initialised by the caller (:372-373) so unused slots read 0.
        size_t cand[3] = {0, 0, 0};
        float dists[3] = {0.f, 0.f, 0.f};
        int count = 0;
        const int capacity = NUM_CANDIDATES_FROM_TREE;
        auto worst = [&]() { return count < capacity ? std::numeric_limits<float>::max() : dists[capacity - 1]; };
        if (count < capacity) dists[capacity - 1] = std::numeric_limits<float>::max();  // init()
        for (size_t idx = 0; idx < invkeys_to_search_.size(); ++idx) {
            const float dist = key_dist(curr_key.data(), invkeys_to_search_[idx].data());
            if (!(dist < dists[capacity - 1])) continue;  // searchLevel: dist < worst_dist
            int i;
            for (i = count; i > 0; --i) {
                if (dists[i - 1] > dist) {
                    if (i < capacity) {
                        dists[i] = dists[i - 1];
                        cand[i] = cand[i - 1];
                    }
                } else
                    break;
            }
            if (i < capacity) {
                dists[i] = dist;
                cand[i] = idx;
            }
            if (count < capacity) count++;
        }
        (void)worst;
        if (count < capacity) {  // slots never written keep the caller's zero-init (:372-373)
            for (int i = count; i < capacity; ++i) cand[i] = 0, dists[i] = (i == capacity - 1) ? std::numeric_limits<float>::max() : 0.f;
        }
        for (int it = 0; it < NUM_CANDIDATES_FROM_TREE; it++) {  // :385-400
            const std::vector<double>& cnd = polarcontexts_[cand[it]];
            double d;
            int al;
            sc_distance(curr_desc.data(), cnd.data(), &d, &al);
            if (d < min_dist) {
                min_dist = d;
                nn_align = al;
                nn_idx = static_cast<int>(cand[it]);
            }
            cand_out[it] = static_cast<int>(cand[it]);
            cand_d[it] = dists[it];
        }
        if (min_dist < cfg.dist_thres) loop_id = nn_idx;  // :406-408
        const double PC_UNIT_SECTORANGLE = 360.0 / double(NS);
        // deg2rad(float) :17-20: degrees * M_PI / 180.0 -> float
        const float deg = static_cast<float>(nn_align * PC_UNIT_SECTORANGLE);
        *yaw = static_cast<float>(deg * M_PI / 180.0);
        *min_dist_out = min_dist;
        *nn_idx_out = nn_idx;
        return loop_id;
    }
};

}  // namespace orc

extern "C" {
void* orc_sc_create(const OrcSCConfig* cfg) {
    auto* m = new orc::SCManager();
    m->cfg = *cfg;
    m->math = {cfg->float_math, cfg->cr_libm};
    return m;
}
void orc_sc_destroy(void* h) { delete static_cast<orc::SCManager*>(h); }
int orc_sc_size(void* h) { return static_cast<int>(static_cast<orc::SCManager*>(h)->polarcontexts_.size()); }
void orc_sc_make(void* h, const float* xyzi, int n, double* desc) {
    auto* m = static_cast<orc::SCManager*>(h);
    orc::make_sc(m->math, m->cfg.max_radius, reinterpret_cast<const orc::P4*>(xyzi), n, desc);
}
void orc_sc_keys(const double* desc, double* ringkey20, double* sectorkey60) { orc::make_keys(desc, ringkey20, sectorkey60); }
void orc_sc_insert_cloud(void* h, const float* xyzi, int n) {
    auto* m = static_cast<orc::SCManager*>(h);
    double desc[orc::NR * orc::NS];
    orc::make_sc(m->math, m->cfg.max_radius, reinterpret_cast<const orc::P4*>(xyzi), n, desc);
    m->insert_desc(desc);
}
void orc_sc_insert_desc(void* h, const double* desc) { static_cast<orc::SCManager*>(h)->insert_desc(desc); }
void orc_sc_get(void* h, int idx, double* desc, float* ringkey20) {
    auto* m = static_cast<orc::SCManager*>(h);
    if (desc) std::memcpy(desc, m->polarcontexts_[idx].data(), sizeof(double) * orc::NR * orc::NS);
    if (ringkey20) std::memcpy(ringkey20, m->invkeys_mat_[idx].data(), sizeof(float) * orc::NR);
}
void orc_sc_distance(const double* sc1, const double* sc2, double* dist, int* shift) { orc::sc_distance(sc1, sc2, dist, shift); }
void orc_sc_distance_full(const double* sc1, const double* sc2, double* dist60) {
    for (int s = 0; s < orc::NS; ++s) dist60[s] = orc::dist_direct(sc1, sc2, s);
}
int orc_sc_detect(void* h, float* yaw, double* min_dist, int* nn_idx, int* cand, float* cand_d) {
    return static_cast<orc::SCManager*>(h)->detect(yaw, min_dist, nn_idx, cand, cand_d);
}
}
