// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.hpp).
//
// Exact k-nearest-neighbour search over a PointXYZI cloud, standing in for
// pcl::KdTreeFLANN<PointXYZI>::setInputCloud / nearestKSearch (laserOdometry.cpp:302, :390, :567-568;
// laserMapping.cpp:559-560, :583, :649).  Third-party (PCL 1.8 + FLANN, neither under /root/reference):
// FLANN KDTreeSingleIndex, leaf size 15, metric L2_Simple<float> (sum of diff*diff over x,y,z in f32,
// in that order), exact search (eps 0), results ascending.  A kd-tree with the same leaf size and the same
// f32 distance is built here so that the CPU baseline pays a comparable cost; the RESULT of an exact
// search does not depend on the tree layout.  Exact-distance ties are broken by the lower point index
// (FLANN's tie order is traversal dependent: unpinned).
#pragma once
#include "orc_common.hpp"

namespace orc {

struct KdTree {
    struct Node {
        int left, right;   // leaf: point range [left, right) in vind; inner: child node ids
        int dim;           // -1 for a leaf
        float divlow, divhigh;
    };
    const P4* pts = nullptr;
    int n = 0;
    std::vector<int> vind;
    std::vector<Node> nodes;
    float bb_lo[3], bb_hi[3];
    static constexpr int kLeaf = 15;

    static inline float coord(const P4& p, int d) { return d == 0 ? p.x : (d == 1 ? p.y : p.z); }
    static inline float dist2(const P4& a, const float* q) {  // L2_Simple<float>
        float result = 0.f;
        float diff = q[0] - a.x;
        result += diff * diff;
        diff = q[1] - a.y;
        result += diff * diff;
        diff = q[2] - a.z;
        result += diff * diff;
        return result;
    }

    void build(const P4* p, int count) {
        pts = p;
        n = count;
        vind.resize(n);
        for (int i = 0; i < n; ++i) vind[i] = i;
        nodes.clear();
        nodes.reserve(n / 4 + 8);
        for (int d = 0; d < 3; ++d) bb_lo[d] = std::numeric_limits<float>::max(), bb_hi[d] = -std::numeric_limits<float>::max();
        for (int i = 0; i < n; ++i)
            for (int d = 0; d < 3; ++d) {
                bb_lo[d] = std::min(bb_lo[d], coord(pts[i], d));
                bb_hi[d] = std::max(bb_hi[d], coord(pts[i], d));
            }
        if (n > 0) {
            float lo[3] = {bb_lo[0], bb_lo[1], bb_lo[2]}, hi[3] = {bb_hi[0], bb_hi[1], bb_hi[2]};
            divide(0, n, lo, hi);
        }
    }

    int divide(int l, int r, float* lo, float* hi) {
        const int id = static_cast<int>(nodes.size());
        nodes.push_back({});
        if (r - l <= kLeaf) {
            nodes[id] = {l, r, -1, 0.f, 0.f};
            for (int d = 0; d < 3; ++d) lo[d] = std::numeric_limits<float>::max(), hi[d] = -std::numeric_limits<float>::max();
            for (int i = l; i < r; ++i)
                for (int d = 0; d < 3; ++d) {
                    lo[d] = std::min(lo[d], coord(pts[vind[i]], d));
                    hi[d] = std::max(hi[d], coord(pts[vind[i]], d));
                }
            return id;
        }
        int cut = 0;
        float span = -1.f;
        for (int d = 0; d < 3; ++d)
            if (hi[d] - lo[d] > span) span = hi[d] - lo[d], cut = d;
        const int mid = (l + r) / 2;
        std::nth_element(vind.begin() + l, vind.begin() + mid, vind.begin() + r,
                         [&](int a, int b) { return coord(pts[a], cut) < coord(pts[b], cut); });
        float llo[3] = {lo[0], lo[1], lo[2]}, lhi[3] = {hi[0], hi[1], hi[2]};
        float rlo[3] = {lo[0], lo[1], lo[2]}, rhi[3] = {hi[0], hi[1], hi[2]};
        const int cl = divide(l, mid, llo, lhi);
        const int cr = divide(mid, r, rlo, rhi);
        nodes[id] = {cl, cr, cut, lhi[cut], rlo[cut]};
        for (int d = 0; d < 3; ++d) lo[d] = std::min(llo[d], rlo[d]), hi[d] = std::max(lhi[d], rhi[d]);
        return id;
    }

    struct Result {
        int k, count;
        int* idx;
        float* d;
        float worst() const { return count < k ? std::numeric_limits<float>::max() : d[k - 1]; }
        void add(float dist, int index) {
            if (count == k && !(dist < d[k - 1] || (dist == d[k - 1] && index < idx[k - 1]))) return;
            int i = count < k ? count : k - 1;
            while (i > 0 && (d[i - 1] > dist || (d[i - 1] == dist && idx[i - 1] > index))) {
                d[i] = d[i - 1];
                idx[i] = idx[i - 1];
                --i;
            }
            d[i] = dist;
            idx[i] = index;
            if (count < k) ++count;
        }
    };

    void search(int node, const float* q, double mindist, double* off, Result& res) const {
        const Node& nd = nodes[node];
        if (nd.dim < 0) {
            for (int i = nd.left; i < nd.right; ++i) {
                const int id = vind[i];
                res.add(dist2(pts[id], q), id);
            }
            return;
        }
        const int d = nd.dim;
        const double val = q[d];
        const double diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
        int best, other;
        double cut;
        if (diff1 + diff2 < 0) {
            best = nd.left, other = nd.right;
            cut = diff2 * diff2;
        } else {
            best = nd.right, other = nd.left;
            cut = diff1 * diff1;
        }
        search(best, q, mindist, off, res);
        const double saved = off[d];
        const double md = mindist + cut - saved;
        off[d] = cut;
        // conservative pruning: the f32 distance of a point may round below the exact lower bound
        if (md * (1.0 - 1e-6) <= static_cast<double>(res.worst())) search(other, q, md, off, res);
        off[d] = saved;
    }

    // returns number found (<= k); idx/d ascending; unfilled slots get idx 0 / FLT_MAX
    int knn(const float* q, int k, int* idx, float* d) const {
        Result res{k, 0, idx, d};
        for (int i = 0; i < k; ++i) idx[i] = 0, d[i] = std::numeric_limits<float>::max();
        if (n == 0) return 0;
        double off[3] = {0, 0, 0}, mind = 0;
        for (int a = 0; a < 3; ++a) {
            if (q[a] < bb_lo[a]) off[a] = (double(q[a]) - bb_lo[a]) * (double(q[a]) - bb_lo[a]);
            if (q[a] > bb_hi[a]) off[a] = (double(q[a]) - bb_hi[a]) * (double(q[a]) - bb_hi[a]);
            mind += off[a];
        }
        search(0, q, mind, off, res);
        return res.count;
    }

    int knn_brute(const float* q, int k, int* idx, float* d) const {
        Result res{k, 0, idx, d};
        for (int i = 0; i < k; ++i) idx[i] = 0, d[i] = std::numeric_limits<float>::max();
        for (int i = 0; i < n; ++i) res.add(dist2(pts[i], q), i);
        return res.count;
    }
};

}  // namespace orc
