// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.hpp).
//
// Stage C: literal CPU restatement of process(), /root/reference/src/laserMapping.cpp:232-906 minus ROS
// plumbing: pose algebra :143-174, rolling 21x21x11 cube window :313-508, 5x5x3 gather :510-540, stack
// downsample :543-551, kNN(5)+PCA edge :578-623, kNN(5)+QR plane :644-688, Ceres solve x2 :563-728,
// transformUpdate :735, map insert :738-784, per-cube downsample :789-802, full-res registration :845-849.
// Lock-step: every call maps one scan (the reference drops queued scans to stay real-time, :300-304).
//
// Third-party arithmetic restated (not under /root/reference):
//  * Eigen::SelfAdjointEigenSolver<Matrix3d> (:606) -> cyclic Jacobi in f64 (eigenvalues ascending);
//  * colPivHouseholderQr().solve (:664)            -> column-pivoted Householder QR least squares;
//  * pcl::KdTreeFLANN (kdtree.hpp), pcl::VoxelGrid (voxel.cpp), ceres::Solve (lm.cpp).
// PARITY UNPINNED at each of those boundaries (the reference has no tests).
#include "orc_common.hpp"
#include "oracle.h"
#include "kdtree.hpp"
#include "lm.hpp"
#include <chrono>

namespace orc {

int voxel_grid(const P4* in, int n, float leaf, int order_mode, P4* out, int* guard_hit);

// symmetric 3x3 eigen-decomposition, cyclic Jacobi; w ascending, V columns = unit eigenvectors
void eig3_sym(const double A[9], double w[3], double V[9]) {
    double a[3][3] = {{A[0], A[1], A[2]}, {A[3], A[4], A[5]}, {A[6], A[7], A[8]}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        const double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (off <= 1e-40 * diag || off == 0.0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {  // A <- A J
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {  // A <- J^T A
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int order[3] = {0, 1, 2};
    std::sort(order, order + 3, [&](int i, int j) { return a[i][i] < a[j][j]; });
    for (int j = 0; j < 3; ++j) {
        w[j] = a[order[j]][order[j]];
        double nrm = 0;
        for (int k = 0; k < 3; ++k) nrm += v[k][order[j]] * v[k][order[j]];
        nrm = std::sqrt(nrm);
        for (int k = 0; k < 3; ++k) V[3 * k + j] = v[k][order[j]] / nrm;
    }
}

// least squares solve of the 5x3 system A n = b by column-pivoted Householder QR (Eigen ColPivHouseholderQR)
void colpiv_qr_solve_5x3(const double A_in[15], const double b_in[5], double x[3]) {
    double A[5][3], b[5];
    for (int i = 0; i < 5; ++i) {
        for (int j = 0; j < 3; ++j) A[i][j] = A_in[3 * i + j];
        b[i] = b_in[i];
    }
    int perm[3] = {0, 1, 2};
    double rdiag[3] = {0, 0, 0};
    double maxpivot = 0;
    for (int k = 0; k < 3; ++k) {
        int best = k;
        double bestn = -1;
        for (int j = k; j < 3; ++j) {
            double s = 0;
            for (int i = k; i < 5; ++i) s += A[i][j] * A[i][j];
            if (s > bestn) bestn = s, best = j;
        }
        if (best != k) {
            for (int i = 0; i < 5; ++i) std::swap(A[i][k], A[i][best]);
            std::swap(perm[k], perm[best]);
        }
        double tail = 0;
        for (int i = k + 1; i < 5; ++i) tail += A[i][k] * A[i][k];
        const double c0 = A[k][k];
        double tau, beta;
        if (tail <= std::numeric_limits<double>::min()) {
            tau = 0, beta = c0;
            for (int i = k + 1; i < 5; ++i) A[i][k] = 0;
        } else {
            beta = std::sqrt(c0 * c0 + tail);
            if (c0 >= 0) beta = -beta;
            for (int i = k + 1; i < 5; ++i) A[i][k] /= (c0 - beta);
            tau = (beta - c0) / beta;
        }
        A[k][k] = beta;
        rdiag[k] = beta;
        maxpivot = std::max(maxpivot, std::fabs(beta));
        if (tau != 0) {
            for (int c = k + 1; c < 3; ++c) {
                double s = A[k][c];
                for (int i = k + 1; i < 5; ++i) s += A[i][k] * A[i][c];
                s *= tau;
                A[k][c] -= s;
                for (int i = k + 1; i < 5; ++i) A[i][c] -= s * A[i][k];
            }
            double s = b[k];
            for (int i = k + 1; i < 5; ++i) s += A[i][k] * b[i];
            s *= tau;
            b[k] -= s;
            for (int i = k + 1; i < 5; ++i) b[i] -= s * A[i][k];
        }
    }
    const double thr = std::numeric_limits<double>::epsilon() * 3.0 * maxpivot;
    int rank = 0;
    for (int k = 0; k < 3; ++k)
        if (std::fabs(rdiag[k]) > thr) ++rank;
    double y[3] = {0, 0, 0};
    for (int k = rank - 1; k >= 0; --k) {
        double s = b[k];
        for (int c = k + 1; c < rank; ++c) s -= A[k][c] * y[c];
        y[k] = s / A[k][k];
    }
    for (int k = 0; k < 3; ++k) x[perm[k]] = y[k];
}

struct Mapper {
    OrcMapConfig cfg;
    // laserMapping.cpp:74-82
    int cenW = 10, cenH = 10, cenD = 5;
    static constexpr int W = 21, H = 21, D = 11, NUM = W * H * D;
    std::vector<std::vector<P4>*> cornerArr, surfArr;
    std::vector<std::vector<P4>> storage;
    int validInd[125];
    int validNum = 0;
    double parameters[7] = {0, 0, 0, 1, 0, 0, 0};  // q_w_curr (x,y,z,w), t_w_curr   :110-112
    Quat q_wmap_wodom{0, 0, 0, 1};                  // :116-117
    V3 t_wmap_wodom{0, 0, 0};
    Quat q_wodom_curr{0, 0, 0, 1};
    V3 t_wodom_curr{0, 0, 0};
    std::vector<P4> cornerFromMap, surfFromMap;
    KdTree kdCorner, kdSurf;

    explicit Mapper(const OrcMapConfig& c) : cfg(c) {
        storage.resize(2 * NUM);
        cornerArr.resize(NUM);
        surfArr.resize(NUM);
        for (int i = 0; i < NUM; ++i) {
            cornerArr[i] = &storage[i];
            surfArr[i] = &storage[NUM + i];
        }
    }
    Quat q_w_curr() const { return {parameters[0], parameters[1], parameters[2], parameters[3]}; }
    V3 t_w_curr() const { return {parameters[4], parameters[5], parameters[6]}; }

    void pointAssociateToMap(const P4& pi, P4& po) const {  // :155-164
        V3 pw = rotate(q_w_curr(), V3{pi.x, pi.y, pi.z}) + t_w_curr();
        po.x = static_cast<float>(pw.x);
        po.y = static_cast<float>(pw.y);
        po.z = static_cast<float>(pw.z);
        po.i = pi.i;
    }

    static int idx(int i, int j, int k) { return i + W * j + W * H * k; }

    // one of the six while-blocks of :324-508: roll the pointer arrays by one cube along `axis`
    // (dir=+1: contents move to higher index) and clear the wrapped slab.
    void shift(int axis, int dir) {
        const int n[3] = {W, H, D};
        const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
        for (int u = 0; u < n[a1]; ++u)
            for (int v = 0; v < n[a2]; ++v) {
                auto at = [&](int s) {
                    int c[3];
                    c[axis] = s, c[a1] = u, c[a2] = v;
                    return idx(c[0], c[1], c[2]);
                };
                if (dir > 0) {
                    std::vector<P4>* cp = cornerArr[at(n[axis] - 1)];
                    std::vector<P4>* sp = surfArr[at(n[axis] - 1)];
                    for (int s = n[axis] - 1; s >= 1; --s) {
                        cornerArr[at(s)] = cornerArr[at(s - 1)];
                        surfArr[at(s)] = surfArr[at(s - 1)];
                    }
                    cornerArr[at(0)] = cp;
                    surfArr[at(0)] = sp;
                    cp->clear();
                    sp->clear();
                } else {
                    std::vector<P4>* cp = cornerArr[at(0)];
                    std::vector<P4>* sp = surfArr[at(0)];
                    for (int s = 0; s < n[axis] - 1; ++s) {
                        cornerArr[at(s)] = cornerArr[at(s + 1)];
                        surfArr[at(s)] = surfArr[at(s + 1)];
                    }
                    cornerArr[at(n[axis] - 1)] = cp;
                    surfArr[at(n[axis] - 1)] = sp;
                    cp->clear();
                    sp->clear();
                }
            }
    }

    int step(const P4* cornerLast, int nCorner, const P4* surfLast, int nSurf, const P4* fullRes, int nFull,
             const double* q_wodom, const double* t_wodom, double* q_out, double* t_out, P4* registered, OrcMapStats* st) {
        using clk = std::chrono::steady_clock;
        auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        OrcMapStats S;
        std::memset(&S, 0, sizeof S);
        auto t0 = clk::now();
        q_wodom_curr = {q_wodom[0], q_wodom[1], q_wodom[2], q_wodom[3]};  // :291-297
        t_wodom_curr = {t_wodom[0], t_wodom[1], t_wodom[2]};
        {  // transformAssociateToMap :143-147
            Quat q = qmul(q_wmap_wodom, q_wodom_curr);
            V3 t = rotate(q_wmap_wodom, t_wodom_curr) + t_wmap_wodom;
            parameters[0] = q.x, parameters[1] = q.y, parameters[2] = q.z, parameters[3] = q.w;
            parameters[4] = t.x, parameters[5] = t.y, parameters[6] = t.z;
        }
        // :313-322
        int centerCubeI = int((parameters[4] + 25.0) / 50.0) + cenW;
        int centerCubeJ = int((parameters[5] + 25.0) / 50.0) + cenH;
        int centerCubeK = int((parameters[6] + 25.0) / 50.0) + cenD;
        if (parameters[4] + 25.0 < 0) centerCubeI--;
        if (parameters[5] + 25.0 < 0) centerCubeJ--;
        if (parameters[6] + 25.0 < 0) centerCubeK--;
        // :324-508
        while (centerCubeI < 3) shift(0, +1), centerCubeI++, cenW++;
        while (centerCubeI >= W - 3) shift(0, -1), centerCubeI--, cenW--;
        while (centerCubeJ < 3) shift(1, +1), centerCubeJ++, cenH++;
        while (centerCubeJ >= H - 3) shift(1, -1), centerCubeJ--, cenH--;
        while (centerCubeK < 3) shift(2, +1), centerCubeK++, cenD++;
        while (centerCubeK >= D - 3) shift(2, -1), centerCubeK--, cenD--;
        // :510-540
        validNum = 0;
        for (int i = centerCubeI - 2; i <= centerCubeI + 2; i++)
            for (int j = centerCubeJ - 2; j <= centerCubeJ + 2; j++)
                for (int k = centerCubeK - 1; k <= centerCubeK + 1; k++)
                    if (i >= 0 && i < W && j >= 0 && j < H && k >= 0 && k < D) validInd[validNum++] = idx(i, j, k);
        cornerFromMap.clear();
        surfFromMap.clear();
        for (int i = 0; i < validNum; i++) {
            cornerFromMap.insert(cornerFromMap.end(), cornerArr[validInd[i]]->begin(), cornerArr[validInd[i]]->end());
            surfFromMap.insert(surfFromMap.end(), surfArr[validInd[i]]->begin(), surfArr[validInd[i]]->end());
        }
        const int nCornerMap = static_cast<int>(cornerFromMap.size()), nSurfMap = static_cast<int>(surfFromMap.size());
        auto t1 = clk::now();
        // :543-551
        std::vector<P4> cornerStack(nCorner + 1), surfStack(nSurf + 1);
        const int nCornerStack = voxel_grid(cornerLast, nCorner, cfg.line_res, cfg.voxel_order, cornerStack.data(), nullptr);
        const int nSurfStack = voxel_grid(surfLast, nSurf, cfg.plane_res, cfg.voxel_order, surfStack.data(), nullptr);
        auto t2 = clk::now();
        S.n_corner_stack = nCornerStack, S.n_surf_stack = nSurfStack, S.n_corner_map = nCornerMap, S.n_surf_map = nSurfMap;
        double t_tree = 0, t_assoc = 0, t_solve = 0;
        if (nCornerMap > 10 && nSurfMap > 50) {  // :555
            auto ta = clk::now();
            kdCorner.build(cornerFromMap.data(), nCornerMap);  // :559-560
            kdSurf.build(surfFromMap.data(), nSurfMap);
            t_tree = ms(ta, clk::now());
            S.solved = 1;
            std::vector<Factor> factors;
            for (int iterCount = 0; iterCount < 2; iterCount++) {  // :563
                auto tb = clk::now();
                factors.clear();
                int corner_num = 0, surf_num = 0;
                int ind[5];
                float sq[5];
                for (int i = 0; i < nCornerStack; i++) {  // :578-623
                    const P4 pointOri = cornerStack[i];
                    P4 pointSel;
                    pointAssociateToMap(pointOri, pointSel);
                    const float qv[3] = {pointSel.x, pointSel.y, pointSel.z};
                    if (cfg.knn_mode == 0)
                        kdCorner.knn(qv, 5, ind, sq);
                    else
                        kdCorner.knn_brute(qv, 5, ind, sq);
                    if (sq[4] < 1.0) {
                        V3 near[5];
                        V3 center{0, 0, 0};
                        for (int j = 0; j < 5; j++) {
                            near[j] = {cornerFromMap[ind[j]].x, cornerFromMap[ind[j]].y, cornerFromMap[ind[j]].z};
                            center = center + near[j];
                        }
                        center = center / 5.0;
                        double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                        for (int j = 0; j < 5; j++) {
                            const V3 z = near[j] - center;
                            const double zz[3] = {z.x, z.y, z.z};
                            for (int r = 0; r < 3; ++r)
                                for (int c = 0; c < 3; ++c) cov[3 * r + c] = cov[3 * r + c] + zz[r] * zz[c];
                        }
                        double w[3], V[9];
                        eig3_sym(cov, w, V);
                        const V3 unit_direction{V[2], V[5], V[8]};
                        if (w[2] > 3 * w[1]) {  // :612
                            Factor f;
                            f.kind = 0;
                            f.cp = {pointOri.x, pointOri.y, pointOri.z};
                            f.a = {0.1 * unit_direction.x + center.x, 0.1 * unit_direction.y + center.y, 0.1 * unit_direction.z + center.z};
                            f.b = {-0.1 * unit_direction.x + center.x, -0.1 * unit_direction.y + center.y, -0.1 * unit_direction.z + center.z};
                            factors.push_back(f);
                            corner_num++;
                        }
                    }
                }
                for (int i = 0; i < nSurfStack; i++) {  // :644-688
                    const P4 pointOri = surfStack[i];
                    P4 pointSel;
                    pointAssociateToMap(pointOri, pointSel);
                    const float qv[3] = {pointSel.x, pointSel.y, pointSel.z};
                    if (cfg.knn_mode == 0)
                        kdSurf.knn(qv, 5, ind, sq);
                    else
                        kdSurf.knn_brute(qv, 5, ind, sq);
                    if (sq[4] < 1.0) {
                        double A[15], B[5] = {-1, -1, -1, -1, -1};
                        for (int j = 0; j < 5; j++) {
                            A[3 * j + 0] = surfFromMap[ind[j]].x;
                            A[3 * j + 1] = surfFromMap[ind[j]].y;
                            A[3 * j + 2] = surfFromMap[ind[j]].z;
                        }
                        double nv[3];
                        colpiv_qr_solve_5x3(A, B, nv);
                        const double nn = std::sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
                        const double negative_OA_dot_norm = 1 / nn;
                        nv[0] /= nn, nv[1] /= nn, nv[2] /= nn;
                        bool planeValid = true;
                        for (int j = 0; j < 5; j++) {
                            if (std::fabs(nv[0] * surfFromMap[ind[j]].x + nv[1] * surfFromMap[ind[j]].y + nv[2] * surfFromMap[ind[j]].z +
                                          negative_OA_dot_norm) > 0.2) {
                                planeValid = false;
                                break;
                            }
                        }
                        if (planeValid) {
                            Factor f;
                            f.kind = 2;
                            f.cp = {pointOri.x, pointOri.y, pointOri.z};
                            f.a = {nv[0], nv[1], nv[2]};
                            f.b = {negative_OA_dot_norm, 0, 0};
                            factors.push_back(f);
                            surf_num++;
                        }
                    }
                }
                auto tc = clk::now();
                t_assoc += ms(tb, tc);
                S.n_edge[iterCount] = corner_num, S.n_plane[iterCount] = surf_num;
                LMSummary sum;
                ceres_solve(factors, parameters, &sum);  // :713-721
                S.lm_iters[iterCount] = sum.iterations;
                S.lm_success[iterCount] = sum.successful_steps;
                S.cost_init[iterCount] = sum.initial_cost;
                S.cost_final[iterCount] = sum.final_cost;
                t_solve += ms(tc, clk::now());
            }
        }
        auto t3 = clk::now();
        {  // transformUpdate :149-153
            q_wmap_wodom = qmul(q_w_curr(), qinv(q_wodom_curr));
            t_wmap_wodom = t_w_curr() - rotate(q_wmap_wodom, t_wodom_curr);
        }
        auto insert = [&](const std::vector<P4>& stack, int cnt, std::vector<std::vector<P4>*>& arr) {  // :738-784
            for (int i = 0; i < cnt; i++) {
                P4 pointSel;
                pointAssociateToMap(stack[i], pointSel);
                int cubeI = int((pointSel.x + 25.0) / 50.0) + cenW;
                int cubeJ = int((pointSel.y + 25.0) / 50.0) + cenH;
                int cubeK = int((pointSel.z + 25.0) / 50.0) + cenD;
                if (pointSel.x + 25.0 < 0) cubeI--;
                if (pointSel.y + 25.0 < 0) cubeJ--;
                if (pointSel.z + 25.0 < 0) cubeK--;
                if (cubeI >= 0 && cubeI < W && cubeJ >= 0 && cubeJ < H && cubeK >= 0 && cubeK < D) arr[idx(cubeI, cubeJ, cubeK)]->push_back(pointSel);
            }
        };
        insert(cornerStack, nCornerStack, cornerArr);
        insert(surfStack, nSurfStack, surfArr);
        auto t4 = clk::now();
        std::vector<P4> tmp;
        for (int i = 0; i < validNum; i++) {  // :789-802
            const int ind = validInd[i];
            tmp.resize(cornerArr[ind]->size() + 1);
            int m = voxel_grid(cornerArr[ind]->data(), static_cast<int>(cornerArr[ind]->size()), cfg.line_res, cfg.voxel_order, tmp.data(), nullptr);
            cornerArr[ind]->assign(tmp.begin(), tmp.begin() + m);
            tmp.resize(surfArr[ind]->size() + 1);
            m = voxel_grid(surfArr[ind]->data(), static_cast<int>(surfArr[ind]->size()), cfg.plane_res, cfg.voxel_order, tmp.data(), nullptr);
            surfArr[ind]->assign(tmp.begin(), tmp.begin() + m);
        }
        auto t5 = clk::now();
        if (fullRes && registered)  // :845-849
            for (int i = 0; i < nFull; i++) pointAssociateToMap(fullRes[i], registered[i]);
        auto t6 = clk::now();
        for (int i = 0; i < 4; ++i) q_out[i] = parameters[i];
        for (int i = 0; i < 3; ++i) t_out[i] = parameters[4 + i];
        S.t_ms[0] = ms(t0, t1), S.t_ms[1] = ms(t1, t2), S.t_ms[2] = t_tree, S.t_ms[3] = t_assoc, S.t_ms[4] = t_solve;
        S.t_ms[5] = ms(t3, t4), S.t_ms[6] = ms(t4, t5) + ms(t5, t6), S.t_ms[7] = ms(t0, t6);
        if (st) *st = S;
        return 0;
    }

    // every cube of the grid, one class: what laserMapping.cpp:824-837 adds up for /laser_cloud_map
    int export_all(int which, P4* out, int cap) const {
        int m = 0;
        for (size_t i = 0; i < cornerArr.size(); i++) {
            const std::vector<P4>& v = which == 0 ? *cornerArr[i] : *surfArr[i];
            for (const P4& p : v) {
                if (m < cap) out[m] = p;
                ++m;
            }
        }
        return m;
    }
    int export_map(int which, P4* out, int cap) const {
        int m = 0;
        for (int i = 0; i < validNum; i++) {
            const std::vector<P4>& v = which == 0 ? *cornerArr[validInd[i]] : *surfArr[validInd[i]];
            for (const P4& p : v) {
                if (m < cap) out[m] = p;
                ++m;
            }
        }
        return m;
    }
};

}  // namespace orc

extern "C" {
void orc_eig3_sym(const double* A9, double* w3, double* V9) { orc::eig3_sym(A9, w3, V9); }
void orc_plane_fit_5x3(const double* A15, const double* b5, double* x3) { orc::colpiv_qr_solve_5x3(A15, b5, x3); }
int orc_ceres_solve(int n, const int* kind, const double* cp, const double* pa, const double* pb, double* x7, double* cost_trace, int* n_trace,
                    int* termination) {
    std::vector<orc::Factor> F(n);
    for (int i = 0; i < n; ++i) {
        F[i].kind = kind[i];
        F[i].cp = {cp[3 * i], cp[3 * i + 1], cp[3 * i + 2]};
        F[i].a = {pa[3 * i], pa[3 * i + 1], pa[3 * i + 2]};
        F[i].b = {pb[3 * i], pb[3 * i + 1], pb[3 * i + 2]};
    }
    orc::LMSummary S;
    orc::ceres_solve(F, x7, &S);
    int m = static_cast<int>(std::min<size_t>(S.cost_trace.size(), 6));
    for (int i = 0; i < m; ++i) cost_trace[i] = S.cost_trace[i];
    *n_trace = m;
    *termination = S.termination;
    return S.iterations;
}
void* orc_map_create(const OrcMapConfig* cfg) { return new orc::Mapper(*cfg); }
void orc_map_destroy(void* h) { delete static_cast<orc::Mapper*>(h); }
int orc_map_step(void* h, const float* corner_last, int n_corner, const float* surf_last, int n_surf, const float* full_res,
                 int n_full, const double* q_wodom, const double* t_wodom, double* q_w_curr, double* t_w_curr, float* registered,
                 OrcMapStats* stats) {
    return static_cast<orc::Mapper*>(h)->step(reinterpret_cast<const orc::P4*>(corner_last), n_corner,
                                             reinterpret_cast<const orc::P4*>(surf_last), n_surf,
                                             reinterpret_cast<const orc::P4*>(full_res), n_full, q_wodom, t_wodom, q_w_curr, t_w_curr,
                                             reinterpret_cast<orc::P4*>(registered), stats);
}
int orc_map_export_all(void* h, int which, float* out, int cap) {
    return static_cast<orc::Mapper*>(h)->export_all(which, reinterpret_cast<orc::P4*>(out), cap);
}
int orc_map_export(void* h, int which, float* out, int cap) {
    return static_cast<orc::Mapper*>(h)->export_map(which, reinterpret_cast<orc::P4*>(out), cap);
}
void orc_map_get_wmap_wodom(void* h, double* q, double* t) {
    auto* m = static_cast<orc::Mapper*>(h);
    q[0] = m->q_wmap_wodom.x, q[1] = m->q_wmap_wodom.y, q[2] = m->q_wmap_wodom.z, q[3] = m->q_wmap_wodom.w;
    t[0] = m->t_wmap_wodom.x, t[1] = m->t_wmap_wodom.y, t[2] = m->t_wmap_wodom.z;
}
}
