// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Thin C wrapper around the reference's OWN vendored nanoflann
// (/root/reference/include/scancontext/nanoflann.hpp + KDTreeVectorOfVectorsAdaptor.h), compiled from
// where those headers lie (never copied) into oracle/_ref/libref_nanoflann.so by oracle/Makefile.
// It instantiates exactly the types SCManager uses (Scancontext.h:41-42: KeyMat = vector<vector<float>>,
// InvKeyTree = KDTreeVectorOfVectorsAdaptor<KeyMat,float>, max leaf 10, Scancontext.cpp:361) and runs the
// same query (Scancontext.cpp:372-378).  Used to pin the oracle's brute-force ring-key KNN (row D5).
#include <memory>
#include <cstddef>
#include <string>
#include <iostream>
#include "scancontext/nanoflann.hpp"
#include "scancontext/KDTreeVectorOfVectorsAdaptor.h"

using KeyMat = std::vector<std::vector<float>>;
using InvKeyTree = KDTreeVectorOfVectorsAdaptor<KeyMat, float>;

extern "C" int ref_ringkey_knn(const float* keys, int n, int dim, const float* queries, int nq, int k, int* out_idx,
                               float* out_d) {
    KeyMat mat(n, std::vector<float>(dim));
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < dim; ++d) mat[i][d] = keys[(size_t)i * dim + d];
    std::unique_ptr<InvKeyTree> tree = std::make_unique<InvKeyTree>(dim, mat, 10);
    for (int q = 0; q < nq; ++q) {
        std::vector<size_t> candidate_indexes(k);
        std::vector<float> out_dists_sqr(k);
        nanoflann::KNNResultSet<float> knnsearch_result(k);
        knnsearch_result.init(&candidate_indexes[0], &out_dists_sqr[0]);
        tree->index->findNeighbors(knnsearch_result, queries + (size_t)q * dim, nanoflann::SearchParams(10));
        for (int j = 0; j < k; ++j) {
            out_idx[(size_t)q * k + j] = static_cast<int>(candidate_indexes[j]);
            out_d[(size_t)q * k + j] = out_dists_sqr[j];
        }
    }
    return 0;
}
