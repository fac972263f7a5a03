// oracle/_ref shim for <nanoflann.hpp> (TEST INFRASTRUCTURE ONLY, see mini_eigen.h).  The GN path never builds a kd-tree
// (ct_icp.cpp:778 asks for A2D | NORMAL, map.h:231 for ALL_BUT_KDTREE); the types only have to exist.  findNeighbors is a
// brute-force scan so that the class stays usable if something does call it.
#ifndef CTGN_ORACLE_NANOFLANN_SHIM_H
#define CTGN_ORACLE_NANOFLANN_SHIM_H
#include <cstddef>
#include <vector>
#include <limits>
namespace nanoflann {
    struct KDTreeSingleIndexAdaptorParams { explicit KDTreeSingleIndexAdaptorParams(std::size_t leaf = 10) : leaf_max_size(leaf) {} std::size_t leaf_max_size; };
    struct SearchParams { explicit SearchParams(int = 32, float = 0.f, bool = true) {} };
    template<class T, class DataSource, typename _DistanceType = T> struct L2_Simple_Adaptor { typedef T ElementType; typedef _DistanceType DistanceType; };
    template<typename Distance, class DatasetAdaptor, int DIM = -1, typename IndexType = std::size_t>
    class KDTreeSingleIndexAdaptor {
    public:
        KDTreeSingleIndexAdaptor(int, const DatasetAdaptor &d, const KDTreeSingleIndexAdaptorParams & = KDTreeSingleIndexAdaptorParams()) : dataset(d) {}
        void buildIndex() {}
        template<typename RESULTSET> bool findNeighbors(RESULTSET &result, const double *q, const SearchParams &) const {
            const std::size_t n = dataset.kdtree_get_point_count();
            for (std::size_t i = 0; i < n; ++i) {
                double d2 = 0;
                for (int k = 0; k < DIM; ++k) { double d = dataset.kdtree_get_pt(i, std::size_t(k)) - q[k]; d2 += d * d; }
                result.addPoint(d2, i);
            }
            return true;
        }
        const DatasetAdaptor &dataset;
    };
}
#endif
