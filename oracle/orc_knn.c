/*
 * oracle/orc_knn.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
 *
 * Exact 5-nearest-neighbour search with the arithmetic of the reference's ikd-Tree, by brute force:
 *   distance           /root/reference/include/ikd-Tree/ikd_Tree.cpp:1291-1295  (float, (dx*dx + dy*dy) + dz*dz)
 *   result convention  /root/reference/include/ikd-Tree/ikd_Tree.cpp:350-380    (ascending distance)
 *   validity           /root/reference/src/laserMapping.cpp:1549,1567           (5 found && sqdist[4] <= 5)
 * The tree's traversal order decides which of several EXACTLY equidistant points survives; that
 * order is not reproducible without the tree, so ties are broken here by the lower map index
 * (the device search uses the same rule). PINNED to the reference's own ikd-Tree (oracle/ref_ikdtree, tests/test_ref_ikdtree_cpu.py).
 */
#include "fastlivo_oracle.h"

#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_knn5(const float *map_xyz, int k, const float *query_xyz, int n, float *nbr_xyz /* n x 5 x 3 */, float *sqdist /* n x 5 */,
             uint8_t *valid /* n */, int32_t *nbr_idx /* n x 5, nullable */, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int i = 0; i < n; i++) {
        float bd[5];
        int bi[5];
        for (int j = 0; j < 5; j++) { bd[j] = INFINITY; bi[j] = -1; }
        const float qx = query_xyz[i * 3], qy = query_xyz[i * 3 + 1], qz = query_xyz[i * 3 + 2];
        for (int m = 0; m < k; m++) {
            const float dx = qx - map_xyz[m * 3], dy = qy - map_xyz[m * 3 + 1], dz = qz - map_xyz[m * 3 + 2];
            const float d = dx * dx + dy * dy + dz * dz;
            if (!(d < bd[4] || (d == bd[4] && m < bi[4]))) continue;
            int pos = 4;
            while (pos > 0 && (d < bd[pos - 1] || (d == bd[pos - 1] && m < bi[pos - 1]))) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; pos--; }
            bd[pos] = d; bi[pos] = m;
        }
        int found = 0;
        for (int j = 0; j < 5; j++) {
            if (bi[j] >= 0) {
                found++;
                nbr_xyz[(i * 5 + j) * 3] = map_xyz[bi[j] * 3];
                nbr_xyz[(i * 5 + j) * 3 + 1] = map_xyz[bi[j] * 3 + 1];
                nbr_xyz[(i * 5 + j) * 3 + 2] = map_xyz[bi[j] * 3 + 2];
            } else {
                nbr_xyz[(i * 5 + j) * 3] = nbr_xyz[(i * 5 + j) * 3 + 1] = nbr_xyz[(i * 5 + j) * 3 + 2] = 0.f;
            }
            sqdist[i * 5 + j] = bd[j];
            if (nbr_idx) nbr_idx[i * 5 + j] = bi[j];
        }
        valid[i] = (uint8_t)(found == 5 && !(bd[4] > 5.0f));
    }
    return 0;
}
