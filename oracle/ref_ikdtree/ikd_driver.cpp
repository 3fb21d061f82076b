// oracle/ref_ikdtree/ikd_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A plain C interface over the REFERENCE's own ikd-Tree, compiled from /root/reference/include/ikd-Tree/ikd_Tree.cpp as it
// lies there (see Makefile).  It exposes exactly the calls FAST-LIVO makes on the tree, so that tests can hold the device
// k-NN / device map (and the brute-force restatements oracle/orc_knn.c, oracle/orc_map.c) to the reference itself:
//   ikdref_build            KD_TREE::Build                 ikd_Tree.cpp:337-348   (laserMapping.cpp:1410-1417, first frame)
//   ikdref_nearest          KD_TREE::Nearest_Search        ikd_Tree.cpp:350-380   (laserMapping.cpp:1543 / :1002)
//   ikdref_add_points       KD_TREE::Add_Points            ikd_Tree.cpp:382-457   (map_incremental, laserMapping.cpp:692-706)
//   ikdref_delete_boxes     KD_TREE::Delete_Point_Boxes    ikd_Tree.cpp:501-520   (lasermap_fov_segment, laserMapping.cpp:363-417)
//   ikdref_flatten          KD_TREE::flatten               ikd_Tree.cpp:1247-1272 (the live points, traversal order)
// The tree's rebuild thread runs as in the reference.  ikdref_wait_rebuild lets a test choose between "whatever the thread
// is doing" and a quiescent tree; the SET of points the tree holds must not depend on it.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include <unistd.h>
#include <pthread.h>

#define private public          // the driver peeks at Rebuild_Ptr / rebuild_ptr_mutex_lock to wait for the rebuild thread
#include "ikd_Tree.h"
#undef private

namespace {
struct Ref {
    KD_TREE *tree;
    void *mem;
};
PointType mk(const float *p)
{
    PointType q;
    q.x = p[0]; q.y = p[1]; q.z = p[2];
    return q;
}
}  // namespace

extern "C" {

// Rebuild_Ptr has no initialiser in the reference (ikd_Tree.h:125) and is read before it is ever written
// (ikd_Tree.cpp:75); the object therefore lives in zeroed storage, which is what a file-scope `KD_TREE ikdtree;`
// (laserMapping.cpp) gets from static initialisation.
void *ikdref_create(float delete_param, float balance_param, float box_length)
{
    Ref *r = new Ref;
    r->mem = std::calloc(1, sizeof(KD_TREE));
    r->tree = new (r->mem) KD_TREE(delete_param, balance_param, box_length);
    return r;
}
int ikdref_wait_rebuild(void *h, int timeout_ms);
// KD_TREE::multi_thread_ptr (ikd_Tree.cpp:182-185) is declared void* and has no return statement: letting the rebuild
// thread RETURN is undefined behaviour (gcc -O3 emits no `ret`, the thread runs into the next function and the process
// dies).  The reference never destroys its tree before exit; a test does.  So the thread is not allowed to return: once it
// is idle (Rebuild_Ptr == nullptr: it only cycles lock/unlock/usleep(100), and usleep is its one cancellation point, reached
// with no mutex held) it is cancelled and joined here, and stop_thread() then finds nothing left to join.
void ikdref_destroy(void *h)
{
    Ref *r = (Ref *)h;
    ikdref_wait_rebuild(h, 60000);
    pthread_cancel(r->tree->rebuild_thread);
    pthread_join(r->tree->rebuild_thread, NULL);
    r->tree->rebuild_thread = 0;
    r->tree->~KD_TREE();
    std::free(r->mem);
    delete r;
}
void ikdref_set_downsample(void *h, float box_length) { ((Ref *)h)->tree->set_downsample_param(box_length); }
int ikdref_size(void *h) { return ((Ref *)h)->tree->size(); }
int ikdref_validnum(void *h) { return ((Ref *)h)->tree->validnum(); }

// returns 1 when the tree is quiescent, 0 on timeout
int ikdref_wait_rebuild(void *h, int timeout_ms)
{
    KD_TREE *t = ((Ref *)h)->tree;
    for (int waited = 0; waited <= timeout_ms * 10; waited++) {
        pthread_mutex_lock(&t->rebuild_ptr_mutex_lock);
        const bool idle = (t->Rebuild_Ptr == nullptr);
        pthread_mutex_unlock(&t->rebuild_ptr_mutex_lock);
        if (idle) return 1;
        usleep(100);
    }
    return 0;
}

void ikdref_build(void *h, const float *xyz, int n)
{
    PointVector v((size_t)n);
    for (int i = 0; i < n; i++) v[i] = mk(xyz + 3 * i);
    ((Ref *)h)->tree->Build(v);
}

// k nearest per query, ascending; found[i] = number returned (< k when the tree holds fewer points).
// Unfilled slots: xyz 0, sqdist +inf.
void ikdref_nearest(void *h, const float *query_xyz, int n, int k, float *out_xyz /* n*k*3 */, float *out_sq /* n*k */, int32_t *found /* n */)
{
    KD_TREE *t = ((Ref *)h)->tree;
    PointVector near;
    std::vector<float> sq;
    for (int i = 0; i < n; i++) {
        t->Nearest_Search(mk(query_xyz + 3 * i), k, near, sq);
        const int f = (int)near.size();
        found[i] = f;
        for (int j = 0; j < k; j++) {
            float *o = out_xyz + ((size_t)i * k + j) * 3;
            if (j < f) { o[0] = near[j].x; o[1] = near[j].y; o[2] = near[j].z; out_sq[(size_t)i * k + j] = sq[j]; }
            else { o[0] = o[1] = o[2] = 0.f; out_sq[(size_t)i * k + j] = INFINITY; }
        }
    }
}

// The same search as the reference runs it: KD_TREE::Nearest_Search inside `#pragma omp parallel for` over the scan points
// (laserMapping.cpp:1516-1519 / :996, MP_PROC_NUM threads, CMakeLists.txt:23-26: 4 on a machine with more than 5 cores).
void ikdref_nearest_mt(void *h, const float *query_xyz, int n, int k, float *out_xyz, float *out_sq, int32_t *found, int nthreads)
{
    KD_TREE *t = ((Ref *)h)->tree;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int i = 0; i < n; i++) {
        PointVector near;
        std::vector<float> sq;
        t->Nearest_Search(mk(query_xyz + 3 * i), k, near, sq);
        const int f = (int)near.size();
        found[i] = f;
        for (int j = 0; j < k; j++) {
            float *o = out_xyz + ((size_t)i * k + j) * 3;
            if (j < f) { o[0] = near[j].x; o[1] = near[j].y; o[2] = near[j].z; out_sq[(size_t)i * k + j] = sq[j]; }
            else { o[0] = o[1] = o[2] = 0.f; out_sq[(size_t)i * k + j] = INFINITY; }
        }
    }
}

int ikdref_add_points(void *h, const float *xyz, int n, int downsample_on)
{
    PointVector v((size_t)n);
    for (int i = 0; i < n; i++) v[i] = mk(xyz + 3 * i);
    return ((Ref *)h)->tree->Add_Points(v, downsample_on != 0);
}

int ikdref_delete_boxes(void *h, const float *boxes /* nb x 6: min xyz, max xyz */, int nb)
{
    std::vector<BoxPointType> b((size_t)nb);
    for (int i = 0; i < nb; i++)
        for (int k = 0; k < 3; k++) { b[i].vertex_min[k] = boxes[6 * i + k]; b[i].vertex_max[k] = boxes[6 * i + 3 + k]; }
    return ((Ref *)h)->tree->Delete_Point_Boxes(b);
}

// live points in the tree's traversal order; returns their number (writes at most cap)
// Waits for the rebuild thread first: while it swaps a rebuilt subtree in, a traversal from Root_Node from another thread can come
// back short (seen once in ~40 runs of the CPU suite as an EMPTY list right behind Delete_Point_Boxes); the reference itself only
// flattens under its own locks.
int ikdref_flatten(void *h, float *out_xyz, int cap)
{
    KD_TREE *t = ((Ref *)h)->tree;
    ikdref_wait_rebuild(h, 60000);
    PointVector v;
    t->flatten(t->Root_Node, v, NOT_RECORD);
    const int n = (int)v.size();
    for (int i = 0; i < n && i < cap; i++) { out_xyz[3 * i] = v[i].x; out_xyz[3 * i + 1] = v[i].y; out_xyz[3 * i + 2] = v[i].z; }
    return n;
}

}  // extern "C"
