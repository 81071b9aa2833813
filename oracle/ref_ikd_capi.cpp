// TEST INFRASTRUCTURE ONLY (oracle/).  Thin C wrapper around the REFERENCE's own ikd-Tree
// (include/ikd-Tree/ikd_Tree.cpp, compiled unmodified from /root/reference by oracle/Makefile
// into oracle/_ref/libikd_ref.so).  Nothing of the reference's source is copied here: this file
// only calls its public API (ikd_Tree.h:225-249) so that tests and bench.py's cpu_baseline can
// drive the real thing through ctypes.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include <ikd_Tree.h>
#include <cstring>
#include <chrono>
#include <thread>

using PointType = pcl::PointXYZINormal;          // include/common_lib.h:161
using Tree = KD_TREE<PointType>;
using PointVector = Tree::PointVector;

static PointVector to_vec(const float* xyz, int n) {
    PointVector v(n);
    for (int i = 0; i < n; ++i) {
        v[i].x = xyz[3 * i + 0];
        v[i].y = xyz[3 * i + 1];
        v[i].z = xyz[3 * i + 2];
    }
    return v;
}

extern "C" {

// KD_TREE embeds a 1M-entry operation queue (ikd_Tree.h:18,45,182) => ~100 MB: heap only.
void* ref_ikd_create(float delete_param, float balance_param, float box_length) {
    return new Tree(delete_param, balance_param, box_length);
}
void ref_ikd_destroy(void* h) { delete static_cast<Tree*>(h); }
void ref_ikd_set_downsample(void* h, float box_length) {
    static_cast<Tree*>(h)->set_downsample_param(box_length);   // ikd_Tree.cpp:40
}
void ref_ikd_build(void* h, const float* xyz, int n) {          // ikd_Tree.cpp:353
    static_cast<Tree*>(h)->Build(to_vec(xyz, n));
}
void ref_ikd_reconstruct(void* h, const float* xyz, int n) {    // ikd_Tree.cpp:1394 (caller laserMapping.cpp:656)
    static_cast<Tree*>(h)->reconstruct(to_vec(xyz, n));
}
int ref_ikd_size(void* h) { return static_cast<Tree*>(h)->size(); }          // ikd_Tree.cpp:79
int ref_ikd_validnum(void* h) { return static_cast<Tree*>(h)->validnum(); }  // ikd_Tree.cpp:129

// Nearest_Search (ikd_Tree.cpp:367) for nq queries; out_xyz [nq*k*3], out_d [nq*k], out_cnt [nq].
// Unfilled slots are left as NaN / +inf.  Returns elapsed seconds of the search loop only.
double ref_ikd_nearest(void* h, const float* q_xyz, int nq, int k, float* out_xyz, float* out_d,
                       int* out_cnt) {
    Tree* t = static_cast<Tree*>(h);
    PointVector near;
    std::vector<float> d;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < nq; ++i) {
        PointType q;
        q.x = q_xyz[3 * i]; q.y = q_xyz[3 * i + 1]; q.z = q_xyz[3 * i + 2];
        t->Nearest_Search(q, k, near, d);
        int c = (int)near.size();
        out_cnt[i] = c;
        for (int j = 0; j < c; ++j) {
            out_xyz[(i * k + j) * 3 + 0] = near[j].x;
            out_xyz[(i * k + j) * 3 + 1] = near[j].y;
            out_xyz[(i * k + j) * 3 + 2] = near[j].z;
            out_d[i * k + j] = d[j];
        }
    }
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

int ref_ikd_add_points(void* h, const float* xyz, int n, int downsample_on) {  // ikd_Tree.cpp:414
    PointVector v = to_vec(xyz, n);
    return static_cast<Tree*>(h)->Add_Points(v, downsample_on != 0);
}
int ref_ikd_delete_boxes(void* h, const float* minmax6, int nb) {              // ikd_Tree.cpp:536
    std::vector<BoxPointType> boxes(nb);
    for (int i = 0; i < nb; ++i) {
        for (int a = 0; a < 3; ++a) {
            boxes[i].vertex_min[a] = minmax6[6 * i + a];
            boxes[i].vertex_max[a] = minmax6[6 * i + 3 + a];
        }
    }
    return static_cast<Tree*>(h)->Delete_Point_Boxes(boxes);
}
// flatten (ikd_Tree.cpp:1326) of the live set into out_xyz (cap points); returns the live count.
int ref_ikd_flatten(void* h, float* out_xyz, int cap) {
    Tree* t = static_cast<Tree*>(h);
    PointVector st;
    t->flatten(t->Root_Node, st, NOT_RECORD);
    int n = (int)st.size();
    for (int i = 0; i < n && i < cap; ++i) {
        out_xyz[3 * i] = st[i].x; out_xyz[3 * i + 1] = st[i].y; out_xyz[3 * i + 2] = st[i].z;
    }
    return n;
}
// acquire_removed_points (ikd_Tree.cpp:559-571): box-deleted points recorded by the rebuilds so far.
int ref_ikd_acquire_removed(void* h, float* out_xyz, int cap) {
    PointVector st;
    static_cast<Tree*>(h)->acquire_removed_points(st);
    int n = (int)st.size();
    for (int i = 0; i < n && i < cap; ++i) {
        out_xyz[3 * i] = st[i].x; out_xyz[3 * i + 1] = st[i].y; out_xyz[3 * i + 2] = st[i].z;
    }
    return n;
}
// The rebuild pthread (ikd_Tree.cpp:201) may still be re-balancing a sub-tree after a mutation;
// give it time to go idle so timing and visibility are stable.
void ref_ikd_quiesce(int ms) { std::this_thread::sleep_for(std::chrono::milliseconds(ms)); }

}  // extern "C"
