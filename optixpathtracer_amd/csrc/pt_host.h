// pt_host.h — host-side structures shared by the translation units of libptamd.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/pt_amd.h"

struct Node8;
struct LeafTri;
struct Grid8 { // origin grid and base exponent of a tree of one-line nodes (pt_bvh8.h PT8_NODE64; zero for 80-byte nodes)
    float glo[3], gstep[3];
    uint32_t ebase; // biased float exponent of the tree's smallest grid step
};

// the traversal structure on the device: the 8-wide compressed tree (pt_bvh8.h) and what the kernels need to know about it
struct PtBvh {
    float bounds[6] = {0, 0, 0, 0, 0, 0};
    float pad = 0.f; // padding of every box: 2^-16 of the scene's largest |coordinate| (half of it confines hits: pt_bvh.h hit_in_box)
    const Node8* nodes8 = nullptr;
    const LeafTri* tris8 = nullptr;
    uint32_t num_nodes8 = 0, num_tris8 = 0;
    int levels8 = 0; // levels of the wide tree = upper bound of its traversal stack depth (one pushed group per level)
    Grid8 grid{}; // PT8_NODE64: what the one-line nodes' origins and steps are measured in (pt_bvh8.h)
    float calib_cost = 0.f; // node steps + 0.6 x triangle tests per calibration ray through the chosen tree (0: no calibration ran — small scenes, forced builder)
    int challengers_skipped = 0; // candidate hierarchies that could not be built (out of memory, a failed check): the standing tree stayed uncompared
    int builder = 0; // hierarchy under the wide tree: 0 LBVH (Morton order, Karras 2012), 1 PLOC (Meister & Bittner 2018) — chosen by calibration rays unless PT_BVH_BUILDER=lbvh|ploc
};

// d_tri_mesh: mesh (= material record) of every triangle, stored with the leaf triangles (may be null: 0)
hipError_t pt_bvh_build(const float* d_verts, const uint32_t* d_idx, const uint32_t* d_tri_mesh, uint32_t ntri, hipStream_t stream, PtBvh* out);
void pt_bvh_free(PtBvh* b);
void pt_bvh_warm(hipStream_t stream); // first use in a process: loads the code object of pt_bvh_build.hip (see there)
