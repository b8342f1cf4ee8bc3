// pt_bvh.h — software acceleration structure and ray search; stands in for the OptiX GAS
// (optixAccelBuild, SimplePathtracer.cpp:457-601) and for optixTrace (deviceProgram.cu:165,190).
//
// Closest hit = smallest t in (tmin,tmax) over all triangles, ties broken by the lowest global
// primitive index, so the answer does not depend on the tree or on traversal order.  The triangle
// test is a sign-consistent scalar-triple-product test (watertight across shared edges like the
// RT-core test it replaces, no backface culling, OPTIX_RAY_FLAG_NONE); every operation is a single
// rounded IEEE op, mirrored one for one by the CPU checker (wtri), so t is bit-identical.
// Box tests are conservative (far plane widened, boxes padded at build) — they may only ever
// admit extra triangles, never reject one the triangle test accepts.
#pragma once
#include "pt_device.h"

// 64-byte binary node: both children's boxes + refs.
//   a = (c0.lo.xyz, c0.hi.x)  b = (c0.hi.yz, c1.lo.xy)  c = (c1.lo.z, c1.hi.xyz)  d = (ref0, ref1, -, -)
// ref >= 0: internal node index.  ref < 0: leaf, ~ref = (first_triangle << 3) | (count-1).
// ref == PT_REF_EMPTY: no child (box is inverted, never hit).
struct Node2 {
    float4 a, b, c, d;
};
#define PT_REF_EMPTY 0x7fffffff
#define PT_LEAF_MAX 4

// 48-byte leaf triangle, in leaf order: t0 = (v0.xyz, v1.x) t1 = (v1.yz, v2.xy) t2 = (v2.z, prim bits, mesh bits, -).
// The hit record of a closest-hit ray holds the INDEX of the leaf triangle (not the primitive): the shade kernel reads the very
// 48 bytes the traversal just pulled through the caches — vertices, primitive (for texcoords) and mesh (material record, what the
// SBT record index gave the reference, deviceProgram.cu:481-489) — instead of a second triangle array in primitive order.
struct LeafTri {
    float4 t0, t1, t2;
};

struct BvhDev {
    const Node2* nodes;
    const LeafTri* tris;
    int32_t root; // ref of the root (internal index 0, or a leaf ref for tiny scenes)
    float hit_pad; // half the builder's box padding (tri_test_det)
};

struct RaySetup {
    v3 o, d, idir, dn;
};
PT_DEV RaySetup ray_setup(v3 o, v3 d) {
    RaySetup r;
    r.o = o;
    r.d = d;
    r.idir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    float inv_dd = 1.0f / dot3(d, d);
    r.dn = scl3(d, inv_dd);
    return r;
}

// returns true with t if the supporting ray (t > 0) hits the triangle; the caller applies (tmin,tmax) and then hit_in_box
PT_DEV bool tri_test_det(const RaySetup& r, v3 v0, v3 v1, v3 v2, float& t, float& det_out);
// Last criterion of a hit, applied by the callers to a candidate that passed the sign test and the t interval: the hit point must lie
// inside the triangle's bounding box widened by hp = half the padding the builder gives every box (2^-17 of the scene's largest
// |coordinate|).  In float arithmetic a needle triangle seen along its axis (or any triangle seen edge-on) turns the three edge functions
// into rounding noise and the sign test accepts rays that pass the triangle at many times its width — hits OUTSIDE the triangle's box,
// which a hierarchy finds or not depending on which other boxes it happens to enter (stadium scene: 5 of 3 M camera rays differed
// between the LBVH- and the PLOC-built tree).  With the criterion every accepted hit lies inside every structure's box of that
// triangle, so brute force and all trees agree again.  (o + t d as one fused multiply-add per axis: the checker calls fmaf.)
PT_DEV bool hit_in_box(const RaySetup& r, v3 v0, v3 v1, v3 v2, float hp, float t) {
    const float px = __builtin_fmaf(r.d.x, t, r.o.x), py = __builtin_fmaf(r.d.y, t, r.o.y), pz = __builtin_fmaf(r.d.z, t, r.o.z);
    return !(px < fminf(fminf(v0.x, v1.x), v2.x) - hp || px > fmaxf(fmaxf(v0.x, v1.x), v2.x) + hp ||
             py < fminf(fminf(v0.y, v1.y), v2.y) - hp || py > fmaxf(fmaxf(v0.y, v1.y), v2.y) + hp ||
             pz < fminf(fminf(v0.z, v1.z), v2.z) - hp || pz > fmaxf(fmaxf(v0.z, v1.z), v2.z) + hp);
}
// det > 0: the ray meets the triangle's front (counter-clockwise) side — what OPTIX_RAY_FLAG_CULL_BACK_FACING keeps
PT_DEV bool tri_test_det(const RaySetup& r, v3 v0, v3 v1, v3 v2, float& t, float& det_out) {
    const v3 A = sub3(v0, r.o), B = sub3(v1, r.o), C = sub3(v2, r.o);
    const v3 CxB = cross3(C, B), AxC = cross3(A, C), BxA = cross3(B, A);
    const float U = dot3(r.d, CxB), V = dot3(r.d, AxC), W = dot3(r.d, BxA);
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    const float det = U + V + W;
    if (det == 0.0f) return false;
    const float Ad = dot3(r.dn, A), Bd = dot3(r.dn, B), Cd = dot3(r.dn, C);
    const float T = U * Ad + V * Bd + W * Cd;
    if (T == 0.0f || ((T < 0.0f) != (det < 0.0f))) return false;
    t = T / det;
    det_out = det;
    return true;
}

PT_DEV bool box_test(float lox, float loy, float loz, float hix, float hiy, float hiz, const RaySetup& r, float tmin,
                     float tmax, float& tnear) {
    float ax = (lox - r.o.x) * r.idir.x, bx = (hix - r.o.x) * r.idir.x;
    float ay = (loy - r.o.y) * r.idir.y, by = (hiy - r.o.y) * r.idir.y;
    float az = (loz - r.o.z) * r.idir.z, bz = (hiz - r.o.z) * r.idir.z;
    float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
    float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) * 1.0000004f;
    tf = fminf(tf, tmax);
    tnear = tn;
    return tn <= tf;
}

#define PT_STACK_DEPTH 64

// Per-lane traversal with the stack in LDS (stack[level * stride + lane_slot]).
// ANY: stop at the first hit in (tmin,tmax) → prim = 1/0.  else closest hit → (t, leaf-triangle index of the hit or -1).
template <bool ANY>
PT_DEV void bvh2_traverse(const BvhDev& bvh, v3 o, v3 d, float tmin, float tmax, uint32_t* stack, uint32_t stride,
                          float& t_out, int32_t& prim_out) {
    const RaySetup r = ray_setup(o, d);
    float best = tmax;
    int32_t bprim = -1, bleaf = -1;
    int sp = 0;
    int32_t node = bvh.root;
    for (;;) {
        if (node >= 0) {
            if (node == PT_REF_EMPTY) goto pop;
            {
                const Node2* n = &bvh.nodes[node];
                const float4 na = n->a, nb = n->b, nc = n->c, nd = n->d;
                float t0, t1;
                const bool h0 = box_test(na.x, na.y, na.z, na.w, nb.x, nb.y, r, tmin, best, t0);
                const bool h1 = box_test(nb.z, nb.w, nc.x, nc.y, nc.z, nc.w, r, tmin, best, t1);
                const int32_t c0 = __float_as_int(nd.x), c1 = __float_as_int(nd.y);
                if (h0 && h1) {
                    const bool swap = t1 < t0;
                    const int32_t nearc = swap ? c1 : c0, farc = swap ? c0 : c1;
                    if (sp < PT_STACK_DEPTH) stack[(sp++) * stride] = (uint32_t)farc;
                    node = nearc;
                    continue;
                } else if (h0) {
                    node = c0;
                    continue;
                } else if (h1) {
                    node = c1;
                    continue;
                }
            }
        } else {
            const uint32_t code = ~(uint32_t)node;
            const uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
            for (uint32_t k = 0; k < cnt; ++k) {
                const LeafTri* tp = &bvh.tris[first + k];
                const float4 a = tp->t0, b = tp->t1, c = tp->t2;
                float t, det;
                const v3 v0 = mk3(a.x, a.y, a.z), v1 = mk3(a.w, b.x, b.y), v2 = mk3(b.z, b.w, c.x);
                if (tri_test_det(r, v0, v1, v2, t, det)) {
                    const int32_t prim = __float_as_int(c.y);
                    if (ANY) {
                        if (t > tmin && t < tmax && hit_in_box(r, v0, v1, v2, bvh.hit_pad, t)) {
                            prim_out = 1;
                            t_out = t;
                            return;
                        }
                    } else if (t > tmin && (t < best || (t == best && bprim >= 0 && prim < bprim)) && hit_in_box(r, v0, v1, v2, bvh.hit_pad, t)) {
                        best = t;
                        bprim = prim;
                        bleaf = (int32_t)(first + k);
                    }
                }
            }
        }
    pop:
        if (sp == 0) break;
        node = (int32_t)stack[(--sp) * stride];
    }
    t_out = best;
    prim_out = ANY ? 0 : bleaf;
}
