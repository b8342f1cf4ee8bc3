// pt_bvh.h — what defines a hit: leaf triangles, the ray/triangle test, the box criterion, the slab test of uncompressed boxes.  The
// structure that stands in for the OptiX GAS (optixAccelBuild, SimplePathtracer.cpp:457-601) and the traversal that stands in for
// optixTrace (deviceProgram.cu:165,190) are pt_bvh8.h; rounds 1-2 also kept a binary tree with two traversal kernels as an A/B path.
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

// 48-byte leaf triangle, in leaf order: t0 = (v0.xyz, v1.x) t1 = (v1.yz, v2.xy) t2 = (v2.z, prim bits, mesh bits, -).
// The hit record of a closest-hit ray holds the INDEX of the leaf triangle (not the primitive): the shade kernel reads the very
// 48 bytes the traversal just pulled through the caches — vertices, primitive (for texcoords) and mesh (material record, what the
// SBT record index gave the reference, deviceProgram.cu:481-489) — instead of a second triangle array in primitive order.
struct LeafTri {
    float4 t0, t1, t2;
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
// The tolerance grows with the distance travelled: t carries a relative error of a few 2^-24, so the computed hit point is off by up to
// ~2^-22 of |t d| along the ray — for a camera tens of scene sizes away more than hp, and a flat, axis-aligned triangle (a floor quad: a box
// of zero thickness) then lost genuine hits at random.  hpe = hp + 2^-21 * t * (|dx| + |dy| + |dz|): for origins within ~16 scene sizes the
// band stays inside the builder's padding (2 hp) and every structure agrees with brute force, as before; beyond, genuine hits are kept and
// only rounding-noise hits may depend on the structure.  Three multiplications and additions, each rounded, the checker's in the same order.
PT_DEV bool hit_in_box(const RaySetup& r, v3 v0, v3 v1, v3 v2, float hp, float t) {
    const float px = __builtin_fmaf(r.d.x, t, r.o.x), py = __builtin_fmaf(r.d.y, t, r.o.y), pz = __builtin_fmaf(r.d.z, t, r.o.z);
    const float l1 = (fabsf(r.d.x) + fabsf(r.d.y)) + fabsf(r.d.z);
    const float hpe = hp + (t * l1) * 4.76837158203125e-07f;
    return !(px < fminf(fminf(v0.x, v1.x), v2.x) - hpe || px > fmaxf(fmaxf(v0.x, v1.x), v2.x) + hpe ||
             py < fminf(fminf(v0.y, v1.y), v2.y) - hpe || py > fmaxf(fmaxf(v0.y, v1.y), v2.y) + hpe ||
             pz < fminf(fminf(v0.z, v1.z), v2.z) - hpe || pz > fmaxf(fmaxf(v0.z, v1.z), v2.z) + hpe);
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
