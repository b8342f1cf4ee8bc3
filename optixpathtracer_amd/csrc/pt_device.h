// pt_device.h — device-side building blocks of the wavefront path tracer (gfx950).
//
// Each function states which reference function it re-implements (paths relative to the
// reference repo, canonical variant HelloPathtracing_original/).  Arithmetic is written as
// explicit single IEEE operations in the reference's evaluation order and the TU is compiled with
// -ffp-contract=off, transcendental calls go through include/pt_detmath.h: the result of every
// function here is bit-identical to the CPU checker built with the same header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pt_amd.h"
#include "../../include/pt_detmath.h"

#define PT_DEV __device__ __forceinline__

struct v3 {
    float x, y, z;
};
PT_DEV v3 mk3(float x, float y, float z) { return v3{x, y, z}; }
PT_DEV v3 mk3(float s) { return v3{s, s, s}; }
PT_DEV v3 add3(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_DEV v3 sub3(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_DEV v3 mul3(v3 a, v3 b) { return v3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_DEV v3 scl3(v3 a, float s) { return v3{a.x * s, a.y * s, a.z * s}; }
PT_DEV v3 neg3(v3 a) { return v3{-a.x, -a.y, -a.z}; }
// sutil/vec_math.h:483-487  float3 / float = a * (1/s)
PT_DEV v3 div3s(v3 a, float s) {
    float inv = 1.0f / s;
    return scl3(a, inv);
}
// sutil/vec_math.h:535-544
PT_DEV float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_DEV v3 cross3(v3 a, v3 b) { return v3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// sutil/vec_math.h:553-557
PT_DEV v3 normalize3(v3 v) {
    float invLen = 1.0f / sqrtf(dot3(v, v));
    return scl3(v, invLen);
}
// sutil/vec_math.h:96-99, 500-503
PT_DEV float lerpf(float a, float b, float t) { return a + t * (b - a); }
PT_DEV v3 lerp3(v3 a, v3 b, float t) { return add3(a, scl3(sub3(b, a), t)); }
// sutil/vec_math.h:119-122
PT_DEV float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
// sutil/vec_math.h:567-570
PT_DEV v3 faceforward3(v3 n, v3 i, v3 nref) { return scl3(n, copysignf(1.0f, dot3(i, nref))); }

#define kPi (3.141592653589793f)
#define k2Pi (3.141592653589793f * 2.0f)
#define kInvPi (1.0f / kPi)
#define kInv2Pi (1.0f / k2Pi)

// ------------------------------------------------------------------ RNG
// cuda/random.h:34-49 tea<4>
PT_DEV uint32_t tea4(uint32_t val0, uint32_t val1) {
    uint32_t v0 = val0, v1 = val1, s0 = 0;
#pragma unroll
    for (int n = 0; n < 4; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
// cuda/random.h:53-59, 96-99
PT_DEV uint32_t lcg(uint32_t& prev) {
    prev = 1664525u * prev + 1013904223u;
    return prev & 0x00FFFFFFu;
}
PT_DEV float rnd(uint32_t& prev) { return (float)lcg(prev) / (float)0x01000000; }

// maths.h:170-225 class Random
struct Rng {
    uint32_t seed1, seed2;
    PT_DEV void init(uint32_t seed) {
        seed1 = 315645664u + seed;
        seed2 = seed1 ^ 0x13ab45feu;
    }
    PT_DEV uint32_t rand() {
        seed1 = (seed2 ^ ((seed1 << 5) | (seed1 >> 27))) ^ (seed1 * seed2);
        seed2 = seed1 ^ ((seed2 << 12) | (seed2 >> 20));
        return seed1;
    }
    PT_DEV float randf() {
        uint32_t value = rand();
        return clampf((float)value * (1.0f / 4294967296.0f), 0.f, 0.999999f);
    }
    // Randf(min,max) maths.h:211-215
    PT_DEV float randf(float mn, float mx) {
        float t = randf();
        return (1.0f - t) * mn + t * mx;
    }
};
// sample.h:252-258 (USE_RANDOM)
PT_DEV void sample2d(Rng& r, float& u1, float& u2) {
    u1 = r.randf(0.0f, 1.0f);
    u2 = r.randf(0.0f, 1.0f);
}

// maths.h:94-108
PT_DEV void basis_from_vector(v3 w, v3& u, v3& v) {
    if (fabsf(w.x) > fabsf(w.y)) {
        float invLen = 1.0f / sqrtf(w.x * w.x + w.z * w.z);
        u = mk3(-w.z * invLen, 0.0f, w.x * invLen);
    } else {
        float invLen = 1.0f / sqrtf(w.y * w.y + w.z * w.z);
        u = mk3(0.0f, w.z * invLen, -w.y * invLen);
    }
    v = cross3(w, u);
}
// maths.h:144-156
PT_DEV v3 safe_normalize(v3 a) {
    float m = dot3(a, a);
    if (m > 0.0f) return scl3(a, 1.0f / sqrtf(m));
    return mk3(0.0f);
}
// maths.h:241-252
PT_DEV v3 uniform_sample_hemisphere(Rng& r) {
    float z = r.randf(0.0f, 1.0f);
    float w = sqrtf(1.0f - z * z);
    float phi = k2Pi * r.randf(0.0f, 1.0f);
    float s, c;
    pt_sincosf(phi, &s, &c);
    return mk3(c * w, s * w, z);
}
// maths.h:254-275
PT_DEV v3 cosine_sample_hemisphere(float u1, float u2) {
    float r = sqrtf(u1);
    float theta = k2Pi * u2;
    float s, c;
    pt_sincosf(theta, &s, &c);
    float sx = r * c, sy = r * s;
    float z = sqrtf(fmaxf(0.0f, 1.0f - sx * sx - sy * sy));
    return mk3(sx, sy, z);
}

// Material.h:39-45
PT_DEV float material_ior(const pt_material& m) {
    if (m.eta == 0.0f) return 2.0f / (1.0f - sqrtf(0.08f * m.specular)) - 1.0f;
    return m.eta;
}

// ------------------------------------------------------------------ Disney BSDF (Disney.cuh)
PT_DEV float sqrf(float a) { return a * a; }
// :35-48
PT_DEV bool refract3(v3 wi, v3 n, float eta, v3& wt) {
    float cosThetaI = dot3(n, wi);
    float sin2ThetaI = fmaxf(0.0f, 1.0f - cosThetaI * cosThetaI);
    float sin2ThetaT = eta * eta * sin2ThetaI;
    if (sin2ThetaT >= 1) return false;
    float cosThetaT = sqrtf(1.0f - sin2ThetaT);
    wt = add3(scl3(neg3(wi), eta), scl3(n, eta * cosThetaI - cosThetaT));
    return true;
}
// :50-55
PT_DEV float schlick_fresnel(float u) {
    float m = clampf(1 - u, 0.0f, 1.0f);
    float m2 = m * m;
    return m2 * m2 * m;
}
// :57-63
PT_DEV float gtr1(float NDotH, float a) {
    if (a >= 1) return kInvPi;
    float a2 = a * a;
    float t = 1 + (a2 - 1) * NDotH * NDotH;
    return (a2 - 1) / (kPi * pt_logf(a2) * t);
}
// :65-70
PT_DEV float gtr2(float NDotH, float a) {
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
    return a2 / (kPi * t * t);
}
// :72-77
PT_DEV float smith_ggx(float NDotv, float alphaG) {
    float a = alphaG * alphaG;
    float b = NDotv * NDotv;
    return 1 / (NDotv + sqrtf(a + b - a * b));
}
// :80-97
PT_DEV float fresnel_dielectric(float VDotN, float etaI, float etaT) {
    float SinThetaT2 = sqrf(etaI / etaT) * (1.0f - VDotN * VDotN);
    if (SinThetaT2 > 1.0f) return 1.0f;
    float LDotN = sqrtf(1.0f - SinThetaT2);
    float eta = etaT / etaI;
    float r1 = (VDotN - eta * LDotN) / (VDotN + eta * LDotN);
    float r2 = (LDotN - eta * VDotN) / (LDotN + eta * VDotN);
    return 0.5f * (sqrf(r1) + sqrf(r2));
}

// BSDFPdf :151-192 (Lambert :127-133)
template <int MODE>
PT_DEV float bsdf_pdf(const pt_material& mat, float etaI, float etaO, v3 n, v3 V, v3 L) {
    if (MODE == PT_BSDF_LAMBERT) return (dot3(L, n) <= 0.0f) ? 0.0f : kInv2Pi;
    if (dot3(L, n) <= 0.0f) {
        float bsdfPdf = 0.0f;
        float brdfPdf = kInv2Pi * mat.subsurface * 0.5f;
        return lerpf(brdfPdf, bsdfPdf, mat.transmission);
    } else {
        float F = fresnel_dielectric(dot3(n, V), etaI, etaO);
        const float a = fmaxf(0.001f, mat.roughness);
        const v3 half = safe_normalize(add3(L, V));
        const float cosThetaHalf = fabsf(dot3(half, n));
        const float pdfHalf = gtr2(cosThetaHalf, a) * cosThetaHalf;
        float pdfSpec = 0.25f * pdfHalf / fmaxf(1.e-6f, dot3(L, half));
        float pdfDiff = fabsf(dot3(L, n)) * kInvPi * (1.0f - mat.subsurface);
        float bsdfPdf = pdfSpec * F;
        float brdfPdf = lerpf(pdfDiff, pdfSpec, 0.5f);
        return lerpf(brdfPdf, bsdfPdf, mat.transmission);
    }
}

// shared GGX half-vector reflection (:207-225 and :286-307)
PT_DEV v3 sample_ggx_reflect(const pt_material& mat, v3 U, v3 V, v3 N, v3 view, float r1, float r2) {
    const float a = fmaxf(0.001f, mat.roughness);
    const float phiHalf = r1 * k2Pi;
    const float cosThetaHalf = sqrtf((1.0f - r2) / (1.0f + (sqrf(a) - 1.0f) * r2));
    const float sinThetaHalf = sqrtf(fmaxf(0.0f, 1.0f - sqrf(cosThetaHalf)));
    float sinPhiHalf, cosPhiHalf;
    pt_sincosf(phiHalf, &sinPhiHalf, &cosPhiHalf);
    v3 half = add3(add3(scl3(U, sinThetaHalf * cosPhiHalf), scl3(V, sinThetaHalf * sinPhiHalf)), scl3(N, cosThetaHalf));
    if (dot3(half, view) <= 0.0f) half = scl3(half, -1.0f);
    return sub3(scl3(half, 2.0f * dot3(view, half)), view);
}

// BSDFSample :196-314 (Lambert :135-142)
template <int MODE>
PT_DEV void bsdf_sample(const pt_material& mat, float etaI, float etaO, v3 U, v3 V, v3 N, v3 view, v3& light, float& pdf,
                        Rng& rand) {
    if (MODE == PT_BSDF_LAMBERT) {
        v3 d = uniform_sample_hemisphere(rand);
        light = add3(add3(scl3(U, d.x), scl3(V, d.y)), scl3(N, d.z));
        pdf = kInv2Pi;
        return;
    }
    if (rand.randf() < mat.transmission) {
        float F = fresnel_dielectric(dot3(N, view), etaI, etaO);
        if (rand.randf() < F) {
            float r1, r2;
            sample2d(rand, r1, r2);
            light = sample_ggx_reflect(mat, U, V, N, view, r1, r2);
        } else {
            float eta = etaI / etaO;
            if (refract3(view, N, eta, light)) {
                pdf = (1.0f - F) * mat.transmission;
                return;
            } else {
                pdf = 0.0f;
                return;
            }
        }
    } else {
        float r1, r2;
        sample2d(rand, r1, r2);
        if (rand.randf() < 0.5f) {
            if (rand.randf() < mat.subsurface) {
                const v3 d = uniform_sample_hemisphere(rand);
                light = sub3(add3(scl3(U, d.x), scl3(V, d.y)), scl3(N, d.z));
            } else {
                const v3 d = cosine_sample_hemisphere(r1, r2);
                light = add3(add3(scl3(U, d.x), scl3(V, d.y)), scl3(N, d.z));
            }
        } else {
            light = sample_ggx_reflect(mat, U, V, N, view, r1, r2);
        }
    }
    pdf = bsdf_pdf<MODE>(mat, etaI, etaO, N, view, light);
}

// BSDFEval :317-426 (Lambert :144-147).  The FP64 sub-expressions are the reference's bare double
// literals (.3 .6 .1 at :328 and .08 at :331); 0.5 at :397 is exact in float (innocuous double rounding).
template <int MODE>
PT_DEV v3 bsdf_eval(const pt_material& mat, v3 albedo, float etaI, float etaO, v3 N, v3 V, v3 L) {
    if (MODE == PT_BSDF_LAMBERT) return scl3(albedo, kInvPi);
    float NDotL = dot3(N, L);
    float NDotV = dot3(N, V);
    v3 H = normalize3(add3(L, V));
    float NDotH = dot3(N, H);
    float LDotH = dot3(L, H);
    v3 Cdlin = albedo;
    float Cdlum = (float)(.3 * (double)Cdlin.x + .6 * (double)Cdlin.y + .1 * (double)Cdlin.z);
    v3 Ctint = Cdlum > 0.0f ? div3s(Cdlin, Cdlum) : mk3(1.0f);
    float spec08 = (float)((double)mat.specular * .08);
    v3 Cspec0 = lerp3(scl3(lerp3(mk3(1.0f), Ctint, mat.specularTint), spec08), Cdlin, mat.metallic);
    v3 bsdf = mk3(0.0f);
    v3 brdf = mk3(0.0f);
    if (mat.transmission > 0.0f) {
        if (NDotL <= 0) {
            float F = fresnel_dielectric(NDotV, etaI, etaO);
            bsdf = mk3(mat.transmission * (1.0f - F) / fabsf(NDotL) * (1.0f - mat.metallic));
        } else {
            float a = fmaxf(0.001f, mat.roughness);
            float Ds = gtr2(NDotH, a);
            float FH = fresnel_dielectric(LDotH, etaI, etaO);
            v3 Fs = lerp3(Cspec0, mk3(1.0f), FH);
            float Gs = smith_ggx(NDotV, a) * smith_ggx(NDotL, a);
            bsdf = scl3(scl3(Fs, Gs), Ds);
        }
    }
    if (mat.transmission < 1.0f) {
        if (NDotL <= 0) {
            if (mat.subsurface > 0.0f) {
                v3 s = mk3(sqrtf(mat.color[0]), sqrtf(mat.color[1]), sqrtf(mat.color[2]));
                float FL = schlick_fresnel(fabsf(NDotL)), FV = schlick_fresnel(NDotV);
                float Fd = (1.0f - 0.5f * FL) * (1.0f - 0.5f * FV);
                brdf = scl3(scl3(scl3(scl3(s, kInvPi), mat.subsurface), Fd), 1.0f - mat.metallic);
            }
        } else {
            float a = fmaxf(0.001f, mat.roughness);
            float Ds = gtr2(NDotH, a);
            float FH = schlick_fresnel(LDotH);
            v3 Fs = lerp3(Cspec0, mk3(1.f), FH);
            float Gs = smith_ggx(NDotV, a) * smith_ggx(NDotL, a);
            float FL = schlick_fresnel(NDotL), FV = schlick_fresnel(NDotV);
            float Fd90 = 0.5f + 2.0f * LDotH * LDotH * mat.roughness;
            float Fd = lerpf(1.0f, Fd90, FL) * lerpf(1.0f, Fd90, FV);
            float Dr = gtr1(NDotH, lerpf(.1f, .001f, mat.clearcoatGloss));
            float Fc = lerpf(.04f, 1.0f, FH);
            float Gr = smith_ggx(NDotL, .25f) * smith_ggx(NDotV, .25f);
            v3 diff = scl3(scl3(scl3(Cdlin, kInvPi * Fd), 1.0f - mat.metallic), 1.0f - mat.subsurface);
            v3 spec = scl3(scl3(Fs, Gs), Ds);
            float cc = mat.clearcoat * Gr * Fc * Dr;
            brdf = add3(add3(diff, spec), mk3(cc));
        }
    }
    return lerp3(brdf, bsdf, mat.transmission);
}

// Path state streams through the caches once per bounce (hundreds of MB per pass); NT = true marks an access non-temporal, so that it does
// not displace the tables every ray and hit look up at random (BVH, probe lines, triangle normals).  Measured (C3, six interleaved runs each):
// k_trace8 loads + stores and k_shade stores non-temporal −1.5 % frame time; k_shade's LOADS as well: +0.5…1 % (they re-read what the
// traversal launch just wrote); k_generate's stores, the ray queues, k_resolve's loads: no difference; k_shade's random table loads (probe lines, triangle normals): +5 % — they do hit the caches.
#ifndef PT_NT_TRACE_LD
#define PT_NT_TRACE_LD 1 // k_trace8: ray / pending-contribution loads
#endif
#ifndef PT_NT_TRACE_ST
#define PT_NT_TRACE_ST 1 // k_trace8: hit records, accumulators
#endif
#ifndef PT_NT_SHADE_LD
#define PT_NT_SHADE_LD 0 // k_shade: state loads
#endif
#ifndef PT_NT_SHADE_ST
#define PT_NT_SHADE_ST 1 // k_shade: state stores
#endif
typedef float pt_f4v __attribute__((ext_vector_type(4)));
typedef float pt_f2v __attribute__((ext_vector_type(2)));
typedef uint32_t pt_u4v __attribute__((ext_vector_type(4)));
template <bool NT> PT_DEV float4 st_ld(const float4* p) {
    if (!NT) return *p;
    const pt_f4v v = __builtin_nontemporal_load(reinterpret_cast<const pt_f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
template <bool NT> PT_DEV float2 st_ld(const float2* p) {
    if (!NT) return *p;
    const pt_f2v v = __builtin_nontemporal_load(reinterpret_cast<const pt_f2v*>(p));
    return make_float2(v.x, v.y);
}
template <bool NT> PT_DEV uint4 st_ld(const uint4* p) {
    if (!NT) return *p;
    const pt_u4v v = __builtin_nontemporal_load(reinterpret_cast<const pt_u4v*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <bool NT> PT_DEV void st_st(float4* p, float4 a) {
    if (!NT) { *p = a; return; }
    const pt_f4v v = {a.x, a.y, a.z, a.w};
    __builtin_nontemporal_store(v, reinterpret_cast<pt_f4v*>(p));
}
template <bool NT> PT_DEV void st_st(float2* p, float2 a) {
    if (!NT) { *p = a; return; }
    const pt_f2v v = {a.x, a.y};
    __builtin_nontemporal_store(v, reinterpret_cast<pt_f2v*>(p));
}
template <bool NT> PT_DEV void st_st(uint4* p, uint4 a) {
    if (!NT) { *p = a; return; }
    const pt_u4v v = {a.x, a.y, a.z, a.w};
    __builtin_nontemporal_store(v, reinterpret_cast<pt_u4v*>(p));
}

// ------------------------------------------------------------------ probe (Probe.cuh)
#define PT_CDF_BLOCK 64
// ProbeSample's per-column data, co-located: one 128-byte line (= one L2 line) holds six consecutive columns of a row — their conditional-CDF
// values and their texels (rgb, pdfX) — so that the end of the column search and the texel fetch touch ONE line instead of three
// (c8 window, CDF group, texel: the round 2/3 layout, profiles/r2_design_history.md).  Columns past the width: cdf = +inf, texel = 0.
#define PT_LINE_COLS 6
struct __attribute__((aligned(128))) ProbeLine {
    float cdf[PT_LINE_COLS];
    uint32_t pad_[2];
    float4 px[PT_LINE_COLS];
};
static_assert(sizeof(ProbeLine) == 128, "one probe line per 128-byte cache line");
struct DevProbe {
    int width, height;
    const float4* data;
    const float *pdfX, *cdfX, *pdfY, *cdfY;
    // row search accelerators built at setProbe: the last CDF value of every PT_CDF_BLOCK-entry block / 8-entry group of cdfY
    const float *c64Y, *c8Y; // [ncy_pad], [height/8]; null = binary search
    int ncy;
    // Column search: lines[row * lpr + l] holds columns 6l .. 6l+5.  guide[row * gpitch + k], k = 0..gk,
    // = the number of the row's lines whose LAST cdf entry is < k / gk.  For r2 in [k/gk, (k+1)/gk) the line of the lower bound lies between
    // guide[k] and guide[k+1] (probe_lower_bound_lines): with gk ≈ width/2 that is one line for 2/3 of the lookups and two for the rest.
    const ProbeLine* lines;
    const uint16_t* guide;
    int lpr, gk, gpitch;
};
// The marginal (row) arrays ProbeSample reads: the global ones, or k_shade's per-workgroup LDS copy (≈4.6 KB for a 1024-row
// probe) so that the row search and pdfY never leave the CU
struct ProbeMarg {
    const float *cdfY, *pdfY, *c64Y, *c8Y;
};
PT_DEV ProbeMarg probe_marg_global(const DevProbe& p) { return ProbeMarg{p.cdfY, p.pdfY, p.c64Y, p.c8Y}; }

// :38-46
PT_DEV void probe_dir_to_uv(v3 dir, float& u, float& v) {
    float theta = pt_acosf(clampf(dir.y, -1.0f, 1.0f));
    float phi = (dir.x == 0.0f && dir.z == 0.0f) ? 0.0f : pt_atan2f(dir.z, dir.x);
    u = (kPi + phi) * kInvPi * 0.5f;
    v = theta * kInvPi;
}
// :48-58
PT_DEV v3 probe_uv_to_dir(float u, float v) {
    float theta = v * kPi;
    float phi = u * 2.0f * kPi;
    float st, ct, sp, cp;
    pt_sincosf(theta, &st, &ct);
    pt_sincosf(phi, &sp, &cp);
    return mk3(-st * cp, ct, -st * sp);
}
PT_DEV int clampi(int v, int a, int b) { return v < a ? a : (v > b ? b : v); }
// :61-67
PT_DEV float4 probe_eval(const DevProbe& p, float u, float v) {
    int px = clampi((int)(u * p.width), 0, p.width - 1);
    int py = clampi((int)(v * p.height), 0, p.height - 1);
    return p.data[(size_t)py * p.width + px];
}
// :69-93 ProbePdf: the solid-angle pdf of ProbeSample for a direction.  The reference's device code never calls it (its caller, the MIS term of
// the miss program, is commented out: deviceProgram.cu:214-224); here for the function table (pt_eval_table 8) and a future MIS miss program.
PT_DEV float probe_pdf(const DevProbe& p, v3 d) {
    float u, v;
    probe_dir_to_uv(d, u, v);
    const int col = clampi((int)(u * p.width), 0, p.width - 1);
    const int row = clampi((int)(v * p.height), 0, p.height - 1);
    float pdf = p.pdfX[(size_t)row * p.width + col] * p.pdfY[row];
    const float sinTheta = pt_sinf(v * kPi);
    if (fabsf(sinTheta) < 0.0001f)
        pdf = 0.0f;
    else
        pdf *= (float)p.width * (float)p.height / (2.0f * kPi * kPi * sinTheta);
    return pdf;
}
// :119-136
PT_DEV int lower_bound(const float* __restrict__ array, int lower, int upper, float value) {
    while (lower < upper) {
        int mid = lower + (upper - lower) / 2;
        if (array[mid] < value)
            lower = mid + 1;
        else
            upper = mid;
    }
    return lower;
}
// LowerBound (:119-136) in three dependent memory steps instead of log2(n).  For a non-decreasing array the lower
// bound equals the number of entries < value, so: (1) count the 64-entry blocks whose LAST entry is < value in the
// compact array c64, (2) inside that block count the 8-entry groups whose last entry is < value in c8, (3) count the
// entries < value of the one remaining group.  Same index as the binary search for every array BuildCDF can produce
// (also an all-NaN row of an all-black probe row: every `<` is false → 0, like the reference's loop).  Each step reads
// one or two 64-byte lines with 16-byte loads; the binary search reads 10-11 different lines one after the other.
// (Measured and rejected: scanning the whole 64-entry block instead of steps 2+3 — more VALU work than it saves.)
PT_DEV int count_lt8(const float* __restrict__ p, float v) { // p is 16-byte aligned, 8 floats
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    return (a.x < v ? 1 : 0) + (a.y < v ? 1 : 0) + (a.z < v ? 1 : 0) + (a.w < v ? 1 : 0) + (b.x < v ? 1 : 0) + (b.y < v ? 1 : 0) +
           (b.z < v ? 1 : 0) + (b.w < v ? 1 : 0);
}
PT_DEV int lower_bound_blocked(const float* __restrict__ row, int n, const float* __restrict__ c64, int nc64,
                               const float* __restrict__ c8, float value) {
    // requires n % 64 == 0 (checked on the host; otherwise the plain binary search is used)
    int b = 0;
    if (nc64 <= 32) {
        for (int k = 0; k < nc64; k += 8) b += count_lt8(c64 + k, value); // c64 rows are padded to a multiple of 8 with +inf
    } else {
        b = lower_bound(c64, 0, nc64, value);
    }
    if (b >= nc64) return n;
    const int g = count_lt8(c8 + b * 8, value); // 0..7: the block's last entry is >= value, so g <= 7
    const int base = b * PT_CDF_BLOCK + (g < 7 ? g : 7) * 8;
    return base + count_lt8(row + base, value);
}

PT_DEV float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
PT_DEV float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
// ProbeSample (:138-169) in two stages, so that k_shade can keep the search's first dependent load in flight next to the hit's own chain
// (hit record → triangle → material): begin = the two random numbers, the row (marginal search, LDS in k_shade) and the guide loads;
// end = the candidate lines, column, texel, pdf and direction.  probe_sample runs the two back to back.  (Plain scalars, no struct:
// conditional expressions over struct members are compiled as selects of addresses and park the struct in scratch.)
PT_DEV void probe_search_begin(const DevProbe& p, const ProbeMarg& pm, Rng& rand, int& row_out, int& lo_out, int& hi_out, float& r2_out) {
    float r1, r2;
    sample2d(rand, r1, r2);
    r2_out = r2;
#ifdef PT_EXP_NO_SEARCH
    row_out = (int)(r1 * p.height);
    lo_out = hi_out = (int)(r2 * p.width) / PT_LINE_COLS;
#else
    int row = pm.c64Y ? lower_bound_blocked(pm.cdfY, p.height, pm.c64Y, p.ncy, pm.c8Y, r1) : lower_bound(pm.cdfY, 0, p.height, r1);
    if (row > p.height - 1) row = p.height - 1; // row/col clamped: unreachable for a valid CDF, guards a degenerate probe
    row_out = row;
    int k = (int)(r2 * (float)p.gk); // gk is a power of two and r2 in [0, 1): exact product, k / gk <= r2 < (k + 1) / gk
    k = k < p.gk ? k : p.gk - 1;
    const uint16_t* __restrict__ g = p.guide + (size_t)row * p.gpitch + k;
    lo_out = g[0];
    hi_out = g[1];
#endif
}
// Column lower bound of r2 in the row through the line layout: the same index as the reference's binary search (LowerBound :119-136) for
// every non-decreasing row (and for an all-NaN row: every `<` is false → 0) — the index is 6 l + (entries < r2 in line l) with l = the number
// of lines whose LAST entry is < r2, and r2 in [k/gk, (k+1)/gk) pins l between the two counts the guide stores for the cell's ends.
// Dependent memory steps: guide (2 x u16, adjacent) → the candidate lines' cdf (24 bytes; the second and third only where needed) → the texel, in the line just read.
PT_DEV void probe_search_end(const DevProbe& p, const ProbeMarg& pm, int row, int lo, int hi, float value, v3& dir, v3& color, float& pdf) {
    const ProbeLine* __restrict__ rl = p.lines + (size_t)row * p.lpr;
#ifdef PT_EXP_NO_SEARCH
    const int l = lo;
    int col = (int)(value * p.width);
#else
    const int last = p.lpr - 1;
    lo = lo < last ? lo : last; // count == lpr: every line ends below the cell; the last line then yields column >= width
    hi = hi < last ? hi : last;
    while (hi - lo > 2) { // a guide cell that spans more than three lines (flat stretch of the CDF): halve on the lines' last entries
        const int mid = lo + (hi - lo) / 2;
        if (rl[mid].cdf[PT_LINE_COLS - 1] < value) lo = mid + 1; else hi = mid;
    }
    const int i1 = lo + 1 < hi ? lo + 1 : hi; // candidates lo <= i1 <= hi (equal indices re-read the same line)
    const float4 a0 = ld4(rl[lo].cdf), b0 = ld4(rl[i1].cdf), c0 = ld4(rl[hi].cdf);
    const float2 a1 = ld2(rl[lo].cdf + 4), b1 = ld2(rl[i1].cdf + 4), c1 = ld2(rl[hi].cdf + 4);
    // (branches, so that the second and third line are only loaded by the lanes that advance: k_shade is bound by the address rate of its
    // scattered loads — with value selects, i.e. all three lines loaded by every lane, it is 9 % slower)
    float4 s0 = a0;
    float2 s1 = a1;
    int l = lo;
    if (i1 > lo && a1.y < value) {
        l = i1; s0 = b0; s1 = b1;
        if (hi > i1 && b1.y < value) { l = hi; s0 = c0; s1 = c1; }
    }
    int col = l * PT_LINE_COLS + (s0.x < value ? 1 : 0) + (s0.y < value ? 1 : 0) + (s0.z < value ? 1 : 0) + (s0.w < value ? 1 : 0) +
              (s1.x < value ? 1 : 0) + (s1.y < value ? 1 : 0);
#endif
    if (col > p.width - 1) col = p.width - 1;
    const float4 px = rl[l].px[col - l * PT_LINE_COLS];
    color = mk3(px.x, px.y, px.z);
    pdf = px.w * pm.pdfY[row];
    float u = col / (float)p.width;
    float v = row / (float)p.height;
    float sinTheta = pt_sinf(v * kPi);
    if (sinTheta == 0.0f)
        pdf = 0.0f;
    else
        pdf *= (p.width * p.height) / (2.0f * kPi * kPi * sinTheta);
    dir = probe_uv_to_dir(u, v);
}
PT_DEV void probe_sample(const DevProbe& p, const ProbeMarg& pm, v3& dir, v3& color, float& pdf, Rng& rand) {
    int row, lo, hi;
    float r2;
    probe_search_begin(p, pm, rand, row, lo, hi, r2);
    probe_search_end(p, pm, row, lo, hi, r2, dir, color, pdf);
}

// ------------------------------------------------------------------ software tex2D (SimplePathtracer.cpp:603-654 settings)
// uchar4 array, wrap addressing, bilinear filter, normalised float read, normalised coordinates, no sRGB — the formula of
// the CUDA C Programming Guide appendix "Texture Fetching" (weights in 1.8 fixed point; rounding to nearest assumed).
// Texel layout: 8 x 4-texel tiles, one tile = 128 contiguous bytes = one L2 line, tiles row-major (tiles_x per row, the last column / row
// of tiles padded).  A bilinear footprint (2 x 2 texels) then lies in one line with probability 7/8 * 3/4 and in two otherwise (four at a
// tile corner: 3 %); in a row-major image it always spans two rows = two lines.  Which texels are read, and every operation on them, is
// unchanged: same bits as the row-major restatement in the checker.
struct DevTex {
    const uint32_t* pixel; // tiled (tex_tiled_index)
    int w, h;
    int tiles_x;
};
PT_HD size_t tex_tiled_index(int x, int y, int tiles_x) { return ((size_t)(y >> 2) * (size_t)tiles_x + (size_t)(x >> 3)) * 32u + (size_t)(((y & 3) << 3) | (x & 7)); }
PT_DEV float texel_ch(uint32_t p, int k) { return (float)((p >> (8 * k)) & 0xffu) / 255.0f; }
// u8lut (may be null): the 256 values (float)b / 255.0f, one IEEE division each, computed once per workgroup (k_shade keeps them in LDS) instead of
// sixteen divisions (~10 instructions each) per lookup; the same bits.
PT_DEV float4 tex2d_wrap_linear(const DevTex& tx, float s, float t, const float* u8lut = nullptr) {
    const int W = tx.w, H = tx.h;
    const float x = (s - floorf(s)) * (float)W, y = (t - floorf(t)) * (float)H;
    const float xB = x - 0.5f, yB = y - 0.5f;
    const float fi = floorf(xB), fj = floorf(yB);
    const float alpha = floorf((xB - fi) * 256.0f + 0.5f) * (1.0f / 256.0f);
    const float beta = floorf((yB - fj) * 256.0f + 0.5f) * (1.0f / 256.0f);
    // wrap: x lies in [0, W] (W itself when s - floor(s) rounds to 1), so i0 = floor(x - 0.5) lies in [-1, W - 1] and i1 in [0, W]: one
    // conditional step each instead of the four integer remainders of ((i % W) + W) % W — the same index for every finite coordinate.
    // The unsigned minimum keeps a non-finite coordinate (NaN converts to 0) inside the image.
    int i0 = (int)fi, j0 = (int)fj;
    int i1 = i0 + 1, j1 = j0 + 1;
    i0 = i0 < 0 ? i0 + W : i0; i1 = i1 >= W ? i1 - W : i1;
    j0 = j0 < 0 ? j0 + H : j0; j1 = j1 >= H ? j1 - H : j1;
    i0 = (int)min((uint32_t)i0, (uint32_t)(W - 1)); i1 = (int)min((uint32_t)i1, (uint32_t)(W - 1));
    j0 = (int)min((uint32_t)j0, (uint32_t)(H - 1)); j1 = (int)min((uint32_t)j1, (uint32_t)(H - 1));
    const uint32_t t00 = tx.pixel[tex_tiled_index(i0, j0, tx.tiles_x)], t10 = tx.pixel[tex_tiled_index(i1, j0, tx.tiles_x)],
                   t01 = tx.pixel[tex_tiled_index(i0, j1, tx.tiles_x)], t11 = tx.pixel[tex_tiled_index(i1, j1, tx.tiles_x)];
    float o[4];
    if (u8lut) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = (1.0f - alpha) * (1.0f - beta) * u8lut[(t00 >> (8 * k)) & 0xffu] + alpha * (1.0f - beta) * u8lut[(t10 >> (8 * k)) & 0xffu] +
                   (1.0f - alpha) * beta * u8lut[(t01 >> (8 * k)) & 0xffu] + alpha * beta * u8lut[(t11 >> (8 * k)) & 0xffu];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = (1.0f - alpha) * (1.0f - beta) * texel_ch(t00, k) + alpha * (1.0f - beta) * texel_ch(t10, k) +
                   (1.0f - alpha) * beta * texel_ch(t01, k) + alpha * beta * texel_ch(t11, k);
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}

// ------------------------------------------------------------------ output transforms
// cuda/helpers.h:34-61
PT_DEV uint32_t quantize8(float x) {
    x = clampf(x, 0.0f, 1.0f);
    uint32_t q = (uint32_t)(x * 256.0f);
    return q < 255u ? q : 255u;
}
PT_DEV float to_srgb1(float c) {
    float invGamma = 1.0f / 2.4f;
    float powed = pt_powf(c, invGamma);
    return c < 0.0031308f ? 12.92f * c : 1.055f * powed - 0.055f;
}
PT_DEV uint32_t make_color(v3 c) {
    uint32_t r = quantize8(to_srgb1(clampf(c.x, 0.0f, 1.0f)));
    uint32_t g = quantize8(to_srgb1(clampf(c.y, 0.0f, 1.0f)));
    uint32_t b = quantize8(to_srgb1(clampf(c.z, 0.0f, 1.0f)));
    return r | (g << 8) | (b << 16) | (255u << 24);
}
