// ref_driver.cpp — thin extern "C" shims over the REFERENCE's own headers, compiled
// from where they lie under /root/reference (never copied).  Test infrastructure:
// builds oracle/_ref/libptref.so, used in this container only to (a) validate
// oracle/pt_oracle.c function by function and (b) generate tests/golden/*.npz via
// tests/golden/make_golden.py.  /root/reference does not exist on the GPU box.
//
// Only headers that compile with what the image already holds are used (CUDA's
// host-side vector_types.h etc. ship inside the Triton wheel).  Disney.cuh,
// LaunchParams.h, Probe.h, CUDABuffer.h and deviceProgram.cu include <optix.h> /
// <optix_device.h>, which the image lacks; no stand-ins are written for them, so
// those parts of the reference are NOT in this library ("oracle").
#include <cfloat>
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cuda_runtime.h>
using std::max; using std::min; using std::abs; using std::isfinite;
// Under nvcc the reference's unqualified sqrt/cos/sin/atan2/... calls on float arguments bind to CUDA's
// float overloads (device semantics).  Make the same overloads visible here, otherwise g++ binds them to
// the C library's double versions and the host build would not compute what the device code computes.
using std::sqrt; using std::cos; using std::sin; using std::tan; using std::acos; using std::atan2;
using std::exp; using std::log; using std::pow; using std::fabs;
#include "maths.h"        // HelloPathtracing_original/maths.h
#include "sample.h"       // HelloPathtracing_original/sample.h
#include "Probe.cuh"      // HelloPathtracing_original/Probe.cuh
#include "Material.h"     // HelloPathtracing_original/Material.h ("Maths.h" resolves to OptixUtils/Maths.h on a case-sensitive fs)
#include <random.h>       // cuda/random.h
#include <cuda/helpers.h> // cuda/helpers.h
#include <sutil/Camera.h>

extern "C" {
uint32_t ref_tea4(uint32_t a, uint32_t b) { return tea<4>(a, b); }
uint32_t ref_lcg(uint32_t* s) { return lcg(*s); }
float ref_rnd(uint32_t* s) { return rnd(*s); }
void ref_random_init(uint32_t* st, uint32_t seed) { Random r(seed); st[0] = r.seed1; st[1] = r.seed2; }
uint32_t ref_rand(uint32_t* st) { Random r; r.seed1 = st[0]; r.seed2 = st[1]; uint32_t v = r.Rand(); st[0] = r.seed1; st[1] = r.seed2; return v; }
float ref_randf(uint32_t* st) { Random r; r.seed1 = st[0]; r.seed2 = st[1]; float v = r.Randf(); st[0] = r.seed1; st[1] = r.seed2; return v; }
void ref_basis_from_vector(const float w[3], float u[3], float v[3]) {
    float3 uu, vv; BasisFromVector(make_float3(w[0], w[1], w[2]), &uu, &vv);
    u[0] = uu.x; u[1] = uu.y; u[2] = uu.z; v[0] = vv.x; v[1] = vv.y; v[2] = vv.z;
}
void ref_uniform_sample_hemisphere(uint32_t seed, float d[3]) { Random r(seed); float3 v = UniformSampleHemisphere(r); d[0] = v.x; d[1] = v.y; d[2] = v.z; }
void ref_cosine_sample_hemisphere(float u1, float u2, float d[3]) { float3 v = CosineSampleHemisphere(u1, u2); d[0] = v.x; d[1] = v.y; d[2] = v.z; }
void ref_probe_dir_to_uv(const float d[3], float uv[2]) { float2 r = ProbeDirToUV(make_float3(d[0], d[1], d[2])); uv[0] = r.x; uv[1] = r.y; }
void ref_probe_uv_to_dir(const float uv[2], float d[3]) { float3 r = ProbeUVToDir(make_float2(uv[0], uv[1])); d[0] = r.x; d[1] = r.y; d[2] = r.z; }
static Probe mk_probe(int w, int h, const float* data, const float* pdfX, const float* cdfX, const float* pdfY, const float* cdfY) {
    Probe p; p.width = w; p.height = h; p.data = (Color*)data; p.offset = make_float3(0.f);
    p.pdfValuesX = (float*)pdfX; p.cdfValuesX = (float*)cdfX; p.pdfValuesY = (float*)pdfY; p.cdfValuesY = (float*)cdfY; return p;
}
void ref_probe_eval(int w, int h, const float* data, const float uv[2], float rgba[4]) {
    Probe p = mk_probe(w, h, data, 0, 0, 0, 0); float4 c = ProbeEval(p, make_float2(uv[0], uv[1])); rgba[0] = c.x; rgba[1] = c.y; rgba[2] = c.z; rgba[3] = c.w;
}
void ref_probe_sample(int w, int h, const float* data, const float* pdfX, const float* cdfX, const float* pdfY, const float* cdfY,
                      uint32_t seed, float dir[3], float color[3], float* pdf, uint32_t st[2]) {
    Probe p = mk_probe(w, h, data, pdfX, cdfX, pdfY, cdfY); Random r(seed); float3 d, c;
    ProbeSample(p, d, c, *pdf, r); dir[0] = d.x; dir[1] = d.y; dir[2] = d.z; color[0] = c.x; color[1] = c.y; color[2] = c.z; st[0] = r.seed1; st[1] = r.seed2;
}
// Probe.cuh:69-93 (not called by the reference's device code: its only caller, the MIS term of the miss program, is commented out, deviceProgram.cu:214-224)
float ref_probe_pdf(int w, int h, const float* data, const float* pdfX, const float* pdfY, const float d[3]) {
    Probe p = mk_probe(w, h, data, pdfX, 0, pdfY, 0); return ProbePdf(p, make_float3(d[0], d[1], d[2]));
}
float ref_luminance(const float c[4]) { return Luminance(make_float4(c[0], c[1], c[2], c[3])); }
uint32_t ref_make_color(const float c[3]) { uchar4 q = make_color(make_float3(c[0], c[1], c[2])); return q.x | (q.y << 8) | (q.z << 16) | ((uint32_t)q.w << 24); }
size_t ref_sizeof_material() { return sizeof(Material); }
void ref_material_default(void* out) { Material m; memcpy(out, &m, sizeof(Material)); }
float ref_material_ior(const void* mat) { Material m; memcpy(&m, mat, sizeof(Material)); return m.GetIndexOfRefraction(); }
void ref_uvw_frame(const float e[3], const float a[3], const float up[3], float fovY, float aspect, float U[3], float V[3], float W[3]) {
    sutil::Camera c(make_float3(e[0], e[1], e[2]), make_float3(a[0], a[1], a[2]), make_float3(up[0], up[1], up[2]), fovY, aspect);
    float3 u, v, w; c.UVWFrame(u, v, w); U[0] = u.x; U[1] = u.y; U[2] = u.z; V[0] = v.x; V[1] = v.y; V[2] = v.z; W[0] = w.x; W[1] = w.y; W[2] = w.z;
}
void ref_faceforward(const float n[3], const float i[3], float out[3]) { float3 N = make_float3(n[0], n[1], n[2]); float3 r = faceforward(N, make_float3(i[0], i[1], i[2]), N); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void ref_normalize(const float n[3], float out[3]) { float3 r = normalize(make_float3(n[0], n[1], n[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
// maths.h:144-156 (note the double-precision 1.0 / sqrt(m)), sutil/vec_math.h:500-503 and :513-516, cuda/helpers.h:34-42
void ref_safe_normalize(const float a[3], float out[3]) { float3 r = SafeNormalize(make_float3(a[0], a[1], a[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void ref_lerp3(const float a[3], const float b[3], float t, float out[3]) { float3 r = lerp(make_float3(a[0], a[1], a[2]), make_float3(b[0], b[1], b[2]), t); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void ref_clamp3(const float v[3], float lo, float hi, float out[3]) { float3 r = clamp(make_float3(v[0], v[1], v[2]), lo, hi); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void ref_to_srgb(const float c[3], float out[3]) { float3 r = toSRGB(make_float3(c[0], c[1], c[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
}
