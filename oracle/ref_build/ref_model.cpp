// ref_model.cpp — second translation unit of oracle/_ref/libptref.so: the REFERENCE's own scene ingestion
// (HelloPathtracing_original/Model.cpp: addVertex :51-84, loadTexture :88-135, loadOBJ :137-212, addBox :214-286, with the
// tinyobjloader and stb_image it vendors under support/), compiled from where it lies under /root/reference and never copied.
// Test infrastructure only: validates optixpathtracer_amd/objloader.py and scenes.add_box and generates
// tests/golden/ref_model.npz (tests/golden/make_golden.py).  /root/reference does not exist on the GPU box.
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <cuda_runtime.h>
using std::max; using std::min;
#define STB_IMAGE_IMPLEMENTATION
#include "support/stb/stb_image.h"
#undef STB_IMAGE_IMPLEMENTATION
#include "Model.cpp"  // HelloPathtracing_original/Model.cpp

extern "C" {
// Models are leaked on purpose: Texture::~Texture delete[]s what stbi_load malloc'ed (Model.h:22-25) — not to be run.
void* refm_load_obj(const char* path) {
    try { return loadOBJ(path); } catch (const std::exception&) { return nullptr; }
}
void* refm_new_model() { return new Model; }
void refm_add_box(void* m, const void* material, const float pos[3], const float ext[3]) {
    Material mat; memcpy(&mat, material, sizeof(Material));
    addBox((Model*)m, mat, make_float3(pos[0], pos[1], pos[2]), make_float3(ext[0], ext[1], ext[2]));
}
int refm_num_meshes(void* m) { return (int)((Model*)m)->meshes.size(); }
int refm_num_textures(void* m) { return (int)((Model*)m)->textures.size(); }
void refm_mesh_sizes(void* m, int i, int out[5]) {
    TriangleMesh* t = ((Model*)m)->meshes[i];
    out[0] = (int)t->vertex.size(); out[1] = (int)t->normal.size(); out[2] = (int)t->texcoord.size(); out[3] = (int)t->index.size(); out[4] = t->diffuseTextureID;
}
void refm_mesh_copy(void* m, int i, float* v, float* n, float* tc, uint32_t* idx, void* material) {
    TriangleMesh* t = ((Model*)m)->meshes[i];
    if (!t->vertex.empty()) memcpy(v, t->vertex.data(), t->vertex.size() * sizeof(float3));
    if (!t->normal.empty()) memcpy(n, t->normal.data(), t->normal.size() * sizeof(float3));
    if (!t->texcoord.empty()) memcpy(tc, t->texcoord.data(), t->texcoord.size() * sizeof(float2));
    if (!t->index.empty()) memcpy(idx, t->index.data(), t->index.size() * sizeof(uint3));
    memcpy(material, &t->material, sizeof(Material));
}
// main.cpp:146-156 loadProbe: stbi_loadf(file, &w, &h, &n, 4) of a Radiance .hdr.  Returns 1 and the size; the pixels (w*h float4) are copied into
// `out` when it is non-null and `cap_floats` suffices (call twice: size, then data).
int refm_loadf(const char* path, int res[2], float* out, size_t cap_floats) {
    int w = 0, h = 0, n = 0;
    float* data = stbi_loadf(path, &w, &h, &n, 4);
    if (!data) return 0;
    res[0] = w; res[1] = h;
    if (out && cap_floats >= (size_t)w * h * 4) memcpy(out, data, sizeof(float) * (size_t)w * h * 4);
    stbi_image_free(data);
    return 1;
}
void refm_texture_size(void* m, int i, int res[2]) { Texture* t = ((Model*)m)->textures[i]; res[0] = t->resolution.x; res[1] = t->resolution.y; }
void refm_texture_copy(void* m, int i, uint32_t* px) { Texture* t = ((Model*)m)->textures[i]; memcpy(px, t->pixel, (size_t)t->resolution.x * t->resolution.y * 4); }
}
