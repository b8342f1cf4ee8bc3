// pt_objload.cpp — pt_load_obj: the reference's scene ingestion, loadOBJ (HelloPathtracing_original/Model.cpp:137-212), as native host
// code behind the C ABI (include/pt_amd.h).  Host only: no HIP call, no GPU needed.
//
// The reference parses with tinyobjloader 2.0.0 (support/tinyobjloader/tiny_obj_loader.h, LoadObj(..., triangulate = true)); neither that
// header nor stb_image travels with this repository, so what decides loadOBJ's arrays is restated here, function by function, from the
// line-cited Python restatement optixpathtracer_amd/objloader.py (which tests/test_objloader.py pins bit for bit to the reference's own
// Model.cpp compiled from /root/reference): tryParseDouble (:836-960), safeGetline (:731-764), parseTriple / fixIndex (:770-790,
// :1100-1148), the `f` / `usemtl` / `mtllib` / `g` / `o` commands (:2407-2601), LoadMtl (:1688-2077) with ParseTextureNameAndOption
// (:1186-1265), the ear-clipping triangulation of exportGroupsToShape in float arithmetic (:1365-1580) — and Model.cpp's own rules:
// one mesh per (shape, material id) in ascending id order, addVertex's back-filled / zero-padded normals and texcoords (:51-84), corners
// added last first (:186-188), ONE knownVertices map per shape shared by its materials (:176; per_mesh_vertex_map != 0 gives every
// mesh its own — what a renderer needs), one knownTextures map per shape (:177).
// Images are not decoded here (the reference uses stb_image): a mesh carries the number of its texture REFERENCE — (shape, file name)
// pairs in loadTexture's order of first use — and the caller decodes the files, drops the unreadable ones and renumbers (loadTexture
// returns -1 for those, Model.cpp:88-135).  This file is compiled with -ffp-contract=off: the triangulation's float expressions round
// once per operation like the reference's.
#include <sys/stat.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/pt_amd.h"

namespace {

struct Idx {
    int v, vt, vn;
};
struct Mtl {
    std::string name;
    float diffuse[3] = {0.f, 0.f, 0.f}, emission[3] = {0.f, 0.f, 0.f};
    std::string diffuse_texname;
};
struct Shape {
    std::vector<Idx> indices; // three per triangle
    std::vector<int> mat_ids; // one per triangle
};

inline bool is_digit(char c) { return c >= '0' && c <= '9'; }
inline bool is_blank(char c) { return c == ' ' || c == '\t'; }

// A decimal field of an OBJ / MTL line, read the way tinyobjloader 2.0.0 reads it (tryParseDouble, tiny_obj_loader.h:836-960).  Restated in
// this file's own terms, not copied: what has to agree with the reference is the grammar and the SEQUENCE of double operations, because
// loadOBJ's float arrays are compared bit for bit (tests/test_objloader.py).
//   grammar    [+|-] digits [ . digits ] [ (e|E) [+|-] digits ]   or   [+|-] . digits [ exponent ]
//              anything else after a complete whole part or fraction ends the number (the field still parses); an exponent marker without
//              digits, or no digit before it, fails the field
//   arithmetic whole part: v = v * 10 + d per digit; fraction digit at decimal place k: v += d * 10^-k, the first seven powers from a table of
//              literals, later ones from pow(10, -k); exponent e != 0: ldexp(v * pow(5, e), e); the sign is a final multiplication by +-1
struct NumberScan {
    const char* at;
    const char* stop;
    bool here(char c) const { return at != stop && *at == c; }
    bool digit_here() const { return at != stop && is_digit(*at); }
    int take_digit() { return *at++ - '0'; }
};
bool try_parse_double(const char* s, const char* s_end, double* result) {
    if (s >= s_end) return false;
    static const double decimal_place[8] = {1e0, 1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e-7};
    NumberScan in{s, s_end};
    const bool negative = in.here('-');
    if (negative || in.here('+')) ++in.at;
    else if (!in.digit_here() && !in.here('.')) return false;
    double v = 0.0;
    if (!in.here('.')) { // a whole part, at least one digit
        if (!in.digit_here()) return false;
        while (in.digit_here()) {
            v *= 10;
            v += in.take_digit();
        }
    }
    if (in.here('.')) {
        ++in.at;
        for (int place = 1; in.digit_here(); ++place) v += in.take_digit() * (place < 8 ? decimal_place[place] : std::pow(10.0, -place));
    }
    int power = 0;
    if (in.here('e') || in.here('E')) {
        ++in.at;
        const bool shrink = in.here('-');
        if (shrink || in.here('+')) ++in.at;
        if (!in.digit_here()) return false;
        // (the reference's `exponent *= 10` overflows a signed int on a field like 1e99999999999 — undefined behaviour there; beyond 10^8 the
        // value is +-inf, 0 or NaN whatever the digits, so the accumulation saturates)
        while (in.digit_here()) {
            const int d = in.take_digit();
            if (power < 100000000) power = power * 10 + d;
        }
        if (shrink) power = -power;
    }
    *result = (negative ? -1 : 1) * (power ? std::ldexp(v * std::pow(5.0, power), power) : v);
    return true;
}

// the `const char** token` the parser walks along a line
struct Cursor {
    const char* p;
    const char* end; // end of the line (no \r or \n inside)
    void skip_blank() {
        while (p < end && is_blank(*p)) ++p;
    }
    // advances to the first character of `stops` (or the end) and returns the word's start
    const char* word(const char* stops, size_t* len) {
        const char* b = p;
        for (; p < end; ++p) {
            bool stop = false;
            for (const char* q = stops; *q; ++q) stop |= (*q == *p);
            if (stop) break;
        }
        *len = (size_t)(p - b);
        return b;
    }
    float real(double def = 0.0) { // parseReal (:962-970)
        skip_blank();
        size_t n;
        const char* b = word(" \t\r", &n);
        double v = def;
        if (!try_parse_double(b, b + n, &v)) v = def;
        return static_cast<float>(v);
    }
    bool at_end() const { return p >= end || *p == '\r' || *p == '\n' || *p == '\0'; }
};

// safeGetline (:731-764) over a whole file: lines end at \n, \r\n or a lone \r
void split_lines(const std::string& text, std::vector<std::pair<const char*, const char*>>& out) {
    const char* b = text.data();
    const char* e = b + text.size();
    const char* s = b;
    for (const char* p = b; p < e; ++p) {
        if (*p == '\n') {
            out.emplace_back(s, p);
            s = p + 1;
        } else if (*p == '\r') {
            out.emplace_back(s, p);
            if (p + 1 < e && p[1] == '\n') ++p;
            s = p + 1;
        }
    }
    if (s < e) out.emplace_back(s, e);
}
bool read_file(const std::string& path, std::string& text) {
    struct stat st;
    if (::stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) return false; // a directory opens with fopen and reports a huge size
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    long n = -1;
    if (std::fseek(f, 0, SEEK_END) == 0) n = std::ftell(f);
    if (n < 0 || std::fseek(f, 0, SEEK_SET) != 0) {
        std::fclose(f);
        return false;
    }
    text.resize((size_t)n);
    const size_t got = n > 0 ? std::fread(&text[0], 1, (size_t)n, f) : 0;
    std::fclose(f);
    text.resize(got);
    return true;
}

// n consecutive parseReal calls on a field list: a missing or malformed field takes the default
void reals(const char* b, const char* e, int n, float* out) {
    Cursor c{b, e};
    for (int k = 0; k < n; ++k) out[k] = c.real(0.0);
}

// fixIndex (:770-790): false = the reference's `return false` (index 0)
bool fix_index(int idx, int n, int* ret) {
    if (idx > 0) { *ret = idx - 1; return true; }
    if (idx == 0) return false;
    *ret = n + idx;
    return true;
}
int atoi_at(const Cursor& c) { // atoi on the rest of the line (the line buffer is NUL-free up to its end; copy bounded)
    char buf[32];
    size_t n = (size_t)(c.end - c.p);
    if (n > sizeof buf - 1) n = sizeof buf - 1;
    std::memcpy(buf, c.p, n);
    buf[n] = 0;
    return std::atoi(buf);
}
// parseTriple (:1100-1148) -> (v, vt, vn) zero-based, -1 = absent
bool parse_triple(Cursor& c, int nv, int nvn, int nvt, Idx* out) {
    Idx r{-1, -1, -1};
    size_t len;
    if (!fix_index(atoi_at(c), nv, &r.v)) return false;
    c.word("/ \t\r", &len);
    if (c.p >= c.end || *c.p != '/') { *out = r; return true; }
    ++c.p;
    if (c.p < c.end && *c.p == '/') { // i//k
        ++c.p;
        if (!fix_index(atoi_at(c), nvn, &r.vn)) return false;
        c.word("/ \t\r", &len);
        *out = r;
        return true;
    }
    if (!fix_index(atoi_at(c), nvt, &r.vt)) return false; // i/j or i/j/k
    c.word("/ \t\r", &len);
    if (c.p >= c.end || *c.p != '/') { *out = r; return true; }
    ++c.p;
    if (!fix_index(atoi_at(c), nvn, &r.vn)) return false;
    c.word("/ \t\r", &len);
    *out = r;
    return true;
}

bool starts_with(const char* b, const char* e, const char* w) {
    const size_t n = std::strlen(w);
    return (size_t)(e - b) >= n && std::memcmp(b, w, n) == 0;
}

// ParseTextureNameAndOption (:1186-1265): options with their arguments, then everything up to the end of the line is the name
std::string texture_name(const char* b, const char* e) {
    static const struct { const char* opt; int nargs; } opts[] = {{"-blendu", 1}, {"-blendv", 1}, {"-clamp", 1}, {"-boost", 1}, {"-bm", 1}, {"-o", 3}, {"-s", 3},
                                                                 {"-t", 3}, {"-texres", 1}, {"-imfchan", 1}, {"-mm", 2}, {"-colorspace", 1}};
    Cursor c{b, e};
    std::string name;
    size_t len;
    while (!c.at_end()) {
        c.skip_blank();
        const char* p = c.p;
        int hit = -1;
        for (size_t k = 0; k < sizeof opts / sizeof opts[0]; ++k) {
            const size_t n = std::strlen(opts[k].opt);
            if (starts_with(p, e, opts[k].opt) && p + n < e && is_blank(p[n])) { hit = (int)k; break; }
        }
        if (hit < 0 && starts_with(p, e, "-type") && p + 5 < e && is_blank(p[5])) {
            c.p = p + 5;
            c.skip_blank();
            c.word(" \t\r", &len);
            continue;
        }
        if (hit < 0) {
            name.assign(p, e);
            c.p = e;
            break;
        }
        c.p = p + std::strlen(opts[hit].opt) + (std::strcmp(opts[hit].opt, "-texres") == 0 ? 0 : 1);
        for (int k = 0; k < opts[hit].nargs; ++k) {
            c.skip_blank();
            c.word(" \t\r", &len);
        }
    }
    return name;
}

// LoadMtl (:1688-2077), the fields loadOBJ reads: diffuse, emission, diffuse_texname
void load_mtl(const std::string& path, std::vector<Mtl>& materials, std::map<std::string, int>& material_map) {
    std::string text;
    if (!read_file(path, text)) return;
    std::vector<std::pair<const char*, const char*>> lines;
    split_lines(text, lines);
    Mtl mat;
    bool has_kd = false;
    for (auto& ln : lines) {
        const char* b = ln.first;
        const char* e = ln.second;
        while (e > b && is_blank(e[-1])) --e; // trailing blanks
        if (b == e) continue;
        while (b < e && is_blank(*b)) ++b;
        if (b == e || *b == '#') continue;
        if (starts_with(b, e, "newmtl") && e - b > 6 && is_blank(b[6])) {
            if (!mat.name.empty()) {
                material_map.insert({mat.name, (int)materials.size()});
                materials.push_back(mat);
            }
            mat = Mtl();
            mat.name.assign(b + 7, e);
            continue;
        }
        if (e - b > 2 && b[0] == 'K' && (b[1] == 'd' || b[1] == 'e') && is_blank(b[2])) {
            Cursor c{b + 2, e};
            float rgb[3];
            rgb[0] = c.real(); rgb[1] = c.real(); rgb[2] = c.real();
            if (b[1] == 'd') {
                std::memcpy(mat.diffuse, rgb, sizeof rgb);
                has_kd = true;
            } else {
                std::memcpy(mat.emission, rgb, sizeof rgb);
            }
            continue;
        }
        if (starts_with(b, e, "map_Kd") && e - b > 6 && is_blank(b[6])) {
            mat.diffuse_texname = texture_name(b + 7, e);
            if (!has_kd) mat.diffuse[0] = mat.diffuse[1] = mat.diffuse[2] = 0.6f;
            continue;
        }
    }
    material_map.insert({mat.name, (int)materials.size()});
    materials.push_back(mat);
}

// exportGroupsToShape's ear clipping (:1382-1580) for one polygon.  All arithmetic in float, one operation at a time.
void triangulate(const std::vector<Idx>& face, const std::vector<float>& V, std::vector<Idx>& out) {
    const int nverts = (int)(V.size() / 3);
    size_t npolys = face.size();
    const size_t nfl = V.size();
    int axes[2] = {1, 2};
    const float eps = 1.1920929e-07f; // std::numeric_limits<float>::epsilon()
    auto vx = [&](int vi, int a) -> float { return V[(size_t)3 * vi + a]; };
    for (size_t k = 0; k < npolys; ++k) {
        const int vi0 = face[k % npolys].v, vi1 = face[(k + 1) % npolys].v, vi2 = face[(k + 2) % npolys].v;
        if ((size_t)(3 * (long long)vi0 + 2) >= nfl || (size_t)(3 * (long long)vi1 + 2) >= nfl || (size_t)(3 * (long long)vi2 + 2) >= nfl || vi0 < 0 || vi1 < 0 || vi2 < 0) continue;
        const float e0x = vx(vi1, 0) - vx(vi0, 0), e0y = vx(vi1, 1) - vx(vi0, 1), e0z = vx(vi1, 2) - vx(vi0, 2);
        const float e1x = vx(vi2, 0) - vx(vi1, 0), e1y = vx(vi2, 1) - vx(vi1, 1), e1z = vx(vi2, 2) - vx(vi1, 2);
        const float cx = std::fabs(e0y * e1z - e0z * e1y);
        const float cy = std::fabs(e0z * e1x - e0x * e1z);
        const float cz = std::fabs(e0x * e1y - e0y * e1x);
        if (cx > eps || cy > eps || cz > eps) {
            if (cx > cy && cx > cz) {
            } else {
                axes[0] = 0;
                if (cz > cx && cz > cy) axes[1] = 1;
            }
            break;
        }
    }
    const int a0 = axes[0], a1 = axes[1];
    float area = 0.f;
    for (size_t k = 0; k < npolys; ++k) {
        const int vi0 = face[k % npolys].v, vi1 = face[(k + 1) % npolys].v;
        if (!(vi0 >= 0 && vi0 < nverts && vi1 >= 0 && vi1 < nverts)) continue;
        area = area + (vx(vi0, a0) * vx(vi1, a1) - vx(vi0, a1) * vx(vi1, a0)) * 0.5f;
    }
    std::vector<Idx> rem(face);
    size_t guess = 0;
    size_t remaining_iter = face.size();
    size_t prev_remaining = rem.size();
    while (rem.size() > 3 && remaining_iter > 0) {
        npolys = rem.size();
        if (guess >= npolys) guess -= npolys;
        if (prev_remaining != npolys) {
            prev_remaining = npolys;
            remaining_iter = npolys;
        } else {
            remaining_iter -= 1;
        }
        Idx ind[3];
        float px[3], py[3];
        for (int k = 0; k < 3; ++k) {
            ind[k] = rem[(guess + k) % npolys];
            const int vi = ind[k].v;
            if (!(vi >= 0 && vi < nverts)) {
                px[k] = 0.f;
                py[k] = 0.f;
            } else {
                px[k] = vx(vi, a0);
                py[k] = vx(vi, a1);
            }
        }
        const float e0x = px[1] - px[0], e0y = py[1] - py[0];
        const float e1x = px[2] - px[1], e1y = py[2] - py[1];
        const float cross = e0x * e1y - e0y * e1x;
        if (cross * area < 0.0f) {
            guess += 1;
            continue;
        }
        bool overlap = false;
        for (size_t other = 3; other < npolys; ++other) {
            const size_t idx = (guess + other) % npolys;
            const int ovi = rem[idx].v;
            if (!(ovi >= 0 && ovi < nverts)) continue;
            const float tx = vx(ovi, a0), ty = vx(ovi, a1);
            bool c = false; // pnpoly (:1353-1363)
            int j = 2;
            for (int i = 0; i < 3; ++i) {
                if (((py[i] > ty) != (py[j] > ty)) && (tx < (px[j] - px[i]) * (ty - py[i]) / (py[j] - py[i]) + px[i])) c = !c;
                j = i;
            }
            if (c) {
                overlap = true;
                break;
            }
        }
        if (overlap) {
            guess += 1;
            continue;
        }
        out.push_back(ind[0]);
        out.push_back(ind[1]);
        out.push_back(ind[2]);
        rem.erase(rem.begin() + (long)((guess + 1) % npolys));
    }
    if (rem.size() == 3) {
        out.push_back(rem[0]);
        out.push_back(rem[1]);
        out.push_back(rem[2]);
    }
}

struct Parsed {
    std::vector<float> V, VN, VT;
    std::vector<Shape> shapes;
    std::vector<Mtl> materials;
};

// tinyobj::LoadObj(..., triangulate = true) (:2158-2745)
bool parse_obj(const std::string& obj_file, const std::string& mtl_basedir, Parsed& P, std::string& err) {
    std::string text;
    if (!read_file(obj_file, text)) {
        err = "Could not read OBJ model from " + obj_file + " : Cannot open file [" + obj_file + "]";
        return false;
    }
    std::vector<std::pair<const char*, const char*>> lines;
    split_lines(text, lines);
    std::map<std::string, int> material_map;
    Shape shape;
    std::vector<std::vector<Idx>> faces; // prim_group.faceGroup
    int material = -1;
    std::string base = mtl_basedir;
    if (!base.empty() && base.back() != '/') base += "/";

    auto do_export = [&]() -> bool {
        if (faces.empty()) return false;
        std::vector<Idx> tris;
        for (const std::vector<Idx>& face : faces) {
            if (face.size() < 3) continue;
            if (face.size() == 3) {
                shape.indices.insert(shape.indices.end(), face.begin(), face.end());
                shape.mat_ids.push_back(material);
                continue;
            }
            tris.clear();
            triangulate(face, P.V, tris);
            shape.indices.insert(shape.indices.end(), tris.begin(), tris.end());
            for (size_t k = 0; k < tris.size() / 3; ++k) shape.mat_ids.push_back(material);
        }
        return true;
    };

    size_t line_no = 0;
    for (auto& ln : lines) {
        ++line_no;
        const char* b = ln.first;
        const char* e = ln.second;
        if (b == e) continue;
        while (b < e && is_blank(*b)) ++b;
        if (b == e || *b == '#') continue;
        const char c1 = e - b > 1 ? b[1] : '\0', c2 = e - b > 2 ? b[2] : '\0';
        if (b[0] == 'v' && is_blank(c1)) {
            float v[3];
            reals(b + 2, e, 3, v);
            P.V.insert(P.V.end(), v, v + 3);
            continue;
        }
        if (b[0] == 'v' && c1 == 'n' && is_blank(c2)) {
            float v[3];
            reals(b + 3, e, 3, v);
            P.VN.insert(P.VN.end(), v, v + 3);
            continue;
        }
        if (b[0] == 'v' && c1 == 't' && is_blank(c2)) {
            float v[2];
            reals(b + 3, e, 2, v);
            P.VT.insert(P.VT.end(), v, v + 2);
            continue;
        }
        if (b[0] == 'f' && is_blank(c1)) {
            Cursor c{b + 2, e};
            c.skip_blank();
            faces.emplace_back();
            std::vector<Idx>& face = faces.back();
            while (!c.at_end()) {
                Idx t;
                if (!parse_triple(c, (int)(P.V.size() / 3), (int)(P.VN.size() / 3), (int)(P.VT.size() / 2), &t)) {
                    err = "Could not read OBJ model from " + obj_file + " : Failed parse `f' line(e.g. zero value for face index. line " + std::to_string(line_no) + ".)";
                    return false;
                }
                face.push_back(t);
                while (c.p < c.end && (is_blank(*c.p) || *c.p == '\r')) ++c.p;
            }
            continue;
        }
        if (starts_with(b, e, "usemtl")) {
            Cursor c{b + 6, e};
            c.skip_blank();
            size_t len;
            const char* w = c.word(" \t\r", &len);
            const auto it = material_map.find(std::string(w, len));
            const int new_id = it == material_map.end() ? -1 : it->second;
            if (new_id != material) {
                do_export();
                faces.clear();
                material = new_id;
            }
            continue;
        }
        if (starts_with(b, e, "mtllib") && e - b > 6 && is_blank(b[6])) {
            const char* p = b + 7;
            if (p < e) { // file names split at single blanks; the first one that opens is read
                const char* s = p;
                for (;; ++p) {
                    if (p == e || *p == ' ') {
                        const std::string fn(s, p);
                        const std::string path = base.empty() ? fn : base + fn;
                        struct stat sb;
                        if (::stat(path.c_str(), &sb) == 0 && S_ISREG(sb.st_mode)) {
                            load_mtl(path, P.materials, material_map);
                            break;
                        }
                        if (p == e) break;
                        s = p + 1;
                    }
                }
            }
            continue;
        }
        if ((b[0] == 'g' || b[0] == 'o') && is_blank(c1)) {
            do_export();
            if (!shape.indices.empty()) P.shapes.push_back(std::move(shape));
            shape = Shape();
            faces.clear();
            continue;
        }
        // l / p / t / s / vw and unknown commands do not change what loadOBJ reads
    }
    const bool ret = do_export();
    if (ret || !shape.indices.empty()) P.shapes.push_back(std::move(shape));
    return true;
}

struct IdxHash {
    size_t operator()(const Idx& k) const {
        uint64_t h = (uint64_t)(uint32_t)k.v * 0x9E3779B97F4A7C15ull;
        h ^= ((uint64_t)(uint32_t)k.vt + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
        h ^= ((uint64_t)(uint32_t)k.vn + 0x165667B1ull) * 0xD6E8FEB86659FD93ull;
        return (size_t)(h ^ (h >> 29));
    }
};
struct IdxEq {
    bool operator()(const Idx& a, const Idx& b) const { return a.v == b.v && a.vt == b.vt && a.vn == b.vn; }
};

struct Mesh {
    std::vector<float> vertex, normal, texcoord;
    std::vector<uint32_t> index;
    pt_material material;
    int32_t tex_ref = -1;
};

pt_material default_material() { // Material() (Material.h:13-36)
    pt_material m;
    std::memset(&m, 0, sizeof m);
    m.color[0] = m.color[1] = m.color[2] = 0.6f;
    m.specular = 0.5f;
    m.roughness = 1.0f;
    m.clearcoatGloss = 1.0f;
    m.bumpTile[0] = m.bumpTile[1] = m.bumpTile[2] = 10.0f;
    return m;
}

thread_local std::string g_obj_error;

} // namespace

struct pt_obj {
    std::vector<Mesh> meshes;
    std::vector<std::string> texture_paths; // texture references in loadTexture's order of first use
};

extern "C" const char* pt_obj_last_error(void) { return g_obj_error.c_str(); }

static int load_obj(const char* obj_path, int per_mesh_vertex_map, pt_obj** out) {
    const std::string obj_file(obj_path);
    const size_t slash = obj_file.rfind('/');
    const std::string model_dir = slash == std::string::npos ? std::string() : obj_file.substr(0, slash + 1);
    Parsed P;
    if (!parse_obj(obj_file, model_dir, P, g_obj_error)) return PT_ERR_INVALID;
    std::unique_ptr<pt_obj> obj(new pt_obj);
    const int nV = (int)(P.V.size() / 3), nVN = (int)(P.VN.size() / 3), nVT = (int)(P.VT.size() / 2);
    for (const Shape& sh : P.shapes) {
        std::unordered_map<Idx, int, IdxHash, IdxEq> known_vertices;
        std::map<std::string, int> known_textures;
        const std::set<int> material_ids(sh.mat_ids.begin(), sh.mat_ids.end());
        for (const int material_id : material_ids) {
            if (per_mesh_vertex_map) known_vertices.clear();
            Mesh mesh;
            mesh.material = default_material();
            for (size_t face_id = 0; face_id < sh.mat_ids.size(); ++face_id) {
                if (sh.mat_ids[face_id] != material_id) continue;
                uint32_t ids[3];
                // the three addVertex calls are ARGUMENTS of one make_uint3 call (Model.cpp:186-188): evaluated right to left — corner 2 first
                for (int corner = 2; corner >= 0; --corner) { // addVertex (:51-84)
                    const Idx key = sh.indices[3 * face_id + corner];
                    const auto it = known_vertices.find(key);
                    if (it != known_vertices.end()) {
                        ids[corner] = (uint32_t)it->second;
                        continue;
                    }
                    if (!(key.v >= 0 && key.v < nV) || key.vn >= nVN || key.vt >= nVT) {
                        g_obj_error = obj_file + ": index (" + std::to_string(key.v) + ", " + std::to_string(key.vt) + ", " + std::to_string(key.vn) + ") out of range";
                        return PT_ERR_INVALID;
                    }
                    const int new_id = (int)(mesh.vertex.size() / 3);
                    known_vertices.emplace(key, new_id);
                    mesh.vertex.insert(mesh.vertex.end(), &P.V[(size_t)3 * key.v], &P.V[(size_t)3 * key.v] + 3);
                    const size_t nvert = mesh.vertex.size() / 3;
                    if (key.vn >= 0)
                        while (mesh.normal.size() / 3 < nvert) mesh.normal.insert(mesh.normal.end(), &P.VN[(size_t)3 * key.vn], &P.VN[(size_t)3 * key.vn] + 3);
                    if (key.vt >= 0)
                        while (mesh.texcoord.size() / 2 < nvert) mesh.texcoord.insert(mesh.texcoord.end(), &P.VT[(size_t)2 * key.vt], &P.VT[(size_t)2 * key.vt] + 2);
                    if (!mesh.texcoord.empty()) mesh.texcoord.resize(2 * nvert, 0.f); // "just for sanity's sake": the vertex count, zero fill
                    if (!mesh.normal.empty()) mesh.normal.resize(3 * nvert, 0.f);
                    ids[corner] = (uint32_t)new_id;
                }
                mesh.index.insert(mesh.index.end(), ids, ids + 3);
                if (material_id >= 0) {
                    const Mtl& md = P.materials[(size_t)material_id];
                    std::memcpy(mesh.material.color, md.diffuse, sizeof md.diffuse);
                    std::memcpy(mesh.material.emission, md.emission, sizeof md.emission);
                    if (md.diffuse_texname.empty()) {
                        mesh.tex_ref = -1;
                    } else {
                        const auto kt = known_textures.find(md.diffuse_texname);
                        if (kt != known_textures.end()) {
                            mesh.tex_ref = kt->second;
                        } else {
                            std::string name = md.diffuse_texname;
                            for (char& ch : name)
                                if (ch == '\\') ch = '/';
                            mesh.tex_ref = (int32_t)obj->texture_paths.size();
                            obj->texture_paths.push_back(model_dir + "/" + name); // loadTexture (:92): modelPath + "/" + fileName
                            known_textures.emplace(md.diffuse_texname, mesh.tex_ref);
                        }
                    }
                }
            }
            if (mesh.vertex.empty()) continue;
            obj->meshes.push_back(std::move(mesh));
        }
    }
    *out = obj.release();
    return PT_OK;
}
// No C++ exception crosses the C ABI: a file of several hundred megabytes can run the host out of memory in the middle of a parse
extern "C" int pt_load_obj(const char* obj_path, int per_mesh_vertex_map, pt_obj** out) {
    if (!obj_path || !out) return PT_ERR_INVALID;
    *out = nullptr;
    try {
        return load_obj(obj_path, per_mesh_vertex_map, out);
    } catch (const std::exception& e) {
        try { g_obj_error = std::string(obj_path) + ": " + e.what(); } catch (...) {}
    } catch (...) {
        try { g_obj_error = std::string(obj_path) + ": unknown failure"; } catch (...) {}
    }
    return PT_ERR_INVALID;
}

extern "C" void pt_obj_free(pt_obj* obj) { delete obj; }
extern "C" uint32_t pt_obj_num_meshes(const pt_obj* obj) { return obj ? (uint32_t)obj->meshes.size() : 0u; }
extern "C" uint32_t pt_obj_num_textures(const pt_obj* obj) { return obj ? (uint32_t)obj->texture_paths.size() : 0u; }
extern "C" const char* pt_obj_texture_path(const pt_obj* obj, uint32_t i) { return (obj && i < obj->texture_paths.size()) ? obj->texture_paths[i].c_str() : nullptr; }
extern "C" int pt_obj_get_mesh(const pt_obj* obj, uint32_t i, pt_obj_mesh* out) {
    if (!obj || !out || i >= obj->meshes.size()) return PT_ERR_INVALID;
    const Mesh& m = obj->meshes[i];
    out->vertex = m.vertex.data();
    out->normal = m.normal.empty() ? nullptr : m.normal.data();
    out->texcoord = m.texcoord.empty() ? nullptr : m.texcoord.data();
    out->index = m.index.data();
    out->num_vertices = (uint32_t)(m.vertex.size() / 3);
    out->num_triangles = (uint32_t)(m.index.size() / 3);
    out->material = m.material;
    out->texture_ref = m.tex_ref;
    return PT_OK;
}
