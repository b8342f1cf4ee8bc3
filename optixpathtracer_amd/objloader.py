"""loadOBJ (HelloPathtracing_original/Model.cpp:137-212): OBJ + MTL + textures -> Model, with the reference's result bit for bit.

The reference parses with tinyobjloader 2.0.0 (vendored: support/tinyobjloader/tiny_obj_loader.h, `LoadObj(..., triangulate=true)`)
and decodes images with stb_image.  Neither travels with this package, so the parts of tinyobjloader that decide what loadOBJ sees
are restated here (line numbers = tiny_obj_loader.h):

  * numbers: `tryParseDouble` (:836-960) — digit-by-digit accumulation in double, `mantissa += digit * pow_lut[k]`, exponent through
    `ldexp(mantissa * pow(5, e), e)` — then the cast to float; a missing or malformed field takes the default (0; :962-970);
  * lines end at \\n, \\r\\n or a lone \\r (`safeGetline`, :731-764); fields are separated by blanks and tabs;
  * `f` (:2407-2447): `i`, `i/j`, `i//k`, `i/j/k`, 1-based or negative = relative to the elements read so far (`fixIndex`, :770-790);
    index 0 fails the whole load;
  * `usemtl` (:2450-2477) takes the first blank-delimited word; a CHANGE of material id flushes the pending faces into the current
    shape; `g` / `o` (:2528-2601) flush and start a new shape (kept only if it has faces); the end of file flushes;
  * `mtllib` (:2480-2525): file names split at single blanks, the first one that opens (relative to the OBJ's directory) is read;
  * LoadMtl (:1688-2077): `newmtl` takes the rest of the line; `Kd` / `Ke` three reals; `map_Kd` options (:1186-1265), then the rest
    of the line is the file name; a material without `Kd` has diffuse 0 — or 0.6 if `map_Kd` comes before ANY `Kd` of the file
    (`has_kd` is never reset, :1703,1786,1942); the last material is pushed even without a name (:2070-2072); of two materials with
    one name the first keeps the name (std::map::insert);
  * polygons are triangulated by ear clipping in float arithmetic on the two axes picked from the first non-degenerate corner
    (`exportGroupsToShape`, :1365-1580) — a fan for convex polygons, something else for concave ones.

and Model.cpp itself, quirks included (they decide the arrays a drop-in must hand to the renderer):

  * one TriangleMesh per (shape, material id) in ascending id order (std::set<int>, :170-172), dropped if it ended up without vertices;
  * `addVertex` (:51-84): one vertex per distinct (position, normal, texcoord) index triple; normals / texcoords are back-filled with
    the current vertex's value when earlier vertices had none, zero-padded when later ones have none;
  * a triangle's corners are added LAST corner first (the three addVertex calls are arguments of one make_uint3 call, :186-188, and
    both g++ and MSVC evaluate arguments right to left), so vertex numbers run 2, 1, 0 within a fresh triangle;
  * ONE `knownVertices` map per SHAPE, shared by its materials (:176): an index triple first seen under another material of the
    shape returns that OTHER mesh's vertex number (and adds no vertex).  Kept, because the reference's arrays are the contract;
    `per_mesh_vertex_map=True` gives every mesh its own map instead;
  * `knownTextures` is per shape too (:177): a file used by two shapes is loaded twice and gets two texture ids;
  * `loadTexture` (:88-135): '\\\\' -> '/', RGBA8, rows mirrored in y, -1 (and a message) when the file cannot be read.

Deliberate differences: a face without material (`materials[-1]`, an out-of-bounds read in the reference) gets the default Material;
out-of-range v/vn/vt indices raise instead of reading past the arrays.  Images are decoded with PIL: pinned bit for bit to the
reference's stb_image for PNG, BMP and TGA (true colour, RLE with alpha, grey: tests/test_objloader.py); JPEG differs in the last bits of
some texels (8 % of the channel values of the committed fixture, by at most 2 of 255), as it does between any two decoders.

Three entry points: `load_model(path)` — what a renderer wants: the native parser (pt_load_obj, include/pt_amd.h) with a vertex map per
mesh; `load_obj(path)` — loadOBJ's arrays bit for bit, shared vertex map and all (native by default, `native=False` for the Python
restatement below, from which csrc/pt_objload.cpp was written).
"""
from __future__ import annotations

import math
import os
import re

import numpy as np

from .scenes import MATERIAL_DTYPE, Material, Model, Texture, TriangleMesh

_F = np.float32
_POW_LUT = (1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001)


def _try_parse_double(s: str):
    """tryParseDouble (tiny_obj_loader.h:836-960): returns the parsed double or None."""
    n = len(s)
    if n == 0:
        return None
    i = 0
    mant = 0.0
    sign = 1.0
    leading_dot = False
    c = s[0]
    if c in "+-":
        sign = -1.0 if c == "-" else 1.0
        i = 1
        if i != n and s[i] == ".":
            leading_dot = True
    elif c.isascii() and c.isdigit():
        pass
    elif c == ".":
        leading_dot = True
    else:
        return None
    read = 0
    if not leading_dot:
        while i != n and s[i].isascii() and s[i].isdigit():
            mant = mant * 10.0 + float(ord(s[i]) - 48)
            i += 1
            read += 1
        if read == 0:
            return None
    exponent = 0
    if i != n:
        if s[i] == ".":
            i += 1
            read = 1
            while i != n and s[i].isascii() and s[i].isdigit():
                mant += float(ord(s[i]) - 48) * (_POW_LUT[read] if read < 8 else math.pow(10.0, -read))
                read += 1
                i += 1
        elif s[i] in "eE":
            pass
        else:
            return sign * mant
        if i != n and s[i] in "eE":
            i += 1
            esign = 1
            if i != n and s[i] in "+-":
                esign = 1 if s[i] == "+" else -1
                i += 1
            elif i != n and s[i].isascii() and s[i].isdigit():
                pass
            else:
                return None
            read = 0
            while i != n and s[i].isascii() and s[i].isdigit():
                exponent = exponent * 10 + (ord(s[i]) - 48)
                i += 1
                read += 1
            exponent *= esign
            if read == 0:
                return None
    if exponent:
        # C semantics where Python raises: pow overflows to inf (so 0e999 is 0 * inf = NaN, as in the reference), ldexp to +-inf
        exponent = max(-1000000, min(1000000, exponent))  # (beyond, the value is +-inf, 0 or NaN whatever the digits)
        try:
            p5 = math.pow(5.0, exponent)
        except OverflowError:
            p5 = math.inf
        v = mant * p5
        try:
            return sign * math.ldexp(v, exponent)
        except OverflowError:
            return sign * math.copysign(math.inf, v)
    return sign * mant


_TOKENS = re.compile(r"[^ \t]+").findall  # fields as parseReal / parseTriple delimit them (blank or tab; a line holds no \r or \n)
_REAL_CACHE: dict = {}


def _reals(text: str, n: int, default=0.0):
    """n consecutive parseReal calls (tiny_obj_loader.h:962-970) on `text`: a missing or malformed field takes the default."""
    toks = _TOKENS(text)
    out = []
    for k in range(n):
        if k < len(toks):
            t = toks[k]
            v = _REAL_CACHE.get(t)
            if v is None:
                d = _try_parse_double(t)
                with np.errstate(over="ignore"):
                    v = _F(default if d is None else d)
                if d is not None and len(_REAL_CACHE) < 1_000_000:  # grid-aligned scenes repeat a few thousand spellings; the memo is dropped after the file
                    _REAL_CACHE[t] = v
            out.append(v)
        else:
            out.append(_F(default))
    return out


def _fix_group(g, n):
    """fixIndex (:770-790) on a matched index field: -1 = field absent, None = the reference's `return false` (index 0)."""
    if g is None:
        return -1
    k = int(g)
    if k > 0:
        return k - 1
    return None if k == 0 else n + k


_FACE_TOKEN = re.compile(r"([+-]?\d+)(?:/([+-]?\d+)?(?:/([+-]?\d+))?)?").fullmatch


class _Cursor:
    """The `const char** token` the parser walks along a line."""

    def __init__(self, s: str, pos: int = 0):
        self.s, self.p = s, pos

    def skip_blank(self):
        s, p = self.s, self.p
        while p < len(s) and s[p] in " \t":
            p += 1
        self.p = p

    def word(self, stops=" \t\r"):
        s, p = self.s, self.p
        e = p
        while e < len(s) and s[e] not in stops:
            e += 1
        self.p = e
        return s[p:e]

    def real(self, default=0.0) -> np.float32:  # parseReal (:962-970)
        self.skip_blank()
        v = _try_parse_double(self.word())
        with np.errstate(over="ignore"):
            return _F(default if v is None else v)

    def string(self) -> str:  # parseString (:793-800)
        self.skip_blank()
        return self.word()

    def at_end(self):
        return self.p >= len(self.s) or self.s[self.p] in "\r\n\0"

    def rest(self):
        return self.s[self.p:]


def _atoi(s: str, p: int) -> int:
    m = re.match(r"[ \t\n\v\f\r]*([+-]?\d+)", s[p:])
    return int(m.group(1)) if m else 0


def _fix_index(idx: int, n: int):
    if idx > 0:
        return idx - 1
    if idx == 0:
        return None
    return n + idx


def _parse_triple(cur: _Cursor, nv, nvn, nvt):
    """parseTriple (:1100-1148) -> (v, vt, vn) zero-based, -1 = absent; None = the reference's `return false`."""
    s = cur.s
    v = _fix_index(_atoi(s, cur.p), nv)
    if v is None:
        return None
    vt = vn = -1
    cur.word("/ \t\r")
    if cur.p >= len(s) or s[cur.p] != "/":
        return (v, vt, vn)
    cur.p += 1
    if cur.p < len(s) and s[cur.p] == "/":  # i//k
        cur.p += 1
        vn = _fix_index(_atoi(s, cur.p), nvn)
        if vn is None:
            return None
        cur.word("/ \t\r")
        return (v, vt, vn)
    vt = _fix_index(_atoi(s, cur.p), nvt)  # i/j or i/j/k
    if vt is None:
        return None
    cur.word("/ \t\r")
    if cur.p >= len(s) or s[cur.p] != "/":
        return (v, vt, vn)
    cur.p += 1
    vn = _fix_index(_atoi(s, cur.p), nvn)
    if vn is None:
        return None
    cur.word("/ \t\r")
    return (v, vt, vn)


def _lines(path):
    """safeGetline (:731-764) over the whole file; bytes map 1:1 to characters."""
    with open(path, "rb") as f:
        text = f.read().decode("latin-1")
    out = re.split(r"\r\n|\n|\r", text)
    if out and out[-1] == "":
        out.pop()
    return out


_TEXOPT_ARGS = {"-blendu": 1, "-blendv": 1, "-clamp": 1, "-boost": 1, "-bm": 1, "-o": 3, "-s": 3, "-t": 3, "-texres": 1, "-imfchan": 1, "-mm": 2, "-colorspace": 1}


def _texture_name(rest: str) -> str:
    """ParseTextureNameAndOption (:1186-1265): options with their arguments, then everything up to the end of the line is the name."""
    cur = _Cursor(rest)
    name = ""
    while not cur.at_end():
        cur.skip_blank()
        s, p = cur.s, cur.p
        hit = None
        for opt, nargs in _TEXOPT_ARGS.items():
            if s.startswith(opt, p) and p + len(opt) < len(s) and s[p + len(opt)] in " \t":
                hit = (opt, nargs)
                break
        if hit is None and s.startswith("-type", p) and p + 5 < len(s) and s[p + 5] in " \t":
            cur.p = p + 5
            cur.skip_blank()
            cur.word()
            continue
        if hit is None:
            name = s[p:]
            cur.p = len(s)
            break
        cur.p = p + len(hit[0]) + (0 if hit[0] == "-texres" else 1)
        for _ in range(hit[1]):
            cur.skip_blank()
            cur.word()
    return name


def _new_material():
    return {"name": "", "diffuse": np.zeros(3, _F), "emission": np.zeros(3, _F), "diffuse_texname": ""}


def _load_mtl(path, materials: list, material_map: dict):
    """LoadMtl (:1688-2077), the fields loadOBJ reads: diffuse, emission, diffuse_texname."""
    mat = _new_material()
    has_kd = False
    for line in _lines(path):
        line = line[: len(line.rstrip(" \t"))]
        if not line:
            continue
        p = 0
        while p < len(line) and line[p] in " \t":
            p += 1
        tok = line[p:]
        if tok == "" or tok[0] == "#":
            continue
        if tok.startswith("newmtl") and len(tok) > 6 and tok[6] in " \t":
            if mat["name"] != "":
                material_map.setdefault(mat["name"], len(materials))
                materials.append(mat)
            mat = _new_material()
            mat["name"] = tok[7:]
            continue
        if len(tok) > 2 and tok[0] == "K" and tok[1] in "de" and tok[2] in " \t":
            cur = _Cursor(tok, 2)
            rgb = np.array([cur.real(), cur.real(), cur.real()], _F)
            if tok[1] == "d":
                mat["diffuse"] = rgb
                has_kd = True
            else:
                mat["emission"] = rgb
            continue
        if tok.startswith("map_Kd") and len(tok) > 6 and tok[6] in " \t":
            mat["diffuse_texname"] = _texture_name(tok[7:])
            if not has_kd:
                mat["diffuse"] = np.full(3, 0.6, _F)
            continue
    material_map.setdefault(mat["name"], len(materials))
    materials.append(mat)


def _triangulate(face, V):
    """exportGroupsToShape's ear clipping (:1382-1580) for one polygon: list of index triples -> list of triangles.  All arithmetic
    in float, one operation at a time, like the reference's real_t expressions."""
    npolys = len(face)
    nfl = 3 * len(V)
    axes = [1, 2]
    eps = _F(np.finfo(np.float32).eps)
    with np.errstate(all="ignore"):
        for k in range(npolys):
            vi0, vi1, vi2 = face[k % npolys][0], face[(k + 1) % npolys][0], face[(k + 2) % npolys][0]
            if 3 * vi0 + 2 >= nfl or 3 * vi1 + 2 >= nfl or 3 * vi2 + 2 >= nfl or min(vi0, vi1, vi2) < 0:
                continue
            v0, v1, v2 = V[vi0], V[vi1], V[vi2]
            e0x, e0y, e0z = v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]
            e1x, e1y, e1z = v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]
            cx = abs(e0y * e1z - e0z * e1y)
            cy = abs(e0z * e1x - e0x * e1z)
            cz = abs(e0x * e1y - e0y * e1x)
            if cx > eps or cy > eps or cz > eps:
                if cx > cy and cx > cz:
                    pass
                else:
                    axes[0] = 0
                    if cz > cx and cz > cy:
                        axes[1] = 1
                break
        a0, a1 = axes
        area = _F(0)
        half = _F(0.5)
        for k in range(npolys):
            vi0, vi1 = face[k % npolys][0], face[(k + 1) % npolys][0]
            if not (0 <= vi0 < len(V) and 0 <= vi1 < len(V)):
                continue
            area = area + (V[vi0][a0] * V[vi1][a1] - V[vi0][a1] * V[vi1][a0]) * half
        rem = list(face)
        out = []
        guess = 0
        remaining_iter = len(face)
        prev_remaining = len(rem)
        zero = _F(0)
        while len(rem) > 3 and remaining_iter > 0:
            npolys = len(rem)
            if guess >= npolys:
                guess -= npolys
            if prev_remaining != npolys:
                prev_remaining = npolys
                remaining_iter = npolys
            else:
                remaining_iter -= 1
            ind = [rem[(guess + k) % npolys] for k in range(3)]
            vx, vy = [], []
            for k in range(3):
                vi = ind[k][0]
                if not (0 <= vi < len(V)):
                    vx.append(zero); vy.append(zero)
                else:
                    vx.append(V[vi][a0]); vy.append(V[vi][a1])
            e0x, e0y = vx[1] - vx[0], vy[1] - vy[0]
            e1x, e1y = vx[2] - vx[1], vy[2] - vy[1]
            cross = e0x * e1y - e0y * e1x
            if cross * area < zero:
                guess += 1
                continue
            overlap = False
            for other in range(3, npolys):
                idx = (guess + other) % npolys
                ovi = rem[idx][0]
                if not (0 <= ovi < len(V)):
                    continue
                tx, ty = V[ovi][a0], V[ovi][a1]
                c = False  # pnpoly (:1353-1363)
                j = 2
                for i in range(3):
                    if (vy[i] > ty) != (vy[j] > ty) and tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i]:
                        c = not c
                    j = i
                if c:
                    overlap = True
                    break
            if overlap:
                guess += 1
                continue
            out.append((ind[0], ind[1], ind[2]))
            del rem[(guess + 1) % npolys]
        if len(rem) == 3:
            out.append((rem[0], rem[1], rem[2]))
    return out


def _quads_clip_to_fan(Q: np.ndarray, V: np.ndarray) -> np.ndarray:
    """For many 4-gons at once: does the ear clipping of _triangulate take its FIRST candidate ear (corners 0,1,2), so that the result is
    the fan (0,1,2),(0,2,3)?  The same float operations in the same order as the scalar code, on arrays.  False = run the scalar code."""
    nv = len(V)
    ok = ((Q >= 0) & (Q < nv)).all(axis=1)
    Qs = np.where(ok[:, None], Q, 0)
    P = V[Qs]  # (m, 4, 3)
    m = len(Q)
    eps = _F(np.finfo(np.float32).eps)
    a0 = np.full(m, 1, np.int64)
    a1 = np.full(m, 2, np.int64)
    found = np.zeros(m, bool)
    with np.errstate(all="ignore"):
        for k in range(4):
            v0, v1, v2 = P[:, k], P[:, (k + 1) % 4], P[:, (k + 2) % 4]
            e0, e1 = v1 - v0, v2 - v1
            cx = np.abs(e0[:, 1] * e1[:, 2] - e0[:, 2] * e1[:, 1])
            cy = np.abs(e0[:, 2] * e1[:, 0] - e0[:, 0] * e1[:, 2])
            cz = np.abs(e0[:, 0] * e1[:, 1] - e0[:, 1] * e1[:, 0])
            corner = ((cx > eps) | (cy > eps) | (cz > eps)) & ~found
            keep = (cx > cy) & (cx > cz)
            a0 = np.where(corner & ~keep, 0, a0)
            a1 = np.where(corner & ~keep & (cz > cx) & (cz > cy), 1, a1)
            found |= corner
        rows = np.arange(m)
        X = np.stack([P[rows, k, a0] for k in range(4)], 1)  # (m, 4) first working axis
        Y = np.stack([P[rows, k, a1] for k in range(4)], 1)
        half = _F(0.5)
        area = np.zeros(m, np.float32)
        for k in range(4):
            j = (k + 1) % 4
            area = area + (X[:, k] * Y[:, j] - Y[:, k] * X[:, j]) * half
        e0x, e0y = X[:, 1] - X[:, 0], Y[:, 1] - Y[:, 0]
        e1x, e1y = X[:, 2] - X[:, 1], Y[:, 2] - Y[:, 1]
        cross = e0x * e1y - e0y * e1x
        ok &= ~(cross * area < _F(0))
        tx, ty = X[:, 3], Y[:, 3]
        inside = np.zeros(m, bool)
        j = 2
        for i in range(3):
            c1 = (Y[:, i] > ty) != (Y[:, j] > ty)
            c2 = tx < (X[:, j] - X[:, i]) * (ty - Y[:, i]) / (Y[:, j] - Y[:, i]) + X[:, i]
            inside ^= c1 & c2
            j = i
        ok &= ~inside
    return ok


def _parse_obj(obj_file: str, mtl_basedir: str):
    """tinyobj::LoadObj(..., triangulate=true) (:2158-2745) -> V, VN, VT, shapes, materials.
    A shape = (indices [(v, vt, vn) x 3 per triangle], material id per triangle)."""
    V, VN, VT = [], [], []
    materials, material_map = [], {}
    shapes = []
    shape = ([], [])
    faces = []  # prim_group.faceGroup
    material = -1
    base = mtl_basedir
    if base and not base.endswith("/"):
        base += "/"

    def Varr():
        return np.array(V, _F).reshape(-1, 3)

    def export():
        if not faces:
            return False
        Vf = Varr()
        quads = [i for i, face in enumerate(faces) if len(face) == 4]
        fan = {}
        if len(quads) >= 32:  # many 4-gons: decide for all of them at once which ones the ear clipping turns into the plain fan
            okq = _quads_clip_to_fan(np.array([[c[0] for c in faces[i]] for i in quads], np.int64), Vf)
            fan = dict(zip(quads, okq.tolist()))
        for i, face in enumerate(faces):
            if len(face) < 3:
                continue
            if len(face) == 3:
                tris = (tuple(face),)
            elif fan.get(i, False):
                tris = ((face[0], face[1], face[2]), (face[0], face[2], face[3]))
            else:
                tris = _triangulate(face, Vf)
            for tri in tris:
                shape[0].extend(tri)
                shape[1].append(material)
        return True

    for ln, line in enumerate(_lines(obj_file), 1):
        if not line:
            continue
        p = 0
        while p < len(line) and line[p] in " \t":
            p += 1
        tok = line[p:]
        if tok == "" or tok[0] == "#":
            continue
        c1 = tok[1] if len(tok) > 1 else "\0"
        c2 = tok[2] if len(tok) > 2 else "\0"
        if tok[0] == "v" and c1 in " \t":
            V.append(_reals(tok[2:], 3))
            continue
        if tok[0] == "v" and c1 == "n" and c2 in " \t":
            VN.append(_reals(tok[3:], 3))
            continue
        if tok[0] == "v" and c1 == "t" and c2 in " \t":
            VT.append(_reals(tok[3:], 2))
            continue
        if tok[0] == "f" and c1 in " \t":
            face = []
            nv, nvn, nvt = len(V), len(VN), len(VT)
            for ft in _TOKENS(tok[2:]):  # the common spellings i, i/j, i//k, i/j/k without going through the cursor
                mt = _FACE_TOKEN(ft)
                if mt is None or ft[-1] == "/":
                    face = None
                    break
                a, b, c = mt.groups()
                a, b, c = _fix_group(a, nv), _fix_group(b, nvt), _fix_group(c, nvn)
                if a is None or b is None or c is None:
                    face = None
                    break
                face.append((a, b, c))
            if face is None:  # anything else: the reference's parser, character by character
                cur = _Cursor(tok, 2)
                cur.skip_blank()
                face = []
                while not cur.at_end():
                    t = _parse_triple(cur, len(V), len(VN), len(VT))
                    if t is None:
                        raise RuntimeError(f"Could not read OBJ model from {obj_file} : Failed parse `f' line(e.g. zero value for face index. line {ln}.)")
                    face.append(t)
                    while cur.p < len(cur.s) and cur.s[cur.p] in " \t\r":
                        cur.p += 1
            faces.append(face)
            continue
        if tok.startswith("usemtl"):
            name = _Cursor(tok, 6).string()
            new_id = material_map.get(name, -1)
            if new_id != material:
                export()
                faces = []
                material = new_id
            continue
        if tok.startswith("mtllib") and len(tok) > 6 and tok[6] in " \t":
            for fn in [x for x in tok[7:].split(" ")] if tok[7:] != "" else []:
                path = fn if not base else base + fn
                if os.path.isfile(path):
                    _load_mtl(path, materials, material_map)
                    break
            continue
        if tok[0] in "go" and c1 in " \t":
            export()
            if shape[0]:
                shapes.append(shape)
            shape = ([], [])
            faces = []
            continue
        # l / p / t / s / vw and unknown commands do not change what loadOBJ reads
    ret = export()
    if ret or shape[0]:
        shapes.append(shape)
    _REAL_CACHE.clear()
    return Varr(), np.array(VN, _F).reshape(-1, 3), np.array(VT, _F).reshape(-1, 2), shapes, materials


def load_texture(model: Model, known: dict, name: str, model_dir: str) -> int:
    """loadTexture (Model.cpp:88-135): returns the texture id or -1."""
    if name == "":
        return -1
    if name in known:
        return known[name]
    fn = model_dir + "/" + name.replace("\\", "/")
    tid = -1
    try:
        from PIL import Image

        with Image.open(fn) as im:
            img = np.asarray(im.convert("RGBA"), np.uint32)  # top row first, like stbi_load(..., STBI_rgb_alpha)
        px = img[..., 0] | (img[..., 1] << 8) | (img[..., 2] << 16) | (img[..., 3] << 24)
        px = px[::-1].copy()  # "stbi loads the pictures mirrored along the y axis - mirror them here" (:112-121)
        tid = len(model.textures)
        model.textures.append(Texture(np.ascontiguousarray(px, np.uint32)))
    except Exception:
        print(f"Could not load texture from {fn}!")
    known[name] = tid
    return tid


def _decode_texture(fn: str):
    """what loadTexture (Model.cpp:88-135) keeps of an image file: RGBA8 texels, rows mirrored in y; None when the file cannot be read."""
    try:
        from PIL import Image

        with Image.open(fn) as im:
            img = np.asarray(im.convert("RGBA"), np.uint32)  # top row first, like stbi_load(..., STBI_rgb_alpha)
        px = img[..., 0] | (img[..., 1] << 8) | (img[..., 2] << 16) | (img[..., 3] << 24)
        return np.ascontiguousarray(px[::-1], np.uint32)
    except Exception:
        print(f"Could not load texture from {fn}!")
        return None


def _load_obj_native(obj_file: str, per_mesh_vertex_map: bool) -> Model:
    """pt_load_obj (include/pt_amd.h, csrc/pt_objload.cpp): the same arrays from native code — 150 MB of OBJ in seconds instead of a minute."""
    import ctypes as C

    from . import _lib

    if os.environ.get("PT_OBJ_LIB"):  # a build of csrc/pt_objload.cpp alone (tools/sanitize.sh: the parser under AddressSanitizer / UBSan)
        L = C.CDLL(os.environ["PT_OBJ_LIB"])
        vp, u32 = C.c_void_p, C.c_uint32
        L.pt_load_obj.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
        L.pt_obj_free.argtypes = [vp]
        L.pt_obj_free.restype = None
        L.pt_obj_num_meshes.argtypes = [vp]
        L.pt_obj_num_meshes.restype = u32
        L.pt_obj_get_mesh.argtypes = [vp, u32, C.POINTER(_lib.ObjMesh)]
        L.pt_obj_num_textures.argtypes = [vp]
        L.pt_obj_num_textures.restype = u32
        L.pt_obj_texture_path.argtypes = [vp, u32]
        L.pt_obj_texture_path.restype = C.c_char_p
        L.pt_obj_last_error.restype = C.c_char_p
    else:
        L = _lib.load_library()
    h = C.c_void_p()
    # (the path goes through as the file system's bytes, and comes back through the same codec: os.fsencode / os.fsdecode round-trip any name)
    if L.pt_load_obj(os.fsencode(obj_file), 1 if per_mesh_vertex_map else 0, C.byref(h)) != 0:
        msg = os.fsdecode(L.pt_obj_last_error())
        raise (ValueError if "out of range" in msg else RuntimeError)(msg)
    try:
        model = Model()
        final_id = {}
        for k in range(L.pt_obj_num_textures(h)):  # loadTexture's order; unreadable files get -1 and no number
            px = _decode_texture(os.fsdecode(L.pt_obj_texture_path(h, k)))
            if px is None:
                final_id[k] = -1
            else:
                final_id[k] = len(model.textures)
                model.textures.append(Texture(px))
        for i in range(L.pt_obj_num_meshes(h)):
            m = _lib.ObjMesh()
            if L.pt_obj_get_mesh(h, i, C.byref(m)) != 0:
                raise RuntimeError("pt_obj_get_mesh failed")
            nv, nt = m.num_vertices, m.num_triangles
            mat = np.frombuffer(bytes(m.material), MATERIAL_DTYPE)[0].copy()  # pt_material == Material, field for field (scenes.MATERIAL_DTYPE)
            mesh = TriangleMesh(np.ctypeslib.as_array(m.vertex, (nv, 3)).copy(), np.ctypeslib.as_array(m.index, (nt, 3)).copy(), mat)
            mesh.normal = np.ctypeslib.as_array(m.normal, (nv, 3)).copy() if m.normal else None
            mesh.texcoord = np.ctypeslib.as_array(m.texcoord, (nv, 2)).copy() if m.texcoord else None
            mesh.diffuseTextureID = final_id.get(m.texture_ref, -1)
            model.meshes.append(mesh)
        return model
    finally:
        L.pt_obj_free(h)


def load_model(obj_file: str) -> Model:
    """The product's route for RENDERING an OBJ scene: native parser, every mesh with its own vertex map.  (loadOBJ as the reference
    wrote it shares one vertex map among the materials of a shape, Model.cpp:176, so a second material's mesh indexes vertices it
    does not own — out of bounds in the reference's renderer too; `load_obj` reproduces that for the parity tests.)"""
    return load_obj(obj_file, per_mesh_vertex_map=True, native=True)


def load_obj(obj_file: str, per_mesh_vertex_map: bool = False, native: bool = True) -> Model:
    """loadOBJ (Model.cpp:137-212), the reference's arrays bit for bit.  `per_mesh_vertex_map` undoes the shared knownVertices map
    (module docstring); `native=False` runs the line-cited Python restatement below instead of pt_load_obj (the two are held equal by
    tests/test_objloader.py, and both to the reference's own Model.cpp)."""
    if not os.path.isfile(obj_file):
        raise RuntimeError(f"Could not read OBJ model from {obj_file} : Cannot open file [{obj_file}]")
    if native:
        return _load_obj_native(obj_file, per_mesh_vertex_map)
    model = Model()
    model_dir = obj_file[: obj_file.rfind("/") + 1]
    V, VN, VT, shapes, materials = _parse_obj(obj_file, model_dir)
    for indices, mat_ids in shapes:
        known_vertices = {}
        known_textures = {}
        for material_id in sorted(set(mat_ids)):
            if per_mesh_vertex_map:
                known_vertices = {}
            vertex, normal, texcoord, index = [], [], [], []
            mesh_material = Material()
            tex_id = -1
            for face_id, fm in enumerate(mat_ids):
                if fm != material_id:
                    continue
                ids = []
                # the three addVertex calls are ARGUMENTS of one make_uint3 call (:186-188): their order of evaluation is the
                # compiler's, and both g++ (the pinned build) and MSVC (the reference's platform) go right to left — corner 2 first
                for key in reversed(indices[3 * face_id: 3 * face_id + 3]):  # addVertex (:51-84)
                    if key in known_vertices:
                        ids.append(known_vertices[key])
                        continue
                    vi, vti, vni = key
                    if not (0 <= vi < len(V)) or vni >= len(VN) or vti >= len(VT):
                        raise ValueError(f"{obj_file}: index {key} out of range")
                    new_id = len(vertex)
                    known_vertices[key] = new_id
                    vertex.append(V[vi])
                    if vni >= 0:
                        while len(normal) < len(vertex):
                            normal.append(VN[vni])
                    if vti >= 0:
                        while len(texcoord) < len(vertex):
                            texcoord.append(VT[vti])
                    if texcoord:  # "just for sanity's sake": resize to the vertex count (zero fill)
                        del texcoord[len(vertex):]
                        while len(texcoord) < len(vertex):
                            texcoord.append(np.zeros(2, _F))
                    if normal:
                        del normal[len(vertex):]
                        while len(normal) < len(vertex):
                            normal.append(np.zeros(3, _F))
                    ids.append(new_id)
                index.append(ids[::-1])
                if material_id >= 0:
                    md = materials[material_id]
                    mesh_material["color"] = md["diffuse"]
                    mesh_material["emission"] = md["emission"]
                    tex_id = load_texture(model, known_textures, md["diffuse_texname"], model_dir)
            if not vertex:
                continue
            mesh = TriangleMesh(np.array(vertex, _F).reshape(-1, 3), np.array(index, np.uint32).reshape(-1, 3), mesh_material)
            mesh.normal = np.array(normal, _F).reshape(-1, 3) if normal else None
            mesh.texcoord = np.array(texcoord, _F).reshape(-1, 2) if texcoord else None
            mesh.diffuseTextureID = tex_id
            model.meshes.append(mesh)
    return model
