"""loadOBJ (HelloPathtracing_original/Model.cpp:137-212) without tinyobjloader/stb: OBJ + MTL → Model.

Same result shape as the reference: one TriangleMesh per (shape, material) holding de-duplicated vertices
(addVertex, :51-84: one vertex per distinct (position, normal, texcoord) index triple), material.color = Kd,
material.emission = Ke, diffuseTextureID from map_Kd (loadTexture, :88-135: RGBA8, rows mirrored in y because the
image loader returns the top row first), faces triangulated as fans like tinyobj's triangulate=true.
One deliberate difference: the reference keeps ONE knownVertices map per shape across its materials (:176), so a
vertex first seen under another material returns an index into the wrong mesh; here the map is per mesh.
Images are decoded with PIL (any format it reads); .ppm/.png are what the tests use.
"""
from __future__ import annotations

import os

import numpy as np

from .scenes import Material, Model, Texture, TriangleMesh


def _load_mtl(path):
    mats, cur = {}, None
    if not os.path.exists(path):
        return mats
    for line in open(path, errors="ignore"):
        t = line.split()
        if not t or t[0].startswith("#"):
            continue
        if t[0] == "newmtl":
            cur = " ".join(t[1:])
            mats[cur] = dict(Kd=(0.6, 0.6, 0.6), Ke=(0.0, 0.0, 0.0), map_Kd="")
        elif cur is not None and t[0] in ("Kd", "Ke") and len(t) >= 4:
            mats[cur][t[0]] = tuple(float(x) for x in t[1:4])
        elif cur is not None and t[0] == "map_Kd":
            mats[cur]["map_Kd"] = t[-1]
    return mats


def load_texture(model: Model, known: dict, name: str, model_dir: str) -> int:
    """loadTexture (Model.cpp:88-135): returns the texture id or -1."""
    if name == "":
        return -1
    if name in known:
        return known[name]
    fn = os.path.join(model_dir, name.replace("\\", "/"))
    tid = -1
    try:
        from PIL import Image

        img = np.asarray(Image.open(fn).convert("RGBA"), np.uint32)  # top row first, like stbi_load
        px = img[..., 0] | (img[..., 1] << 8) | (img[..., 2] << 16) | (img[..., 3] << 24)
        px = px[::-1].copy()  # "stbi loads the pictures mirrored along the y axis - mirror them here" (:112-121)
        tid = len(model.textures)
        model.textures.append(Texture(np.ascontiguousarray(px, np.uint32)))
    except Exception:
        print(f"Could not load texture from {fn}!")
    known[name] = tid
    return tid


def load_obj(obj_file: str) -> Model:
    model = Model()
    model_dir = os.path.dirname(obj_file)
    V, VN, VT = [], [], []
    mats = {}
    shapes = []  # list of (name, faces) with faces = list of (material, [(v,vt,vn) x3])
    cur_faces, cur_mat = [], None
    for line in open(obj_file, errors="ignore"):
        t = line.split()
        if not t or t[0].startswith("#"):
            continue
        if t[0] == "v":
            V.append([float(x) for x in t[1:4]])
        elif t[0] == "vn":
            VN.append([float(x) for x in t[1:4]])
        elif t[0] == "vt":
            VT.append([float(x) for x in t[1:3]])
        elif t[0] == "mtllib":
            mats.update(_load_mtl(os.path.join(model_dir, " ".join(t[1:]))))
        elif t[0] == "usemtl":
            cur_mat = " ".join(t[1:])
        elif t[0] in ("o", "g"):
            if cur_faces:
                shapes.append(cur_faces)
            cur_faces = []
        elif t[0] == "f":
            idx = []
            for tok in t[1:]:
                p = (tok.split("/") + ["", ""])[:3]

                def fix(s, n):
                    if s == "":
                        return -1
                    k = int(s)
                    return k - 1 if k > 0 else n + k

                idx.append((fix(p[0], len(V)), fix(p[1], len(VT)), fix(p[2], len(VN))))
            for k in range(1, len(idx) - 1):  # fan triangulation
                cur_faces.append((cur_mat, (idx[0], idx[k], idx[k + 1])))
    if cur_faces:
        shapes.append(cur_faces)
    V = np.array(V, np.float32).reshape(-1, 3)
    VN = np.array(VN, np.float32).reshape(-1, 3)
    VT = np.array(VT, np.float32).reshape(-1, 2)
    for faces in shapes:
        known_tex = {}
        mat_ids = []
        for mname, _ in faces:
            if mname not in mat_ids:
                mat_ids.append(mname)
        for mname in sorted(mat_ids, key=lambda x: (x is None, str(x))):  # std::set<int> order ~ material order
            known, verts, norms, tcs, tris = {}, [], [], [], []
            for fm, tri in faces:
                if fm != mname:
                    continue
                ids = []
                for key in tri:
                    if key not in known:
                        known[key] = len(verts)
                        verts.append(V[key[0]])
                        if key[2] >= 0:
                            norms.append(VN[key[2]])
                        if key[1] >= 0:
                            tcs.append(VT[key[1]])
                    ids.append(known[key])
                tris.append(ids)
            if not verts:
                continue
            md = mats.get(mname, dict(Kd=(0.6, 0.6, 0.6), Ke=(0.0, 0.0, 0.0), map_Kd=""))
            mesh = TriangleMesh(np.array(verts, np.float32), np.array(tris, np.uint32), Material(color=md["Kd"], emission=md["Ke"]))
            if len(tcs) == len(verts):
                mesh.texcoord = np.array(tcs, np.float32)
            mesh.diffuseTextureID = load_texture(model, known_tex, md["map_Kd"], model_dir)
            model.meshes.append(mesh)
    return model
