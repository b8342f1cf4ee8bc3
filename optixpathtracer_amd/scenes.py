"""Procedural inputs for the hot path: scenes, probes, cameras.

The reference ships no assets (all its scenes are absolute C:/ paths,
HelloPathtracing_original/main.cpp:171-175), so the synthetic inputs BASELINE.md §3 /
SURVEY.md §8(d) specify are generated here, deterministically, as plain numpy arrays
in the layout the C-ABI takes (include/pt_amd.h):

  * Model: list of TriangleMesh {vertex (nv,3) f32, index (nt,3) u32, material} — the
    reference's Model/TriangleMesh (Model.h:10-42), one mesh per material like loadOBJ.
  * Material: numpy structured dtype with the reference's field order (Material.h:47-68).
  * ProbeData: width, height, data (h,w,4) f32 + BuildCDF arrays (Probe.h:29-77).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# Material.h:47-68 — 26 x 4 bytes = 104 bytes, same order
MATERIAL_DTYPE = np.dtype(
    [
        ("emission", "<f4", 3),
        ("color", "<f4", 3),
        ("absorption", "<f4", 3),
        ("eta", "<f4"),
        ("metallic", "<f4"),
        ("subsurface", "<f4"),
        ("specular", "<f4"),
        ("roughness", "<f4"),
        ("specularTint", "<f4"),
        ("anisotropic", "<f4"),
        ("sheen", "<f4"),
        ("sheenTint", "<f4"),
        ("clearcoat", "<f4"),
        ("clearcoatGloss", "<f4"),
        ("transmission", "<f4"),
        ("bump", "<f4"),
        ("bumpTile", "<f4", 3),
        ("flags", "<i4"),
    ]
)
assert MATERIAL_DTYPE.itemsize == 104

MATERIAL_FLAG_NONE = 0
MATERIAL_FLAG_SHADOW_CATCHER = 1  # Material.h:9


def Material(**kw) -> np.ndarray:
    """Material() with the reference's constructor defaults (Material.h:13-37)."""
    m = np.zeros((), dtype=MATERIAL_DTYPE)
    m["color"] = (0.6, 0.6, 0.6)
    m["specular"] = 0.5
    m["roughness"] = 1.0
    m["clearcoatGloss"] = 1.0
    m["bumpTile"] = (10.0, 10.0, 10.0)
    for k, v in kw.items():
        m[k] = v
    return m


@dataclass
class TriangleMesh:  # Model.h:10-19
    vertex: np.ndarray  # (nv,3) f32
    index: np.ndarray  # (nt,3) u32, local to this mesh
    material: np.ndarray  # MATERIAL_DTYPE scalar
    diffuseTextureID: int = -1
    texcoord: np.ndarray = None  # (nv,2) f32 or None
    normal: np.ndarray = None  # (nv,3) f32 or None — uploaded but never read by the reference's device code (deviceProgram.cu:488-490 shades flat)


@dataclass
class Texture:  # Model.h:21-29: RGBA8 pixels (h,w) uint32, row 0 first (after loadTexture's y mirror)
    pixel: np.ndarray

    @property
    def resolution(self):
        return (self.pixel.shape[1], self.pixel.shape[0])


@dataclass
class Model:  # Model.h:31-42
    meshes: list = field(default_factory=list)
    textures: list = field(default_factory=list)

    def flatten_textures(self):
        """Global texcoords (nv,2) or None, per-mesh texture id, per-mesh has-texcoord flag."""
        any_uv = any(m.texcoord is not None and len(m.texcoord) for m in self.meshes)
        tcs = []
        for m in self.meshes:
            if m.texcoord is not None and len(m.texcoord):
                tcs.append(np.ascontiguousarray(m.texcoord, np.float32))
            else:
                tcs.append(np.zeros((len(m.vertex), 2), np.float32))
        tc = np.ascontiguousarray(np.concatenate(tcs)) if any_uv else None
        mesh_tex = np.array([m.diffuseTextureID for m in self.meshes], np.int32)
        has_uv = np.array([1 if (m.texcoord is not None and len(m.texcoord)) else 0 for m in self.meshes], np.uint8)
        return tc, mesh_tex, has_uv

    @property
    def num_triangles(self) -> int:
        return int(sum(len(m.index) for m in self.meshes))

    def flatten(self):
        """Global arrays the C-ABI takes: verts (nv,3), idx (nt,3) global, tri_mesh (nt,), mats (nmesh,)."""
        verts, idx, tri_mesh, base = [], [], [], 0
        for mi, m in enumerate(self.meshes):
            if len(m.index) and int(np.max(m.index)) >= len(m.vertex):  # pt_create refuses such a scene too (the reference would read past the mesh's vertex buffer)
                raise ValueError(f"mesh {mi}: vertex index {int(np.max(m.index))} out of range ({len(m.vertex)} vertices)")
            verts.append(np.ascontiguousarray(m.vertex, dtype=np.float32))
            idx.append(np.ascontiguousarray(m.index, dtype=np.uint32) + np.uint32(base))
            tri_mesh.append(np.full(len(m.index), mi, dtype=np.uint32))
            base += len(m.vertex)
        mats = np.array([m.material for m in self.meshes], dtype=MATERIAL_DTYPE)
        return (
            np.ascontiguousarray(np.concatenate(verts)),
            np.ascontiguousarray(np.concatenate(idx)),
            np.ascontiguousarray(np.concatenate(tri_mesh)),
            mats,
        )


def _quads_to_mesh(quads, material) -> TriangleMesh:
    q = np.asarray(quads, dtype=np.float32).reshape(-1, 4, 3)
    n = len(q)
    base = (np.arange(n, dtype=np.uint32) * 4)[:, None]
    tri = np.concatenate([base + np.array([0, 1, 2], np.uint32), base + np.array([0, 2, 3], np.uint32)], axis=1)
    return TriangleMesh(q.reshape(-1, 3).copy(), tri.reshape(-1, 3).astype(np.uint32), material)


def add_box(model: Model, material, pos, extend) -> None:
    """addBox (Model.cpp:214-286): `extend` is the HALF extent; 36 unshared vertices, 12 triangles in
    the reference's order (front, back, left, right, top, bottom), one mesh."""
    f = np.float32
    px, py, pz = (f(v) for v in pos)
    ex, ey, ez = (f(v) for v in extend)
    A = (-ex + px, -ey + py, ez + pz)
    B = (ex + px, -ey + py, ez + pz)
    C = (ex + px, ey + py, ez + pz)
    D = (-ex + px, ey + py, ez + pz)
    E = (-ex + px, -ey + py, -ez + pz)
    F = (ex + px, -ey + py, -ez + pz)
    G = (ex + px, ey + py, -ez + pz)
    H = (-ex + px, ey + py, -ez + pz)
    verts = [A, B, C, A, C, D, E, H, G, E, G, F, E, A, D, E, D, H, B, F, G, B, G, C, D, C, G, D, G, H, E, A, B, E, B, F]
    v = np.array(verts, np.float32)
    idx = np.arange(36, dtype=np.uint32).reshape(12, 3)
    nrm = np.repeat(np.array([(0, 0, 1), (0, 0, -1), (-1, 0, 0), (1, 0, 0), (0, 1, 0), (0, -1, 0)], np.float32), 6, axis=0)  # :243-258
    model.meshes.append(TriangleMesh(v, idx, np.array(material, dtype=MATERIAL_DTYPE), texcoord=np.zeros((36, 2), np.float32), normal=nrm))


def cornell_box() -> Model:
    """Cornell-32: the classic 32-triangle box (556 x 548.8 x 559.2, open front at z=0), 4 meshes.
    Materials are Disney defaults with the classic colours; the ceiling quad carries emission 15
    (visible on primary hits only — deviceProgram.cu:558-560)."""
    white = Material(color=(0.8, 0.8, 0.8))
    red = Material(color=(0.8, 0.05, 0.05))
    green = Material(color=(0.05, 0.8, 0.05))
    light = Material(color=(0.8, 0.8, 0.8), emission=(15.0, 15.0, 15.0))
    W = [
        [(552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2)],  # floor
        [(556, 548.8, 0), (556, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0)],  # ceiling
        [(549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556, 548.8, 559.2)],  # back
        # short block
        [(130, 165, 65), (82, 165, 225), (240, 165, 272), (290, 165, 114)],
        [(290, 0, 114), (290, 165, 114), (240, 165, 272), (240, 0, 272)],
        [(130, 0, 65), (130, 165, 65), (290, 165, 114), (290, 0, 114)],
        [(82, 0, 225), (82, 165, 225), (130, 165, 65), (130, 0, 65)],
        [(240, 0, 272), (240, 165, 272), (82, 165, 225), (82, 0, 225)],
        # tall block
        [(423, 330, 247), (265, 330, 296), (314, 330, 456), (472, 330, 406)],
        [(423, 0, 247), (423, 330, 247), (472, 330, 406), (472, 0, 406)],
        [(472, 0, 406), (472, 330, 406), (314, 330, 456), (314, 0, 456)],
        [(314, 0, 456), (314, 330, 456), (265, 330, 296), (265, 0, 296)],
        [(265, 0, 296), (265, 330, 296), (423, 330, 247), (423, 0, 247)],
    ]
    R = [[(552.8, 0, 0), (549.6, 0, 559.2), (556, 548.8, 559.2), (556, 548.8, 0)]]
    G = [[(0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2)]]
    L = [[(343, 548.6, 227), (343, 548.6, 332), (213, 548.6, 332), (213, 548.6, 227)]]
    m = Model([_quads_to_mesh(W, white), _quads_to_mesh(R, red), _quads_to_mesh(G, green), _quads_to_mesh(L, light)])
    assert m.num_triangles == 32
    return m


def checker_texture(w=64, h=32, cells=8, c0=(220, 40, 40), c1=(240, 240, 200)) -> Texture:
    """RGBA8 checkerboard with a per-pixel ramp so bilinear filtering and wrap are exercised."""
    ys, xs = np.mgrid[0:h, 0:w]
    sel = ((xs * cells // w) + (ys * cells // h)) % 2
    rgb = np.where(sel[..., None] == 0, np.array(c0), np.array(c1)).astype(np.uint32)
    rgb[..., 2] = (rgb[..., 2] + xs * 3 + ys * 5) % 256
    pix = rgb[..., 0] | (rgb[..., 1] << 8) | (rgb[..., 2] << 16) | (np.uint32(255) << 24)
    return Texture(np.ascontiguousarray(pix.astype(np.uint32)))


def textured_scene() -> Model:
    """A textured ground quad (texcoords run past [0,1] → wrap), a textured tilted quad, an untextured box and a mesh
    that names a texture but has no texcoords (hasTexture && texcoord is false → material colour, deviceProgram.cu:512)."""
    m = Model()
    ground = _quads_to_mesh([[(-4, 0, -4), (4, 0, -4), (4, 0, 4), (-4, 0, 4)]], Material(color=(0.1, 0.9, 0.1)))
    ground.texcoord = np.array([[-1.5, -1.5], [2.5, -1.5], [2.5, 2.5], [-1.5, 2.5]], np.float32)
    ground.diffuseTextureID = 0
    wall = _quads_to_mesh([[(-2, 0, 2), (2, 0, 2.5), (2, 2.5, 3), (-2, 2.5, 2.5)]], Material(color=(0.9, 0.1, 0.1), roughness=0.4))
    wall.texcoord = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    wall.diffuseTextureID = 1
    m.meshes += [ground, wall]
    add_box(m, Material(color=(0.3, 0.4, 0.8)), (0.0, 0.5, 0.0), (0.5, 0.5, 0.5))
    m.meshes[-1].diffuseTextureID = 0  # named texture, no texcoords
    m.textures = [checker_texture(), checker_texture(48, 40, 6, (30, 60, 200), (250, 250, 60))]
    return m


CORNELL_CAMERA = dict(eye=(278.0, 273.0, -900.0), lookat=(278.0, 273.0, 0.0), up=(0.0, 1.0, 0.0), fovY=35.0)
# the reference's lost-empire camera (HelloPathtracing_sv4_vmv23/main.cpp:227-231)
TERRAIN_CAMERA = dict(eye=(-70.0, 40.0, 100.0), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fovY=45.0)


def two_box_scene(shadow_catcher: bool = True) -> Model:
    """The reference's commented-out procedural scene (main.cpp:165-169): a unit box on an
    8 x 0.2 x 8 ground box flagged SHADOW_CATCHER."""
    m = Model()
    boxMat = Material()
    add_box(m, boxMat, (0.0, 0.5, 0.0), (0.5, 0.5, 0.5))
    if shadow_catcher:
        boxMat["flags"] |= MATERIAL_FLAG_SHADOW_CATCHER
    add_box(m, boxMat, (0.0, -0.1, 0.0), (4.0, 0.1, 4.0))
    return m


def catcher_stack_scene() -> Model:
    """Shadow-catcher stress scene (no reference counterpart): the two-box scene plus a second catcher slab standing behind the
    box and an ordinary diffuse slab BELOW the catcher ground.  Secondary rays cross the catcher slabs (two faces each, two
    pass-throughs that do not consume depth, deviceProgram.cu:503-508) and still find ordinary geometry afterwards, so a path
    is traced up to max_depth + 1 + (number of pass-throughs) times."""
    m = Model()
    add_box(m, Material(color=(0.85, 0.8, 0.7), metallic=0.6, roughness=0.35), (0.0, 0.5, 0.0), (0.5, 0.5, 0.5))
    catcher = Material()
    catcher["flags"] |= MATERIAL_FLAG_SHADOW_CATCHER
    add_box(m, catcher, (0.0, -0.1, 0.0), (4.0, 0.1, 4.0))   # ground slab
    add_box(m, catcher, (0.0, 1.5, 2.0), (3.0, 1.5, 0.1))    # wall slab behind the box
    add_box(m, Material(color=(0.3, 0.5, 0.8)), (0.0, -1.2, 0.0), (5.0, 0.4, 5.0))  # ordinary slab under the ground
    add_box(m, Material(color=(0.8, 0.3, 0.2), roughness=0.6), (0.0, 1.2, 3.2), (2.0, 1.0, 0.3))  # ordinary slab behind the wall
    return m


TWO_BOX_CAMERA = dict(eye=(3.0, 2.5, -4.0), lookat=(0.0, 0.4, 0.0), up=(0.0, 1.0, 0.0), fovY=40.0)


def material_presets():
    """8 presets exercising every BSDFSample/BSDFEval branch (Disney.cuh:196-426)."""
    return [
        Material(color=(0.25, 0.45, 0.15)),  # matte grass
        Material(color=(0.45, 0.35, 0.25), roughness=0.8),  # dirt
        Material(color=(0.5, 0.5, 0.52), roughness=0.5, specular=0.6),  # rock
        Material(color=(0.9, 0.9, 0.95), roughness=0.3, clearcoat=1.0, clearcoatGloss=0.8),  # snow / clearcoat
        Material(color=(0.9, 0.7, 0.3), metallic=1.0, roughness=0.25),  # metal
        Material(color=(0.8, 0.4, 0.3), subsurface=0.6, roughness=0.7),  # subsurface clay
        Material(color=(0.9, 0.95, 1.0), transmission=0.9, roughness=0.05),  # glass-like
        Material(color=(0.2, 0.3, 0.7), specularTint=0.8, roughness=0.4, specular=0.9),  # tinted plastic
    ]


def _value_noise(n: int, seed: int, octaves=5) -> np.ndarray:
    rng = np.random.default_rng(seed)
    out = np.zeros((n, n), np.float64)
    amp, tot = 1.0, 0.0
    xs = np.arange(n, dtype=np.float64)
    for o in range(octaves):
        cells = 4 * (2**o)
        lat = rng.random((cells + 2, cells + 2))
        g = xs * (cells / n)
        i0 = np.floor(g).astype(np.int64)
        f = g - i0
        f = f * f * (3.0 - 2.0 * f)
        a = lat[i0][:, i0]
        b = lat[i0 + 1][:, i0]
        c = lat[i0][:, i0 + 1]
        d = lat[i0 + 1][:, i0 + 1]
        fx, fz = f[:, None], f[None, :]
        out += amp * ((a * (1 - fx) + b * fx) * (1 - fz) + (c * (1 - fx) + d * fx) * fz)
        tot += amp
        amp *= 0.5
    return out / tot


def voxel_terrain(n: int = 360, seed: int = 1234, extent: float = 100.0, target_tris: int = 1_000_000) -> Model:
    """~1M-triangle voxel terrain "in the spirit of lost_empire" (SURVEY.md §8d): value-noise
    integer heights on an n x n column grid; top quads + exposed side quads (one quad per voxel
    face) as axis-aligned triangle pairs; 8 material presets by height band → 8 meshes.
    The height amplitude is bisected (deterministically) so the count lands near target_tris."""
    noise = _value_noise(n, seed)
    cell = 2.0 * extent / n

    def heights(amp):
        return np.floor(noise * amp).astype(np.int64)

    def count(H):
        dx = np.abs(np.diff(H, axis=0)).sum()
        dz = np.abs(np.diff(H, axis=1)).sum()
        hmin = H.min()
        skirt = (H[0, :] - hmin).sum() + (H[-1, :] - hmin).sum() + (H[:, 0] - hmin).sum() + (H[:, -1] - hmin).sum()
        return 2 * int(n * n + dx + dz + skirt)

    lo, hi = 1.0, 400.0
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        if count(heights(mid)) < target_tris:
            lo = mid
        else:
            hi = mid
    H = heights(hi)
    hmin, hmax = int(H.min()), int(H.max())
    y0 = -0.5 * (hmax - hmin) * cell  # centre vertically around 0

    def X(i):
        return -extent + i * cell

    def Y(k):
        return y0 + (k - hmin) * cell

    quads, levels = [], []
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    # tops
    x0, x1, z0, z1, y = X(ii), X(ii + 1), X(jj), X(jj + 1), Y(H)
    top = np.stack(
        [np.stack([x0, y, z0], -1), np.stack([x0, y, z1], -1), np.stack([x1, y, z1], -1), np.stack([x1, y, z0], -1)], -2
    ).reshape(-1, 4, 3)
    quads.append(top)
    levels.append(H.reshape(-1))

    def side_faces(Ha, Hb, fixed_axis, fixed_coord, span0, span1):
        """unit quads between columns a and b for k in [min,max) on the plane fixed_axis=fixed_coord"""
        lo_, hi_ = np.minimum(Ha, Hb), np.maximum(Ha, Hb)
        cnt = (hi_ - lo_).astype(np.int64)
        tot = int(cnt.sum())
        if tot == 0:
            return np.zeros((0, 4, 3)), np.zeros((0,), np.int64)
        rep = np.repeat(np.arange(len(cnt)), cnt)
        offs = np.arange(tot) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        k = lo_[rep] + offs
        f, s0, s1 = fixed_coord[rep], span0[rep], span1[rep]
        ya, yb = Y(k), Y(k + 1)
        if fixed_axis == 0:
            q = np.stack(
                [np.stack([f, ya, s0], -1), np.stack([f, yb, s0], -1), np.stack([f, yb, s1], -1), np.stack([f, ya, s1], -1)], -2
            )
        else:
            q = np.stack(
                [np.stack([s0, ya, f], -1), np.stack([s0, yb, f], -1), np.stack([s1, yb, f], -1), np.stack([s1, ya, f], -1)], -2
            )
        return q, k + 1

    # interior faces along x and z
    Ha, Hb = H[:-1, :].reshape(-1), H[1:, :].reshape(-1)
    q, lv = side_faces(Ha, Hb, 0, X(ii[1:, :]).reshape(-1), X(jj[1:, :]).reshape(-1), X(jj[1:, :] + 1).reshape(-1))
    quads.append(q); levels.append(lv)
    Ha, Hb = H[:, :-1].reshape(-1), H[:, 1:].reshape(-1)
    q, lv = side_faces(Ha, Hb, 2, X(jj[:, 1:]).reshape(-1), X(ii[:, 1:]).reshape(-1), X(ii[:, 1:] + 1).reshape(-1))
    quads.append(q); levels.append(lv)
    # skirts down to hmin on the four borders
    base = np.full(n, hmin, np.int64)
    ar = np.arange(n)
    for fixed_axis, fc, Hedge in ((0, X(0), H[0, :]), (0, X(n), H[-1, :])):
        q, lv = side_faces(base, Hedge, fixed_axis, np.full(n, fc), X(ar), X(ar + 1))
        quads.append(q); levels.append(lv)
    for fixed_axis, fc, Hedge in ((2, X(0), H[:, 0]), (2, X(n), H[:, -1])):
        q, lv = side_faces(base, Hedge, fixed_axis, np.full(n, fc), X(ar), X(ar + 1))
        quads.append(q); levels.append(lv)

    Q = np.concatenate(quads).astype(np.float32)
    LV = np.concatenate(levels)
    band = np.clip(((LV - hmin) * 8) // max(1, (hmax - hmin + 1)), 0, 7)
    presets = material_presets()
    model = Model()
    for b in range(8):
        sel = Q[band == b]
        if len(sel):
            model.meshes.append(_quads_to_mesh(sel, presets[b]))
    return model


def band_texture(band: int, size: int = 1024) -> Texture:
    """Deterministic RGBA8 texture of one height band: three octaves of value noise in the band's hue plus a brick-like grid, so that
    neighbouring texels differ (a filter or layout error shows) and the texture tiles without a seam only approximately (wrap shows)."""
    rng = np.random.default_rng(9000 + band)
    base = np.array([(0.55, 0.45, 0.30), (0.35, 0.55, 0.25), (0.50, 0.50, 0.50), (0.70, 0.65, 0.45), (0.30, 0.40, 0.60), (0.60, 0.35, 0.30),
                     (0.80, 0.80, 0.85), (0.25, 0.30, 0.25)][band % 8])
    n = _value_noise(size, 500 + band, octaves=6)
    ys, xs = np.mgrid[0:size, 0:size]
    grid = ((xs % 64 < 3) | (ys % 32 < 3)).astype(np.float64)
    speck = rng.random((size, size))
    rgb = np.clip((0.55 + 0.9 * (n[..., None] - 0.5)) * base[None, None, :] * (1.0 - 0.45 * grid[..., None]) + 0.08 * (speck[..., None] - 0.5), 0.0, 1.0)
    px = (rgb * 255.0 + 0.5).astype(np.uint32)
    return Texture(np.ascontiguousarray(px[..., 0] | (px[..., 1] << 8) | (px[..., 2] << 16) | np.uint32(0xFF000000)))


def textured_terrain(n: int = 360, seed: int = 1234, extent: float = 100.0, target_tris: int = 1_000_000, tex_size: int = 1024, tile: float = 12.5) -> Model:
    """The C3 terrain with what the reference's real inputs have (main.cpp:171-194: sponza, lost_empire, San Miguel through loadOBJ):
    every mesh carries texcoords and a diffuse texture, so every closest hit takes the tex2D branch (deviceProgram.cu:512-523).
    One tex_size^2 RGBA8 texture per height band; texcoords = world position / tile (top faces: x,z; side faces: x+z, y), i.e. they run
    over roughly [-8, 8] and exercise the wrap on both sides of 0."""
    m = voxel_terrain(n, seed, extent, target_tris)
    for b, mesh in enumerate(m.meshes):
        v = mesh.vertex.reshape(-1, 4, 3)
        e1, e2 = v[:, 1] - v[:, 0], v[:, 3] - v[:, 0]
        nrm = np.cross(e1, e2)
        top = np.abs(nrm[:, 1]) > 0
        uv = np.empty((len(v), 4, 2), np.float32)
        t = np.float32(tile)
        uv[top, :, 0] = v[top, :, 0] / t
        uv[top, :, 1] = v[top, :, 2] / t
        uv[~top, :, 0] = (v[~top, :, 0] + v[~top, :, 2]) / t
        uv[~top, :, 1] = v[~top, :, 1] / t
        mesh.texcoord = np.ascontiguousarray(uv.reshape(-1, 2))
        mesh.diffuseTextureID = b
        m.textures.append(band_texture(b, tex_size))
    return m


def _fmt9(a: np.ndarray):
    """float32 -> decimal with 9 significant digits: within 5e-10 relative of the value, far inside the half-ulp (>= 3e-8) any
    decimal -> double -> float reader needs to land on the same float again (tinyobjloader's digit-by-digit parser included)."""
    return np.char.mod("%.9g", a.astype(np.float64))


def write_obj(model: Model, obj_path: str) -> str:
    """Writes `model` as OBJ + MTL + PNG textures the way a DCC tool would: one `o` per mesh with its `usemtl`, `v` / `vt` per vertex,
    faces as QUADS when consecutive triangle pairs form (a,b,c),(a,c,d) (the loader's triangulation gives the pair back), triangles
    otherwise.  objloader.load_obj(obj_path) reproduces the model's triangles in order, with bit-identical corner positions and texcoords
    (vertex numbering inside a mesh changes: addVertex numbers corners in the order loadOBJ meets them)."""
    import os

    from PIL import Image

    d = os.path.dirname(os.path.abspath(obj_path))
    os.makedirs(d, exist_ok=True)
    stem = os.path.splitext(os.path.basename(obj_path))[0]
    with open(os.path.join(d, stem + ".mtl"), "w") as f:
        for i, mesh in enumerate(model.meshes):
            c, e = mesh.material["color"], mesh.material["emission"]
            f.write(f"newmtl m{i}\nKd {float(c[0]):.9g} {float(c[1]):.9g} {float(c[2]):.9g}\nKe {float(e[0]):.9g} {float(e[1]):.9g} {float(e[2]):.9g}\n")
            if mesh.diffuseTextureID >= 0:
                f.write(f"map_Kd {stem}_tex{mesh.diffuseTextureID}.png\n")
    for t, tex in enumerate(model.textures):
        px = tex.pixel[::-1]  # the file holds the top row first; loadTexture mirrors it back (Model.cpp:112-121)
        rgba = np.stack([(px >> s) & 0xFF for s in (0, 8, 16, 24)], -1).astype(np.uint8)
        Image.fromarray(rgba, "RGBA").save(os.path.join(d, f"{stem}_tex{t}.png"))
    with open(obj_path, "w") as f:
        f.write(f"mtllib {stem}.mtl\n")
        vbase = tbase = 0
        for i, mesh in enumerate(model.meshes):
            f.write(f"o mesh{i}\nusemtl m{i}\n")
            V = _fmt9(mesh.vertex)
            f.write("\n".join("v " + " ".join(r) for r in V) + "\n")
            has_tc = mesh.texcoord is not None and len(mesh.texcoord)
            if has_tc:
                T = _fmt9(mesh.texcoord)
                f.write("\n".join("vt " + " ".join(r) for r in T) + "\n")
            idx = mesh.index.astype(np.int64)
            nt = len(idx)
            quad = np.zeros(nt, bool)
            if nt >= 2:
                a, b = idx[:-1], idx[1:]
                quad[:-1] = (a[:, 0] == b[:, 0]) & (a[:, 2] == b[:, 1])
                quad[1:] &= ~quad[:-1]  # pairs do not overlap
                quad[-1] = False
            k = 0
            out = []
            while k < nt:
                ids = [idx[k, 0], idx[k, 1], idx[k, 2]]
                if quad[k]:
                    ids.append(idx[k + 1, 2])
                    k += 2
                else:
                    k += 1
                out.append("f " + " ".join(f"{vbase + j + 1}/{tbase + j + 1}" if has_tc else f"{vbase + j + 1}" for j in ids))
            f.write("\n".join(out) + "\n")
            vbase += len(mesh.vertex)
            tbase += len(mesh.texcoord) if has_tc else 0
    return obj_path


# ------------------------------------------------------------------ second 1 M-triangle workload: "teapots in a stadium"
STADIUM_CAMERA = dict(eye=(-62.0, 9.0, 21.0), lookat=(0.0, 4.0, 0.0), up=(0.0, 1.0, 0.0), fovY=50.0)


def _rotation(rng) -> np.ndarray:
    """uniformly random rotation matrix (QR of a Gaussian matrix, sign-fixed)"""
    q, r = np.linalg.qr(rng.standard_normal((3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def _grid_quads(P: np.ndarray) -> np.ndarray:
    """(nu+1, nv+1, 3) grid of surface points -> (nu*nv, 4, 3) quads"""
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    return np.stack([a, b, c, d], -2).reshape(-1, 4, 3)


def _tube(curve: np.ndarray, radius: np.ndarray, nv: int) -> np.ndarray:
    """closed tube of `nv` sides around a closed polyline (n,3); radius (n,) or (n,nv): rotated, elongated, non-axis-aligned triangles"""
    n = len(curve)
    T = np.roll(curve, -1, 0) - np.roll(curve, 1, 0)
    T /= np.linalg.norm(T, axis=1, keepdims=True)
    ref = np.array([0.31, 0.87, 0.38])
    N = np.cross(T, ref)
    N /= np.linalg.norm(N, axis=1, keepdims=True)
    B = np.cross(T, N)
    phi = np.linspace(0.0, 2.0 * np.pi, nv + 1)
    rad = radius if np.ndim(radius) == 2 else np.asarray(radius)[:, None] * np.ones(nv + 1)
    P = curve[:, None, :] + rad[..., None] * (np.cos(phi)[None, :, None] * N[:, None, :] + np.sin(phi)[None, :, None] * B[:, None, :])
    P = np.concatenate([P, P[:1]], 0)  # close along the curve
    return _grid_quads(P)


def stadium_scene(target_tris: int = 1_000_000, seed: int = 4321) -> Model:
    """~1 M triangles that are everything the voxel terrain is not (the reference's real inputs are sponza /
    San Miguel / lost_empire, HelloPathtracing_sv4_vmv23/main.cpp:188-194): a stadium bowl of long thin bench and railing strips, a
    colonnade of fluted, slightly tilted columns on an ellipse, a partial roof of huge tilted quads over a two-triangle ground
    (edges of 600 units), a field strewn with randomly oriented blade triangles (aspect 20-200 : 1), and in the middle "teapots":
    finely tessellated, noise-displaced torus knots and spheres under random rotations (edges of 0.003-0.05 units).  Edge lengths span
    five decades, almost nothing is axis-aligned, the camera stands inside.  Deterministic in (target_tris, seed); the object counts
    scale with target_tris so that small versions serve the brute-force parity tests."""
    rng = np.random.default_rng(seed)
    s = target_tris / 1_000_000.0
    presets = material_presets()
    parts = {k: [] for k in range(8)}  # quads by material preset
    tris = {k: [] for k in range(8)}   # loose triangles by material preset

    # ground: two huge triangles; roof: a ring of large tilted quads over the stands
    g = 300.0
    parts[1].append(np.array([[[-g, 0, -g], [-g, 0, g], [g, 0, g], [g, 0, -g]]], np.float64))
    nroof = max(8, int(48 * min(1.0, s * 4)))
    th = np.linspace(0, 2 * np.pi, nroof + 1)
    inner = np.stack([70 * np.cos(th), np.full_like(th, 34.0), 52 * np.sin(th)], -1)
    outer = np.stack([118 * np.cos(th), np.full_like(th, 52.0), 96 * np.sin(th)], -1)
    parts[2].append(np.stack([inner[:-1], outer[:-1], outer[1:], inner[1:]], -2))

    # stands: tiers of bench tops + risers (long thin quads along the ellipse), thin railings every few tiers
    ntier = max(4, int(round(44 * min(1.0, s * 3))))
    nseg = max(48, int(round(640 * min(1.0, s * 3))))
    th = np.linspace(0, 2 * np.pi, nseg + 1)
    for t in range(ntier):
        a0, b0, y0 = 72 + 1.0 * t, 54 + 0.95 * t, 1.0 + 0.7 * t
        a1, b1, y1 = a0 + 1.0, b0 + 0.95, y0 + 0.7
        p00 = np.stack([a0 * np.cos(th), np.full_like(th, y0), b0 * np.sin(th)], -1)
        p01 = np.stack([a1 * np.cos(th), np.full_like(th, y0), b1 * np.sin(th)], -1)
        p11 = np.stack([a1 * np.cos(th), np.full_like(th, y1), b1 * np.sin(th)], -1)
        parts[t % 3].append(np.stack([p00[:-1], p01[:-1], p01[1:], p00[1:]], -2))  # bench top
        parts[2].append(np.stack([p01[:-1], p11[:-1], p11[1:], p01[1:]], -2))      # riser
        if t % 4 == 0:  # railing: a 0.04-wide strip in 96 long pieces — aspect ratios of several hundred
            tr = np.linspace(0, 2 * np.pi, 97)
            r0 = np.stack([a0 * np.cos(tr), np.full_like(tr, y0 + 1.1), b0 * np.sin(tr)], -1)
            r1 = r0 + np.array([0.0, 0.04, 0.0])
            parts[4].append(np.stack([r0[:-1], r1[:-1], r1[1:], r0[1:]], -2))

    # colonnade: fluted columns on an ellipse, each turned to face the centre and tilted by up to 2 degrees
    ncol = max(6, int(round(96 * min(1.0, s * 3))))
    nside, nh = 24, max(4, int(round(18 * min(1.0, s * 4))))
    phi = np.linspace(0, 2 * np.pi, nside + 1)
    hh = np.linspace(0.0, 1.0, nh + 1)
    for c in range(ncol):
        ang = 2 * np.pi * c / ncol
        base = np.array([66 * np.cos(ang), 0.0, 48 * np.sin(ang)])
        rad = (0.9 - 0.15 * hh)[:, None] * (1.0 + 0.06 * np.cos(12 * phi))[None, :]  # taper + flutes
        local = np.stack([rad * np.cos(phi)[None, :], 16.0 * hh[:, None] * np.ones_like(phi)[None, :], rad * np.sin(phi)[None, :]], -1)
        tilt = np.deg2rad(rng.uniform(-2, 2, 2))
        cx, sx, cz, sz = np.cos(tilt[0]), np.sin(tilt[0]), np.cos(tilt[1]), np.sin(tilt[1])
        Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        ca, sa = np.cos(-ang), np.sin(-ang)
        Ry = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]])
        P = local @ (Ry @ Rx @ Rz).T + base
        parts[3 if c % 2 else 2].append(_grid_quads(P))
        # lintel: a rotated box-like beam (4 long quads) to the next column
        nxt = np.array([66 * np.cos(ang + 2 * np.pi / ncol), 16.0, 48 * np.sin(ang + 2 * np.pi / ncol)])
        top = base + np.array([0.0, 16.0, 0.0])
        d = nxt - top
        side = np.cross(d, [0, 1, 0.0])
        side = 0.6 * side / np.linalg.norm(side)
        up = np.array([0.0, 0.9, 0.0])
        corners = [top - side, top + side, top + side + up, top - side + up]
        ring = np.array(corners + [corners[0]])
        parts[2].append(np.stack([ring[:-1], ring[:-1] + d, ring[1:] + d, ring[1:]], -2))

    # blades: randomly oriented long thin triangles all over the field
    fixed = sum(len(q) * 2 for v in parts.values() for q in v)
    budget = max(0, target_tris - fixed)
    nblade = int(budget * 0.16)
    c = np.stack([rng.uniform(-60, 60, nblade), rng.uniform(0.0, 0.3, nblade), rng.uniform(-44, 44, nblade)], -1)
    d = rng.standard_normal((nblade, 3))
    d[:, 1] = np.abs(d[:, 1]) + 0.3
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    L = np.exp(rng.uniform(np.log(0.08), np.log(3.0), nblade))
    wdt = L / np.exp(rng.uniform(np.log(20.0), np.log(200.0), nblade))
    sd = np.cross(d, rng.standard_normal((nblade, 3)))
    sd /= np.linalg.norm(sd, axis=1, keepdims=True)
    blade = np.stack([c - sd * wdt[:, None], c + sd * wdt[:, None], c + d * L[:, None]], 1)
    which = rng.integers(0, 2, nblade)
    tris[0].append(blade[which == 0])
    tris[7].append(blade[which == 1])

    # "teapots": finely tessellated, displaced torus knots and spheres under random rotations, scattered over the field
    budget -= nblade
    per_obj = 16384
    nobj = max(1, budget // per_obj)
    per_obj = max(256, budget // nobj)
    for o in range(nobj):
        R = _rotation(rng)
        scale = np.exp(rng.uniform(np.log(0.25), np.log(2.2)))
        centre = np.array([rng.uniform(-45, 45), scale * 1.6 + rng.uniform(0.0, 6.0), rng.uniform(-32, 32)])
        mat = int(rng.integers(3, 8))
        if o % 3 != 2:  # torus knot tube: 8 sides, (per_obj / 16) segments
            nu = max(16, per_obj // 16)
            t = np.linspace(0, 2 * np.pi, nu, endpoint=False)
            p_, q_ = [(2, 3), (3, 5), (2, 5), (3, 4)][o % 4]
            rr = 1.0 + 0.45 * np.cos(q_ * t)
            curve = np.stack([rr * np.cos(p_ * t), 0.45 * np.sin(q_ * t), rr * np.sin(p_ * t)], -1)
            radius = 0.11 * (1.0 + 0.35 * np.sin(17 * t) * np.cos(5 * t))
            Q = _tube(curve, radius, 8)
        else:  # displaced sphere
            nu = max(8, int(np.sqrt(per_obj / 4)))
            nv = max(8, per_obj // (2 * nu))
            u = np.linspace(0, np.pi, nu + 1)[:, None]
            v = np.linspace(0, 2 * np.pi, nv + 1)[None, :]
            rad = 1.0 + 0.12 * np.sin(9 * u) * np.cos(7 * v) + 0.05 * np.sin(23 * v + 3 * u)
            P = np.stack([rad * np.sin(u) * np.cos(v), rad * np.cos(u) * np.ones_like(v), rad * np.sin(u) * np.sin(v)], -1)
            Q = _grid_quads(P)
        parts[mat].append((Q.reshape(-1, 3) * scale) @ R.T + centre)

    model = Model()
    for k in range(8):
        vs, ix, base = [], [], 0
        for q in parts[k]:
            q = np.asarray(q, np.float64).reshape(-1, 4, 3)
            m = _quads_to_mesh(q, presets[k])
            vs.append(m.vertex); ix.append(m.index + np.uint32(base)); base += len(m.vertex)
        for t3 in tris[k]:
            t3 = np.asarray(t3, np.float32).reshape(-1, 3)
            if len(t3):
                vs.append(t3); ix.append(np.arange(len(t3), dtype=np.uint32).reshape(-1, 3) + np.uint32(base)); base += len(t3)
        if vs:
            model.meshes.append(TriangleMesh(np.ascontiguousarray(np.concatenate(vs), np.float32), np.ascontiguousarray(np.concatenate(ix), np.uint32), presets[k]))
    return model


# ------------------------------------------------------------------ probes


@dataclass
class ProbeData:  # Probe.h:8-88
    width: int
    height: int
    data: np.ndarray  # (h,w,4) f32
    pdfValuesX: np.ndarray = None
    cdfValuesX: np.ndarray = None
    pdfValuesY: np.ndarray = None
    cdfValuesY: np.ndarray = None
    valid: bool = False

    def BuildCDF(self, builder=None) -> "ProbeData":
        """ProbeData::BuildCDF (Probe.h:29-77). `builder(data,w,h)->(pdfX,cdfX,pdfY,cdfY)` is the
        native implementation (libptamd's pt_build_cdf); the numpy fallback below performs the same
        sequential float32 running sums (np.cumsum on f32 accumulates left to right in f32)."""
        if builder is not None:
            self.pdfValuesX, self.cdfValuesX, self.pdfValuesY, self.cdfValuesY = builder(self.data, self.width, self.height)
        else:
            d = self.data.astype(np.float32)
            with np.errstate(divide="ignore", invalid="ignore"):  # an all-black row gives 1/0 and NaN entries, like Probe.h:53
                return self._build_numpy(d)
        self.valid = True
        return self

    def _build_numpy(self, d):
        if True:  # (kept as one block: the arithmetic order below mirrors Probe.h line by line)
            lum = (d[..., 0] * np.float32(0.3) + d[..., 1] * np.float32(0.6)) + d[..., 2] * np.float32(0.1)
            cx = np.cumsum(lum, axis=1, dtype=np.float32)
            tot = cx[:, -1].copy()
            inv = (np.float32(1.0) / tot).astype(np.float32)
            self.pdfValuesX = (lum * inv[:, None]).astype(np.float32)
            self.cdfValuesX = (cx * inv[:, None]).astype(np.float32)
            cy = np.cumsum(tot, dtype=np.float32)
            self.pdfValuesY = (tot / cy[-1]).astype(np.float32)
            self.cdfValuesY = (cy / cy[-1]).astype(np.float32)
        self.valid = True
        return self


def _uv_to_dir(u, v):
    # Probe.cuh:48-58 (float64 here: used only to paint the synthetic probe)
    th, ph = v * math.pi, u * 2.0 * math.pi
    return np.stack([-np.sin(th) * np.cos(ph), np.cos(th), -np.sin(th) * np.sin(ph)], -1)


SUN_DIR = np.array([-0.35, 0.70, -0.62]) / np.linalg.norm([-0.35, 0.70, -0.62])


def sky_probe(width: int = 2048, height: int = 1024, sun_radius_deg: float = 2.0, sun_radiance: float = 50.0) -> ProbeData:
    """Vertical sky gradient (zenith (1,1,1.2) → horizon .6 → ground .1) + one sun disc centred on
    SUN_DIR (shines into the Cornell box's open front). Texel (col,row) maps to the direction the
    sampler returns for it: ProbeUVToDir(col/W, row/H) (Probe.cuh:157-167, texel corner)."""
    v = (np.arange(height, dtype=np.float64) / height)[:, None]
    u = (np.arange(width, dtype=np.float64) / width)[None, :]
    t_up = np.clip(v * 2.0, 0.0, 1.0)
    t_dn = np.clip((v - 0.5) * 2.0, 0.0, 1.0)
    zen = np.array([1.0, 1.0, 1.2])
    hor = np.array([0.6, 0.6, 0.6])
    gnd = np.array([0.1, 0.1, 0.1])
    col = np.where((v < 0.5)[..., None], zen * (1 - t_up[..., None]) + hor * t_up[..., None], hor * (1 - t_dn[..., None]) + gnd * t_dn[..., None])
    col = np.broadcast_to(col, (height, width, 3)).copy()
    d = _uv_to_dir(np.broadcast_to(u, (height, width)), np.broadcast_to(v, (height, width)))
    cosang = d @ SUN_DIR
    col[cosang >= math.cos(math.radians(sun_radius_deg))] = sun_radiance
    data = np.concatenate([col, np.ones((height, width, 1))], -1).astype(np.float32)
    return ProbeData(width, height, np.ascontiguousarray(data))


def constant_probe(width: int = 64, height: int = 32, value: float = 1.0) -> ProbeData:
    """The reference's loadColor() constant probe (HelloPathtracing_sv4_vmv23/main.cpp:167-180)."""
    data = np.full((height, width, 4), value, np.float32)
    data[..., 3] = 1.0
    return ProbeData(width, height, data)


def disc_probe(width: int = 100, height: int = 50) -> ProbeData:
    """Probe.cuh:207-242's commented-out ProbeCreateTest: radiance 10 where dot(dir,+y) >= .95, else 0.05."""
    v = (np.arange(height, dtype=np.float64) / height)[:, None]
    u = (np.arange(width, dtype=np.float64) / width)[None, :]
    d = _uv_to_dir(np.broadcast_to(u, (height, width)), np.broadcast_to(v, (height, width)))
    val = np.where(d[..., 1] >= 0.95, 10.0, 0.05)
    data = np.stack([val, val, val, np.ones_like(val)], -1).astype(np.float32)
    return ProbeData(width, height, np.ascontiguousarray(data))


def spots_probe(width: int = 1000, height: int = 64, seed: int = 3, fill: float = 0.01) -> ProbeData:
    """A search-hostile probe: black except for a few texels whose radiance spans six decades, with some rows entirely black (their
    conditional CDF is 0/0 = NaN like the reference's BuildCDF, Probe.h:29-77) and long flat stretches in every other row's CDF."""
    rng = np.random.default_rng(seed)
    val = np.where(rng.random((height, width)) < fill, 10.0 ** rng.uniform(-3, 3, (height, width)), 0.0)
    val[rng.integers(0, height, max(1, height // 8))] = 0.0
    val[height // 2, : width // 3] = 1.0  # one row with a long linear ramp in its CDF
    data = np.stack([val, val * 0.5, val * 0.25, np.ones_like(val)], -1).astype(np.float32)
    return ProbeData(width, height, np.ascontiguousarray(data))


def uvw_frame(eye, lookat, up, fovY, aspect):
    """sutil::Camera::UVWFrame (sutil/Camera.cpp:34-45) in float32; tanf via the host libm."""
    f = np.float32
    eye, lookat, up = (np.asarray(a, f) for a in (eye, lookat, up))

    def dot(a, b):
        return f(f(f(a[0] * b[0]) + f(a[1] * b[1])) + f(a[2] * b[2]))

    def cross(a, b):
        return np.array([f(a[1] * b[2]) - f(a[2] * b[1]), f(a[2] * b[0]) - f(a[0] * b[2]), f(a[0] * b[1]) - f(a[1] * b[0])], f)

    def normalize(v):
        return (v * f(f(1.0) / np.sqrt(dot(v, v), dtype=f))).astype(f)

    W = (lookat - eye).astype(f)
    wlen = np.sqrt(dot(W, W), dtype=f)
    U = normalize(cross(W, up))
    V = normalize(cross(U, W))
    ang = f(f(f(0.5) * f(fovY)) * f(3.14159265358979323846)) / f(180.0)
    vlen = f(wlen * f(np.tan(ang, dtype=f)))
    V = (V * vlen).astype(f)
    U = (U * f(vlen * f(aspect))).astype(f)
    return U, V, W
