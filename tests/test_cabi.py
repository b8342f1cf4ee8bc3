"""The drop-in boundary without a GPU: libptamd.so loads, exports every symbol include/pt_amd.h declares,
struct layouts match the reference's, and the host-side entry points (BuildCDF, UVWFrame — host code in
the reference too) agree with the checker and with the reference-header golden vectors."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, assert_bits_equal
from optixpathtracer_amd import _lib, scenes


def _header_functions():
    src = open(os.path.join(ROOT, "include", "pt_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.load_library()
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"libptamd.so lacks {n}"
    assert set(names) == set(_lib.EXPORTS)
    assert b"gfx950" in L.pt_version()


def test_no_torch_types_in_abi_and_layouts():
    src = open(os.path.join(ROOT, "include", "pt_amd.h")).read()
    assert "torch" not in src and "at::" not in src and "std::" not in src
    assert C.sizeof(_lib.Material) == 104 == scenes.MATERIAL_DTYPE.itemsize
    offs = {n: getattr(_lib.Material, n).offset for n, _ in _lib.Material._fields_}
    for name in scenes.MATERIAL_DTYPE.names:
        assert offs[name] == scenes.MATERIAL_DTYPE.fields[name][1], name


def test_build_cdf_host_entry(orc_det):
    for probe in (scenes.disc_probe(), scenes.sky_probe(128, 64), scenes.constant_probe()):
        a = _lib.build_cdf(probe.data, probe.width, probe.height)
        b = orc_det.build_cdf(probe.data, probe.width, probe.height)
        c = scenes.ProbeData(probe.width, probe.height, probe.data).BuildCDF()
        for x, y, z in zip(a, b, (c.pdfValuesX, c.cdfValuesX, c.pdfValuesY, c.cdfValuesY)):
            assert_bits_equal(x, y, "pt_build_cdf vs checker")
            assert_bits_equal(x, z, "pt_build_cdf vs numpy mirror")
    assert _lib.load_library().pt_build_cdf(None, 4, 4, None, None, None, None) != 0


def test_uvw_frame_host_entry_against_reference_golden():
    from optixpathtracer_amd.renderer import Camera

    G = np.load(os.path.join(ROOT, "tests", "golden", "ref_tables.npz"))
    for row in G["cam_table"]:
        U, V, W = Camera(tuple(row[0:3]), tuple(row[3:6]), tuple(row[6:9]), float(row[9]), float(row[10])).UVWFrame()
        got = np.concatenate([U, V, W])
        assert np.abs(got - row[11:20]).max() <= 2 * np.spacing(np.abs(row[11:20]).max())


def test_scene_generators_match_baseline_configs():
    m = scenes.cornell_box()
    assert m.num_triangles == 32 and len(m.meshes) == 4
    v, idx, tm, mats = m.flatten()
    assert v.dtype == np.float32 and idx.dtype == np.uint32 and idx.max() < len(v)
    assert np.allclose(v.min(0), (0, 0, 0)) and np.allclose(v.max(0), (556, 548.8, 559.2))
    t = scenes.voxel_terrain()
    assert t.num_triangles == 1_000_000 and len(t.meshes) == 8
    t2 = scenes.voxel_terrain()
    assert all(np.array_equal(a.vertex, b.vertex) for a, b in zip(t.meshes, t2.meshes))  # deterministic
    b = scenes.two_box_scene()
    assert b.num_triangles == 24 and b.meshes[1].material["flags"] == 1 and b.meshes[0].material["flags"] == 0
    p = scenes.sky_probe(512, 256)
    assert (p.data[..., 0] == 50).sum() > 10  # the sun disc exists


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "optixpathtracer_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                s = open(os.path.join(dirpath, f), errors="ignore").read()
                for pat in ("import oracle", "from oracle", "oracle/", "liborc", "orc_"):
                    assert pat not in s, f"{f} references the checker ({pat})"


def test_headers_compile_as_c99_and_cxx17(tmp_path):
    """include/pt_amd.h is a plain C header; the C++ facade over it (the reference's SampleRenderer surface) compiles
    with a host compiler alone — no hipcc, no torch types on the application side."""
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not (shutil.which("gcc") and shutil.which("g++")):
        pytest.skip("no host compiler")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(root, "include", "pt_amd.h")], check=True)
    src = tmp_path / "facade.cpp"
    src.write_text('#include "optixpathtracer_amd/csrc/SampleRenderer.h"\nint main() { return 0; }\n')
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", root, "-I", os.path.join(root, "include"), str(src)], check=True)


def test_bench_source_hash_ignores_comments_only():
    """PMC-derived numbers in the bench line are gated on bench.source_hash(): it must not move when a comment is edited and must
    move when code does; says (as a skip) when the committed PMC files were measured on other kernel sources."""
    import json

    import bench

    a = bench._strip_comments('int x = 1; // note\n/* block\n comment */ float y = 2.f;\n\nconst char* s = "// not a comment";\n')
    b = bench._strip_comments('int x = 1;\n float y = 2.f;\nconst char* s = "// not a comment";\n')
    assert a == b
    assert bench._strip_comments("int x = 1;") != bench._strip_comments("int x = 2;")
    import glob

    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))[-1]  # the file bench.py reads
    for path in (newest, os.path.join(ROOT, "profiles", "stadium_" + os.path.basename(newest))):
        pmc = json.load(open(path))
        if pmc["src_hash"] != bench.source_hash():  # not an error of the code: bench.py then simply does not quote the PMC-derived numbers
            pytest.skip(f"{os.path.relpath(path, ROOT)} was measured on other kernel sources (re-run tools/r4_final.sh to quote its numbers)")


def test_reference_main_cpp_statements_compile_against_the_facade(tmp_path):
    """What the reference's application does with its renderer (HelloPathtracing_original/main.cpp:131-144 initLaunchParams, :211-218
    construction / setCamera / resize / setProbe, :245 output_buffer.setStream(sample.stream), :259 and :286 subframe_index, :262 resize,
    :273 sample.render(output_buffer)), restated statement by statement against csrc/SampleRenderer.h: it must compile as it stands.  The
    output buffer is a stand-in for sutil::CUDAOutputBuffer<uint32_t> with the three members the renderer and main.cpp touch."""
    import shutil
    import subprocess

    if not shutil.which("g++"):
        pytest.skip("no host compiler")
    src = tmp_path / "main_snippet.cpp"
    src.write_text(r'''
#include "optixpathtracer_amd/csrc/SampleRenderer.h"
using namespace ptamd;
struct OutputBuffer {                      // sutil::CUDAOutputBuffer<uint32_t>: map() yields a device pointer (sutil/CUDAOutputBuffer.h)
    uint32_t* map() { return d; }
    void unmap() {}
    void setStream(stream_t s) { stream = s; }
    void resize(int, int) {}
    uint32_t* d = nullptr; stream_t stream = nullptr;
};
void initLaunchParams(SampleRenderer& pathtracer) {   // main.cpp:131-144
    LaunchParams& params = pathtracer.launchParams;
    params.samples_per_launch = 32;
    params.frame.subframe_index = 0u;
    const float light_size = 200.f;
    params.light.emission = make_float3(15.0f, 15.0f, 15.0f);
    params.light.corner = make_float3(-1000 - light_size, 1200, -light_size);
    params.light.v1 = make_float3(2.f * light_size, 0, 0);
    params.light.v2 = make_float3(0, 0, 2.f * light_size);
    params.light.normal = normalize(cross(params.light.v1, params.light.v2));
}
int run(const Model* model, const Camera& camera, const ProbeData& probe) {
    SampleRenderer sample(model);          // main.cpp:211
    sample.setCamera(camera);              // :212
    int2 fbSize = make_int2(1200, 1024);   // :214
    sample.resize(fbSize);                 // :215
    initLaunchParams(sample);              // :217
    sample.setProbe(probe);                // :218
    OutputBuffer output_buffer;
    output_buffer.setStream(sample.stream);             // :245
    bool camera_changed = true, resize_dirty = true;
    if (camera_changed || resize_dirty) sample.launchParams.frame.subframe_index = 0;  // :258-259
    if (resize_dirty) { sample.resize(fbSize); output_buffer.resize(fbSize.x, fbSize.y); resize_dirty = false; }  // :261-265
    sample.render(output_buffer);                       // :273
    sample.launchParams.frame.subframe_index += 1;      // :286
    static_assert(sizeof(ParallelogramLight) == 60, "LaunchParams.h:32-38: five float3");
    return sample.denoiserOn ? 0 : 1;                   // SimplePathtracer.h:63: defaults to true
}
int main() { return 0; }
''')
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", ROOT, "-I", os.path.join(ROOT, "include"), str(src)], check=True)


def test_stream_contract_is_documented():
    src = open(os.path.join(ROOT, "include", "pt_amd.h")).read()
    assert "STREAM CONTRACT" in src and "hipStreamNonBlocking" in src and "pt_wait_event" in src and "VERSIONING" in src
    L = _lib.load_library()
    assert L.pt_stats_size() == C.sizeof(_lib.Stats)
    assert b"ptamd 0.4" in L.pt_version()
