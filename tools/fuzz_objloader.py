#!/usr/bin/env python3
"""tools/fuzz_objloader.py N — mutation fuzz of the native OBJ / MTL parser (pt_load_obj, csrc/pt_objload.cpp): the committed fixtures with random
byte flips, deleted and duplicated lines, truncations, huge / negative / zero indices, NUL bytes, lone CRs and unterminated last lines.  The parser
may refuse a file (PT_ERR_INVALID) but must never crash, hang or read out of bounds — run it under the sanitizer build (tools/sanitize.sh, PT_OBJ_LIB).
Where both loaders accept a file, the native arrays must equal the Python restatement's."""
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optixpathtracer_amd import objloader  # noqa: E402

FIX = os.path.join(ROOT, "tests", "golden", "obj_fixture")


def mutate(data: bytes, rng) -> bytes:
    b = bytearray(data)
    for _ in range(int(rng.integers(1, 6))):
        kind = int(rng.integers(0, 8))
        if not b:
            break
        if kind == 0:  # flip bytes
            for _ in range(int(rng.integers(1, 8))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif kind == 1:  # truncate
            del b[int(rng.integers(0, len(b))):]
        elif kind == 2:  # delete a line
            lines = bytes(b).split(b"\n")
            if len(lines) > 1:
                del lines[int(rng.integers(0, len(lines)))]
            b = bytearray(b"\n".join(lines))
        elif kind == 3:  # duplicate a line somewhere else
            lines = bytes(b).split(b"\n")
            lines.insert(int(rng.integers(0, len(lines) + 1)), lines[int(rng.integers(0, len(lines)))])
            b = bytearray(b"\n".join(lines))
        elif kind == 4:  # hostile face lines
            pool = [b"f 0 1 2", b"f -99999 1 2", b"f 1 2 99999999", b"f 1/2/3/4 2 3", b"f 1//", b"f / / /", b"f 2147483647 2147483648 -2147483649", b"f 1 2",
                    b"f " + b" ".join(b"%d" % k for k in range(1, 70)), b"f 1/-1/-1 2/0/1 3//0", b"usemtl", b"mtllib", b"mtllib  a  b ", b"v 1e999 -1e999 nan", b"v .", b"v 1e", b"vt", b"g", b"o"]
            lines = bytes(b).split(b"\n")
            lines.insert(int(rng.integers(0, len(lines) + 1)), pool[int(rng.integers(0, len(pool)))])
            b = bytearray(b"\n".join(lines))
        elif kind == 5:  # NUL bytes and lone CRs
            for _ in range(int(rng.integers(1, 4))):
                b.insert(int(rng.integers(0, len(b) + 1)), int(rng.choice([0, 13])))
        elif kind == 6:  # digits into long numbers
            pos = int(rng.integers(0, len(b) + 1))
            b[pos:pos] = bytes(rng.integers(48, 58, int(rng.integers(1, 400))).astype(np.uint8))
        else:  # swap two chunks
            i, j = sorted(int(x) for x in rng.integers(0, len(b) + 1, 2))
            b = b[j:] + b[i:j] + b[:i]
    return bytes(b)


def same(a, b):
    if len(a.meshes) != len(b.meshes):
        return False
    for x, y in zip(a.meshes, b.meshes):
        if x.vertex.tobytes() != y.vertex.tobytes() or x.index.tobytes() != y.index.tobytes() or np.array(x.material).tobytes() != np.array(y.material).tobytes():
            return False
        if (x.normal is None) != (y.normal is None) or (x.normal is not None and x.normal.tobytes() != y.normal.tobytes()):
            return False
        if (x.texcoord is None) != (y.texcoord is None) or (x.texcoord is not None and x.texcoord.tobytes() != y.texcoord.tobytes()):
            return False
    return True


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(20261005)
    names = [f[:-4] for f in sorted(os.listdir(FIX)) if f.endswith(".obj")]
    accepted = refused = compared = 0
    d = tempfile.mkdtemp(prefix="ptamd_fuzz_")
    try:
        shutil.copytree(os.path.join(FIX, "tex"), os.path.join(d, "tex"))
        for it in range(n):
            name = names[int(rng.integers(0, len(names)))]
            obj = open(os.path.join(FIX, name + ".obj"), "rb").read()
            mtl = open(os.path.join(FIX, name + ".mtl"), "rb").read()
            which = int(rng.integers(0, 3))
            if which != 1:
                obj = mutate(obj, rng)
            if which != 0:
                mtl = mutate(mtl, rng)
            open(os.path.join(d, name + ".obj"), "wb").write(obj)
            open(os.path.join(d, name + ".mtl"), "wb").write(mtl)
            path = os.path.join(d, name + ".obj")
            try:
                a = objloader.load_obj(path, native=True)
                accepted += 1
            except (RuntimeError, ValueError):
                refused += 1
                continue
            if it % 4 == 0:  # the slow restatement on a quarter of the accepted files
                try:
                    b = objloader.load_obj(path, native=False)
                except Exception:
                    continue  # the restatement may give up where the C semantics carry on (numbers beyond double, NULs in names)
                if not same(a, b):
                    shutil.copy(path, "/tmp/fuzz_mismatch.obj")
                    shutil.copy(os.path.join(d, name + ".mtl"), "/tmp/fuzz_mismatch.mtl")
                    raise SystemExit(f"iteration {it} ({name}): native and Python loaders disagree; inputs kept as /tmp/fuzz_mismatch.*")
                compared += 1
    finally:
        shutil.rmtree(d, ignore_errors=True)
    print(f"[fuzz] {n} mutated OBJ / MTL sets: {accepted} parsed, {refused} refused, {compared} compared equal with the Python restatement; no crash")


if __name__ == "__main__":
    main()
