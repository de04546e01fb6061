"""Mesh file readers that replace the file-loading half of mitsuba.load_dict (bake_shading.py:46-61). CPU only."""
import struct

import numpy as np


def _tet():
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    f = np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]], np.int32)
    return v, f


def test_obj_reader(tmp_path):
    from iris_amd.utils.path_tracing import load_mesh
    v, f = _tet()
    p = tmp_path / "scene.obj"
    with open(p, "w") as fh:
        fh.write("# comment\n")
        for a in v:
            fh.write("v {} {} {}\n".format(*a))
        fh.write("vn 0 0 1\nvt 0 0\n")
        fh.write("f 1/1/1 3/1/1 2/1/1\nf 1//1 2//1 4//1\nf 1 4 3\n")
        fh.write("f -3 -2 -1\n")                      # negative (relative) indices
        fh.write("f 1 2 3 4\n")                       # quad -> fan of two triangles
    vv, ff = load_mesh(str(p))
    np.testing.assert_array_equal(vv, v)
    np.testing.assert_array_equal(ff[:4], f)
    np.testing.assert_array_equal(ff[4:], [[0, 1, 2], [0, 2, 3]])


def test_ply_readers(tmp_path):
    from iris_amd.utils.path_tracing import load_mesh
    v, f = _tet()
    hdr = "ply\nformat {} 1.0\ncomment made by test\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\n" \
          "element face 4\nproperty list uchar int vertex_indices\nend_header\n"
    pa = tmp_path / "a.ply"
    with open(pa, "w") as fh:
        fh.write(hdr.format("ascii"))
        for a in v:
            fh.write("{} {} {} 7\n".format(*a))
        for t in f:
            fh.write("3 {} {} {}\n".format(*t))
    pb = tmp_path / "b.ply"
    with open(pb, "wb") as fh:
        fh.write(hdr.format("binary_little_endian").encode())
        for a in v:
            fh.write(struct.pack("<fffB", *a, 7))
        for t in f:
            fh.write(struct.pack("<Biii", 3, *t))
    for p in (pa, pb):
        vv, ff = load_mesh(str(p))
        np.testing.assert_array_equal(vv, v)
        np.testing.assert_array_equal(ff, f)
