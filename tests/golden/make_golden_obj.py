#!/usr/bin/env python3
"""Generates ref_obj_polygons.npz FROM THE REFERENCE'S OWN OBJ LOADER (oracle/_ref/libref_obj.so: the unmodified
RayTracer/src/tiny_obj_loader.h compiled as the application compiles it, oracle/Makefile): what it makes of polygons.obj --
per triangle the vertex indices, the normal indices and the material id, in its order.

Run in the build container, where /root/reference exists:   python tests/golden/make_golden_obj.py
The fixture is data only; the loader itself does not travel to the GPU box.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "libref_obj.so")


def main():
    assert os.path.exists(LIB), "build oracle/_ref first (make -C oracle ref)"
    L = C.CDLL(LIB)
    L.ref_obj_triangles.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ref_obj_triangles.restype = C.c_int
    cap = 4096
    v = np.zeros((cap, 3), np.int32); n = np.zeros((cap, 3), np.int32); m = np.zeros(cap, np.int32)
    count = L.ref_obj_triangles(os.path.join(HERE, "polygons.obj").encode(), cap, v.ctypes.data_as(C.c_void_p),
                                n.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p))
    assert 0 < count <= cap, count
    np.savez_compressed(os.path.join(HERE, "ref_obj_polygons.npz"), v_idx=v[:count], n_idx=n[:count], mat=m[:count])
    print("%d triangles from the reference's loader:" % count)
    print(v[:count].tolist())


if __name__ == "__main__":
    main()
