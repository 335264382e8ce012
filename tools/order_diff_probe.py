"""Probe: pixels of config 4 whose primary id differs between the reference walk order and front to back; the oracle's
answer for exactly those rays."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
import bench
from oracle import orc

cfg = bench.CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 4]
scene = bench.build_scene(cfg)
w, h = cfg["w"], cfg["h"]
cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
r = rt.Renderer(0); r.download = False
st = r.GetSettings(); st.SceneMode, st.Bounces = 0, 1
res = {}
for order in (1, 2):
    st.TraversalOrder = order
    r.OnResize(w, h); r.ResetFrameIndex(); r.record_ids(True)
    r.Render(scene, cam)
    res[order] = r.read_primary_ids().ravel().copy()
diff = np.flatnonzero(res[1] != res[2])
print("pixels:", w * h, "differ:", len(diff))
if len(diff):
    osc = orc.Scene()
    osc.set_spheres(scene.spheres); osc.set_materials(scene.materials)
    osc.set_mesh(scene.verts, scene.tri_idx, scene.normals9, scene.tri_mat)
    osc.import_kd(scene.kd_tree.arrays())
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h).reshape(-1, 3)
    sel = diff[:40]
    ids, t, u, v, cnt = osc.trace_primary(0, cam.GetPosition(), dirs[sel])
    for k, px in enumerate(sel):
        print("pixel (%d,%d) ref-order %x  front-to-back %x  oracle %x t=%r u=%r v=%r dir=%s" % (
            px % w, px // w, int(res[1][px]) & 0xFFFFFFFF, int(res[2][px]) & 0xFFFFFFFF, int(ids[k]) & 0xFFFFFFFF, t[k], u[k], v[k], dirs[px]))
