"""Probe: trace-kernel time of the 70k-triangle mesh scene for the default and a slightly yawed camera at two sizes."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import raytracer_amd as rt
from raytracer_amd import scenes
scene = rt.Scene()
mats = scenes.default_materials()
v,t,n9,tm,mm = scenes.displaced_grid(187, mat_offset=len(mats))
scene.set_triangles(v,t,n9,tm,nthreads=16); scene.materials = np.concatenate([mats,mm])
r = rt.Renderer(0); r.download=False
st = r.GetSettings(); st.SceneMode, st.Bounces = 0, 8
for (w,h) in ((960,540),(1920,1080),(3840,2160)):
  r.OnResize(w,h)
  for name, fwd in (('default', None), ('yawed', np.array([0.0123, -0.1, -1.0], np.float32))):
    cam = rt.Camera(70.0,0.05,100.0); cam.OnResize(w,h)
    if fwd is not None:
        fwd = fwd/np.linalg.norm(fwd); cam.SetPose(cam.GetPosition(), fwd.astype(np.float32))
    r.set_stats(True); r.reset_counters(); r.ResetFrameIndex(); r.Render(scene, cam); c = r.read_counters()
    r.set_stats(False); r.kernel_time(reset=True); r.ResetFrameIndex()
    t0=time.perf_counter()
    for i in range(8): r.Render(scene, cam)
    r.sync(); dt=(time.perf_counter()-t0)/8; kt = r.kernel_time()
    print(w,h,name, 'rays', c.rays, 'nodes/ray %.1f tris/ray %.1f'%(c.nodes_visited/c.rays, c.tris_tested/c.rays), 'trace kernel ms %.3f frame ms %.3f'%(kt.total_ms/kt.launches, dt*1e3), 'Mrays/s %.1f'%(c.rays/dt/1e6))
