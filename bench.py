#!/usr/bin/env python3
"""Benchmark of the tracing path: Mrays/s (primary + bounce) and ms/frame.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|1|3|4|5] [--no-other-configs] [--no-cpu-baseline]

A step is one Renderer::Render (one 1-spp frame, one launch) of the configured scene.  Default workload = BASELINE.json
configs[1]: the reference application's default sphere scene at 1920x1080, 5 bounces, 64 accumulated frames.  With N > 1
(launched under torch.distributed.run, one rank per GPU) the frames are sample-split: every rank renders K frames of its own
frame indices, the accumulation buffers are summed once over RCCL onto rank 0 at the end of the timed region and rank 0 resolves
the image; value = rays traced by all ranks / max-over-ranks time (weak scaling).

Rank 0 prints ONE JSON line.  `roofline` prices the trace kernel (device time from HIP events on its launch stream) against
the bound that applies to the workload; `cpu_baseline` times the CPU oracle on a bounded sample of the same workload on this
box's host cores.  Beside the one-Render-per-launch figures the line carries
  * `throughput_mode`: the same frames as launches of 16 Renders (rt_render_frames), what a progressive accumulation uses;
  * `with_d2h`: ms per Render including the hand-off of the RGBA32F frame to host memory (rt_map_image, and the pipelined
    variant that copies frame k beside the trace of frame k + 1) -- what the drop-in Renderer::Render costs;
  * `other_configs` (one-GPU default run only): the tree configurations 3, 4 and 6 (Cornell-box-class OBJ), measured the same way in the same process.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# Algorithmic work per ray / per pixel-frame:
FLOPS_PER_SPHERE_RAY = lambda n_spheres: 6 + 24 * n_spheres + 120   # intersection + closest-hit/shade/scatter/roulette
FLOPS_PER_PIXEL_RESOLVE = 40
BYTES_PER_PIXEL_FRAME = 40          # accumulation read 12 + write 12 + RGBA32F write 16
BYTES_PER_NODE, BYTES_PER_TRI, BYTES_PER_HIT = 32, 40, 40
# fp32 VALU ceiling without FMA: the chip's 157.3 TFLOP/s vector peak counts an FMA as two operations, and this path may not
# fuse (bit-exact contract), so one operation per lane per issue slot: 256 CU x 128 lanes/clk x 2.4 GHz.
PEAK_VALU_NOFMA_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0
FRAMES_PER_LAUNCH = 16              # throughput mode (= RT_MAX_LAUNCH_FRAMES of tree scenes)
ORDERS = {"auto": 0, "reference": 1, "front-to-back": 2}

CONFIGS = {
    1: dict(name="default sphere scene 1280x720, 5 bounces (BASELINE configs[0])", w=1280, h=720, bounces=5, mode=1, mesh=None),
    2: dict(name="default sphere scene 1920x1080, 5 bounces, 64 spp (BASELINE configs[1])", w=1920, h=1080, bounces=5, mode=1, mesh=None),
    3: dict(name="69,938-triangle displaced grid + KD tree, 1920x1080, 8 bounces (BASELINE configs[2])", w=1920, h=1080,
            bounces=8, mode=0, mesh=("grid", 187)),
    # 2236 x 2236 quads over 30 x 30 units: quads of 13 mm.  (Over the 3 x 3 units of config 3 every triangle would fall under
    # the reference's |determinant| < 1e-5 rejection, Intersections.cpp:64, and the mesh could not be hit at all.)
    4: dict(name="9,999,392-triangle displaced grid (30x30 units) + KD tree in HBM, 3840x2160, 8 bounces, screen-tile split (BASELINE configs[3])",
            w=3840, h=2160, bounces=8, mode=0, mesh=("grid", 2236, 10.0), split="tile"),
    # The reference application's own kind of scene (RaytracerApp.cpp:27-131 loads a Cornell-box OBJ): 32 wall-sized triangles and a
    # 5,000-triangle water surface in the transparent material "water", through the OBJ ingest.  Not a BASELINE configuration.
    6: dict(name="Cornell-box-class OBJ (32 large triangles + 5,000-triangle water surface, material \"water\"), 1920x1080, 8 bounces",
            w=1920, h=1080, bounces=8, mode=0, mesh=("cornell", 50)),
    5: dict(name="default spheres + 9,999,392-triangle mesh, 3840x2160, 8 bounces, progressive, sample split (BASELINE configs[4])",
            w=3840, h=2160, bounces=8, mode=2, mesh=("grid", 2236, 10.0)),
}


KERNEL_TIME_SOURCE = ("HIP events on the launch stream; sphere scenes: around every n-th single-Render launch of the timed region (n = steps / 4, "
                      "at most 8; launches n / 2, n / 2 + n, ...), tree scenes and launches of 16 Renders: around every launch.  The two event records of a launch cost a "
                      "1080p sphere frame about 8 us, so timing every launch slows what is being timed; a timed launch stands alone between "
                      "its two markers and its time includes its own dispatch, while an untimed launch is set up behind the running one, so "
                      "ms_per_step can come out slightly below kernel_avg_ms")


def timing_every(steps, per_launch=1, mode=1):
    """Which launches get HIP events (rt_set_kernel_timing): see KERNEL_TIME_SOURCE.  Tree scenes: every launch (their frames
    differ by +-10 % in kernel time, a sample of them would not give the mean; the events cost them 2 % at most)."""
    return 1 if per_launch > 1 or mode != 1 else max(1, min(8, steps // 4))


def host_threads():
    """Threads for host-side work: this process's CPU share (a GPU box shows the whole machine's CPUs to os.cpu_count())."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for var in ("OMP_NUM_THREADS", "MAX_JOBS"):
        v = os.environ.get(var, "")
        if v.isdigit() and int(v) >= 1:
            n = min(n, int(v))
    return max(1, n)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def settle_gc():
    """After a scene is built: everything alive moves out of the collector's sight.  A full collection over the host copy of a
    10 M-triangle scene takes ~50 ms, and the interpreter starts one whenever enough objects have been allocated -- seen as a
    Render of 50 instead of 3 ms once in about two hundred."""
    gc.collect()
    gc.freeze()


def build_scene(cfg):
    import raytracer_amd as rt
    from raytracer_amd import scenes
    scene = rt.Scene()
    scene.spheres = scenes.default_spheres()
    mats = scenes.default_materials()
    if cfg["mesh"] and cfg["mesh"][0] == "cornell":
        # as the application does it: the file's materials first, its own seven after them (RaytracerApp.cpp:132-187)
        import tempfile
        from raytracer_amd import ingest
        with tempfile.TemporaryDirectory() as tmp:
            verts, tris, normals9, tri_mat, mats, names = ingest.load_obj(scenes.write_cornell_class_obj(tmp, cfg["mesh"][1]), app_materials=True)
        scene.spheres = scenes.default_spheres(mat_offset=len(names))
        scene.set_triangles(verts, tris, normals9, tri_mat, max_nodes=4_000_000, nthreads=min(16, host_threads()))
        scene.materials = mats
        return scene
    if cfg["mesh"]:
        kind, n = cfg["mesh"][:2]
        if kind == "grid":
            k = cfg["mesh"][2] if len(cfg["mesh"]) > 2 else 1.0
            verts, tris, normals9, tri_mat, mesh_mats = scenes.displaced_grid(n, x_range=(-1.5 * k, 1.5 * k), z_range=(-3.0 * k, 0.0),
                                                                              mat_offset=len(mats))
        else:
            verts, tris, normals9, tri_mat, mesh_mats = scenes.cube_sphere(n, mat_offset=len(mats))
        scene.set_triangles(verts, tris, normals9, tri_mat, nthreads=min(16, host_threads()))
        mats = np.concatenate([mats, mesh_mats])
    scene.materials = mats
    return scene


def cpu_baseline(cfg, scene, cam, budget_frames):
    """Time the oracle (CPU restatement, per-pixel-hash RNG mode) on a bounded sample of the workload: on all host threads of
    this process's CPU share, and on one thread over a quarter of the sample."""
    from oracle import orc
    w, h = cfg["w"], cfg["h"]
    threads = host_threads()
    oscene = orc.Scene()
    oscene.set_spheres(scene.spheres); oscene.set_materials(scene.materials)
    big = bool(cfg["mesh"] and cfg["mesh"][1] > 1000)
    row_step = 64 if big else 1
    if scene.kd_tree is not None:
        oscene.set_mesh(scene.verts, scene.tri_idx, scene.normals9, scene.tri_mat)
        oscene.import_kd(scene.kd_tree.arrays())
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
    st = orc.Settings(1, cfg["mode"], 1, 0, cfg["bounces"])

    def run(nthreads, frames, step):
        acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32)
        cnt = orc.Counters()
        t0 = time.perf_counter()
        for f in range(1, frames + 1):
            oscene.render_frame(st, cam.GetPosition(), dirs, w, h, f, acc, img, counters=cnt, nthreads=nthreads, row_step=step)
        return cnt.rays, time.perf_counter() - t0

    # how the restatement compares with the reference's own objects (measured where they can be built: the build container)
    port_vs_reference = None
    try:
        with open(os.path.join(ROOT, "profiles", "r3_reference_vs_oracle_cpu_timing.json")) as f:
            t = json.loads(f.read().strip().splitlines()[0])
        port_vs_reference = {"oracle_over_reference_time": t["oracle_over_reference_time"], "reference_Mrays_s_one_thread": t["reference_Mrays_s"],
                             "oracle_Mrays_s_one_thread": t["oracle_Mrays_s"], "rays": t["rays"], "same_hits_and_t_bits": t["same_hits_and_t_bits"],
                             "what": "KDTreeCPU::intersect of the reference's unmodified objects (oracle/_ref) against the oracle's restatement on the same "
                                     "primary rays of the 69,938-triangle mesh, one thread each, in the build container (tools/ref_vs_oracle_timing.py); "
                                     "committed, not measured in this run"}
    except (OSError, ValueError, KeyError, IndexError):
        pass
    rays, dt = run(threads, budget_frames, row_step)
    one_step = row_step * (16 if big else 8)                     # the one-thread leg sees an eighth (a sixteenth) of the rows of one frame
    rays1, dt1 = run(1, 1, one_step)
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port", "cpu": cpu_model(),
            "host_cpus_visible": os.cpu_count(),
            "sample": "%d frame(s) of the same %dx%d workload%s, oracle (CPU restatement, hash RNG), %.1f s wall"
                      % (budget_frames, w, h, "" if row_step == 1 else ", every %dth row" % row_step, dt),
            "ms_per_frame_extrapolated": dt / budget_frames * 1e3 * row_step, "port_vs_reference": port_vs_reference,
            "one_thread": {"value": rays1 / dt1 / 1e6, "unit": "Mrays/s", "cores": 1,
                           "sample": "1 frame, every %dth row, %.1f s wall" % (one_step, dt1),
                           "ms_per_frame_extrapolated": dt1 * 1e3 * one_step}}


def _traffic_from_profile(config, frames_per_launch=1):
    """HBM bytes per launch of the trace kernel from the committed PMC pass (profiles/traffic.json), or None."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    key = "config%d" % config + ("_x%d" % frames_per_launch if frames_per_launch > 1 else "")
    try:
        with open(p) as f:
            return json.load(f).get(key, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def _issue_from_profile(config, frames_per_launch=1):
    """Wave-instructions per launch of the trace kernel from the committed PMC pass (profiles/issue.json), or None."""
    p = os.path.join(ROOT, "profiles", "issue.json")
    key = "config%d" % config + ("_x%d" % frames_per_launch if frames_per_launch > 1 else "")
    try:
        with open(p) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


SIMDS = 1024                         # 256 CUs x 4
ISSUE_CYCLES = 2.0                   # a SIMD starts a wave-instruction every other cycle at best (32 lanes per cycle, 64-lane waves)


def issue_roofline(config, frames_per_launch, kernel_s, clock_hz):
    """The bound these kernels run against: the rate at which the chip's SIMDs can start wave-instructions of ANY kind (vector,
    scalar, memory, LDS: a wave issues in order).  Instructions per launch from the committed PMC pass, time from this run."""
    rec = _issue_from_profile(config, frames_per_launch)
    if not rec or not rec.get("instructions") or not kernel_s:
        return None
    peak = SIMDS * clock_hz / ISSUE_CYCLES
    achieved = rec["instructions"] / kernel_s
    return {"bound": "instruction issue", "achieved": achieved / 1e12, "peak": peak / 1e12, "unit": "T wave-instructions/s", "frac": achieved / peak,
            "instructions_per_launch": rec["instructions"], "valu": rec.get("valu"), "salu": rec.get("salu"), "vmem": rec.get("vmem"),
            "lds": rec.get("lds"), "smem": rec.get("smem"), "valu_lane_use": rec.get("valu_lane_use"),
            "wave_cycles_waiting": rec.get("wave_cycles_waiting"), "clock_hz": clock_hz,
            "peak_note": "%d SIMDs x clock / %.0f cycles per wave-instruction" % (SIMDS, ISSUE_CYCLES),
            "source": rec.get("source", "") + "; committed, not measured in this run"}


def chip_busy(r, frame=1):
    """Share of the chip's wave slots that held a running wave between the first wave's start and the last wave's end of one
    Render (per-wave clocks of the timed kernel, one untimed launch): what is left of 1 is waves that ended early."""
    r.set_stats(2)
    try:
        r.render_only(frame)
        wc = r.read_wave_clock()
    finally:
        r.set_stats(False)
    if not len(wc):
        return None
    start, end = wc[:, 0].astype(np.float64), wc[:, 1].astype(np.float64)
    span = end.max() - start.min()
    return {"fraction": float((end - start).sum() / (len(wc) * span)) if span > 0 else None, "waves": int(len(wc)),
            "span_us": float(span / 100.0), "waiting_without_a_path": float(wc[:, 6].astype(np.float64).sum() / max((end - start).sum(), 1.0)),
            "note": "sum over waves of (end - start) / (waves x (last end - first start)), 100 MHz clock, one Render"}


TRAFFIC_SOURCE = "profiles/traffic.json: FETCH_SIZE + WRITE_SIZE of a rocprofv3 --pmc pass over this command, committed; not measured in this run"


def roofline(config_id, cfg, scene, r, st, order_name, frames, cnt, kt, world, profiling=False, clock_hz=2.4e9):
    """Price the trace kernel of the timed launches (cnt / kt: their counters and event times) against its bound."""
    px = cfg["w"] * cfg["h"]
    timed = max(kt.launches, 1)                                      # launches with HIP events around them (KERNEL_TIMING_EVERY)
    avg_kernel_s = kt.total_ms / timed / 1e3
    frames_per_launch = max(1, round(kt.frames / timed))             # 1, or 16 under --profile throughput
    launches = max(1, round(len(frames) / frames_per_launch))        # all launches of the timed region
    rays_per_launch = cnt.rays / launches
    px = px * frames_per_launch                                      # pixel-frames per launch
    traffic = _traffic_from_profile(config_id, frames_per_launch)
    if cfg["mode"] == 1:
        flops = rays_per_launch * FLOPS_PER_SPHERE_RAY(len(scene.spheres)) + px * FLOPS_PER_PIXEL_RESOLVE
        achieved = flops / avg_kernel_s / 1e12
        return {"bound": "valu", "achieved": achieved, "peak": PEAK_VALU_NOFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_VALU_NOFMA_TFLOPS, "traffic": traffic, "traffic_source": TRAFFIC_SOURCE,
                "hbm_algorithmic_GBs": px * BYTES_PER_PIXEL_FRAME / avg_kernel_s / 1e9,
                "hbm_moved_GBs": (traffic / avg_kernel_s / 1e9) if traffic else None,
                "kernel": "rt_trace_spheres_kernel", "kernel_avg_ms": avg_kernel_s * 1e3, "launches": launches,
                "launches_timed": int(kt.launches), "kernel_time_source": KERNEL_TIME_SOURCE,
                "flops_per_ray": FLOPS_PER_SPHERE_RAY(len(scene.spheres)), "rays_per_launch": rays_per_launch,
                "issue": issue_roofline(config_id, frames_per_launch, avg_kernel_s, clock_hz),
                "chip_busy": chip_busy(r) if not profiling else None,
                "note": "fp32 VALU without FMA (bit-exact contract); no MFMA in this path; accum and dirs are float4 "
                        "(32 B + 16 B per pixel against the 24 B + 0 B of the 40 B/pixel model), hence traffic about twice the algorithmic bytes"}
    # node / triangle counts of the same frames, from an untimed pass with the counting kernel
    r.set_stats(True); r.reset_accum(); r.reset_counters()
    for f in frames:
        r.render_only(f)
    c2 = r.read_counters()
    order_used = "front-to-back" if r.traversal_in_use() == 2 else "reference"
    ref_counts = None
    if order_used != "reference":                # what the reference's own walk order costs on the same rays
        st.TraversalOrder = 1; r.push_settings()
        r.reset_accum(); r.reset_counters()
        for f in frames[:4]:
            r.render_only(f)
        c3 = r.read_counters()
        ref_counts = {"nodes_per_ray": (c3.nodes_visited - c3.prepass_nodes) / max(c3.rays, 1),
                      "tris_per_ray": (c3.tris_tested - c3.prepass_tris) / max(c3.rays, 1)}
        st.TraversalOrder = ORDERS[order_name]; r.push_settings()
    r.set_stats(False)
    # grazing ties: primary ids that differ between the reference walk order and front to back (one frame each, untimed)
    order_ties = None
    if order_used != "reference" and world == 1 and not profiling:   # (a profiler pass keeps to launches of one kind)
        ids = {}
        for o in (1, 2):
            st.TraversalOrder = o; r.push_settings(); r.record_ids(True)
            r.render_only(1); ids[o] = r.read_primary_ids()
        r.record_ids(False)
        st.TraversalOrder = ORDERS[order_name]; r.push_settings()
        order_ties = {"primary_ids_differing_from_reference_order": int((ids[1] != ids[2]).sum()), "pixels": int(cfg["w"] * cfg["h"])}
    per_launch = lambda v: v / launches              # the counting pass rendered the same frames once each
    per_frame = lambda v: v / max(len(frames), 1)
    # what the trace kernel itself walked: the counters minus the part that stands for the column rays (primary rays with
    # direction.x == +0), which the pre-pass resolves without walking the tree ray by ray
    own_nodes = per_launch(c2.nodes_visited - c2.prepass_nodes)
    own_tris = per_launch(c2.tris_tested - c2.prepass_tris)
    # (the counting pass traces every ray; primary rays of empty work units are not traced by the timed kernel: their one node
    # and their hit record are taken out again)
    # ... and so are the first steps of primary walks that start at their unit's entry node)
    skipped = cnt.empty_unit_rays / launches
    spared_nodes, spared_tris = cnt.entry_nodes_spared / launches, cnt.entry_tris_spared / launches
    bytes_ = ((own_nodes - skipped - spared_nodes) * BYTES_PER_NODE + (own_tris - spared_tris) * BYTES_PER_TRI
              + (rays_per_launch - skipped) * BYTES_PER_HIT + px * BYTES_PER_PIXEL_FRAME)
    achieved = bytes_ / avg_kernel_s / 1e9
    small = cfg["mesh"][1] < 1000
    return {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
            "traffic": traffic, "traffic_source": TRAFFIC_SOURCE,
            "hbm_moved_GBs": (traffic / avg_kernel_s / 1e9) if traffic else None,
            "frac_hbm_moved": (traffic / avg_kernel_s / 1e9 / PEAK_HBM_GBS) if traffic else None,
            "algorithmic_bytes_per_launch": bytes_,
            "kernel": "rt_trace_kernel<%s, %s>" % ("mesh" if cfg["mode"] == 0 else "spheres+mesh", order_used),
            "kernel_avg_ms": avg_kernel_s * 1e3, "launches": launches,
            "launches_timed": int(kt.launches), "kernel_time_source": KERNEL_TIME_SOURCE,
            "prepass_avg_ms": kt.prepass_total_ms / timed,
            "nodes_per_ray": (own_nodes - skipped - spared_nodes) / max(rays_per_launch, 1),
            "tris_per_ray": (own_tris - spared_tris) / max(rays_per_launch, 1),
            "column_rays_reference_equivalent": {"nodes_per_frame": per_frame(c2.prepass_nodes), "tris_per_frame": per_frame(c2.prepass_tris)},
            "rays_per_launch": rays_per_launch, "rays_not_traced_per_launch": skipped,
            "primary_walk_steps_spared_per_launch": {"nodes": spared_nodes, "tris": spared_tris}, "traversal": order_used, "reference_order_counts": ref_counts,
            "grazing_ties": order_ties,
            "near_tie_rewalks_per_frame": per_frame(cnt.near_ties),
            "issue": issue_roofline(config_id, frames_per_launch, avg_kernel_s, clock_hz),
            "chip_busy": chip_busy(r) if not profiling else None,
            "note": "algorithmic bytes = 32 B x nodes visited + 40 B x triangles tested by the trace kernel (+40 B per traced ray for the hit's "
                    "normals/material, +40 B per pixel for accumulation and image), counts from the device counters of the walk order "
                    "that was timed; primary rays of 16-pixel units that miss the root box altogether are not traced and not priced, nor are the first steps of primary walks that start at their unit's common entry node; `frac` prices them against the HBM peak as the contract asks, `frac_hbm_moved` is what the PMC "
                    "counters saw reach HBM; neither is the bound that applies: the walk is a chain of dependent 32-48 B reads made by waves that issue in order, "
                    "so `issue` (wave-instructions started per second against what the SIMDs can start) and `chip_busy` (waves that ended early) say how far the kernel is from its roof; "
                    + ("nodes + leaf records (a few MB to a few tens of MB) sit in L2 / Infinity Cache at this size, not in HBM" if small
                       else "nodes (8 x 0.5 GB) + leaf records (2.4 GB) are HBM-resident: random 32-48 B reads")}


def throughput_mode(r, cfg, torch, launches, rays_per_frame, bytes_per_frame=None, n_spheres=9, config_id=0, clock_hz=2.4e9):
    """The same workload as launches of FRAMES_PER_LAUNCH Renders (rt_render_frames): what a progressive accumulation uses."""
    r.set_kernel_timing(1)
    r.reset_accum(); r.render_only(1001, FRAMES_PER_LAUNCH)                     # warm: slot planes are allocated here
    r.sync(); torch.cuda.synchronize()
    r.reset_accum(); r.reset_counters(); r.kernel_time(reset=True)
    t0 = time.perf_counter()
    for i in range(launches):
        r.render_only(1 + i * FRAMES_PER_LAUNCH, FRAMES_PER_LAUNCH)
    r.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cnt, kt = r.read_counters(), r.kernel_time()
    frames = launches * FRAMES_PER_LAUNCH
    kernel_ms_per_frame = kt.total_ms / max(kt.frames, 1)
    out = {"frames_per_launch": FRAMES_PER_LAUNCH, "launches": launches, "ms_per_frame": dt / frames * 1e3,
           "value": cnt.rays / dt / 1e6, "unit": "Mrays/s", "kernel_ms_per_frame": kernel_ms_per_frame,
           "prepass_ms_per_frame": kt.prepass_total_ms / max(kt.frames, 1), "rays_per_frame": cnt.rays / frames}
    if cfg["mode"] == 1:
        flops = rays_per_frame * FLOPS_PER_SPHERE_RAY(n_spheres) + cfg["w"] * cfg["h"] * FLOPS_PER_PIXEL_RESOLVE
        out["roofline_frac"] = flops / (kernel_ms_per_frame / 1e3) / 1e12 / PEAK_VALU_NOFMA_TFLOPS
    elif bytes_per_frame:
        # (the algorithmic bytes of the walk are served mostly by L1 / L2 / Infinity Cache in launches of 16 Renders: priced against
        # the HBM peak they are no fraction of a bound -- what reaches HBM is)
        launch_s = kernel_ms_per_frame / 1e3 * FRAMES_PER_LAUNCH
        traffic = _traffic_from_profile(config_id, FRAMES_PER_LAUNCH)
        out["algorithmic_GBs"] = bytes_per_frame / (kernel_ms_per_frame / 1e3) / 1e9
        out["frac_hbm_moved"] = (traffic / launch_s / 1e9 / PEAK_HBM_GBS) if traffic else None
        out["traffic_per_launch"] = traffic
        out["traffic_source"] = TRAFFIC_SOURCE
    out["issue"] = issue_roofline(config_id, FRAMES_PER_LAUNCH, kernel_ms_per_frame / 1e3 * FRAMES_PER_LAUNCH, clock_hz)
    return out


def first_hit_reuse(r, st, cfg, torch, steps, launches):
    """Tree scenes with rt_settings.reuse_first_hits: the primary hit of every pixel is found once (the camera stands still
    during an accumulation) and each Render starts its paths from it -- same frames bit for bit, but not the reference's
    amount of work per Render, so it is reported beside the main figures and never in them.  `value` counts the reference's
    TraceRay invocations (as everywhere); `traced_value` only the rays whose hit was searched for in these launches."""
    st.ReuseFirstHits = True; r.push_settings()
    out = {"note": "primary hits kept per pixel while camera, directions and tree are unchanged (found by one extra launch, "
                   "outside the timed region like the other per-camera set-up); frames are bit-identical to those without it"}
    for name, per, n in (("one_render_per_launch", 1, steps), ("throughput_mode", FRAMES_PER_LAUNCH, launches)):
        r.set_kernel_timing(timing_every(n, per, cfg["mode"]))
        r.reset_accum(); r.render_only(1001, per); r.sync(); torch.cuda.synchronize()
        r.reset_accum(); r.reset_counters(); r.kernel_time(reset=True)
        t0 = time.perf_counter()
        for i in range(n):
            r.render_only(1 + i * per, per)
        r.sync(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        cnt, kt = r.read_counters(), r.kernel_time()
        frames = n * per
        out[name] = {"frames_per_launch": per, "launches": n, "ms_per_frame": dt / frames * 1e3, "value": cnt.rays / dt / 1e6,
                     "traced_value": (cnt.rays - cnt.primary_hits_reused - cnt.empty_unit_rays) / dt / 1e6, "unit": "Mrays/s",
                     "kernel_ms_per_frame": kt.total_ms / max(kt.frames, 1), "rays_per_frame": cnt.rays / frames,
                     "primary_hits_reused_per_frame": cnt.primary_hits_reused / frames,
                     "empty_unit_rays_per_frame": cnt.empty_unit_rays / frames}
    st.ReuseFirstHits = False; r.push_settings()
    return out


def with_d2h(r, torch, steps):
    """ms per Render including the frame's way to host memory, as the drop-in Renderer::Render ends (Renderer.cpp:134)."""
    out = {}
    r.sync(); r.set_stream(None)                     # the library's own streams: the copy stream then runs beside the render stream
    for name, pipelined in (("ms_per_step_map_image", False), ("ms_per_step_map_image_pipelined", True)):
        r.reset_accum()
        for f in range(1, 9):                        # (the first calls of either kind set up buffers: one of them takes 5 ms)
            r.render_only(f); r.map_image(pipelined)
        r.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(1, steps + 1):
            r.render_only(f); r.map_image(pipelined)
        r.sync(); torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / steps * 1e3
    out["note"] = ("rt_render + rt_map_image per step: the RGBA32F frame lands in page-locked host memory of the library; pipelined = the "
                   "copy of frame k runs beside the trace of frame k + 1 and each call returns the previous frame")
    return out


def view_change(r, scene, cam, cfg, torch, steps, kinds=("rotate", "translate")):
    """The camera-move frame: the reference restarts its accumulation whenever the camera moves (RaytracerApp.cpp:259-263,
    Camera.cpp:21-103), so while the user navigates every Render is a first frame that follows a new rt_set_camera.  N Renders,
    each after a slightly different pose, everything that a new view costs inside the timed region (ray directions, column
    numbering, unit classification, pre-pass, full resolve).  `rotate`: the view direction turns by 2 mrad per Render about the
    vertical axis (no pixel keeps direction.x == +0); `translate`: the camera moves 5 mm forward per Render (the centre column of
    the default camera keeps direction.x == +0, so the column pre-pass runs every time)."""
    import numpy as np
    pos0, fwd0 = cam.GetPosition().copy(), cam.GetDirection().copy()
    out = {"steps": steps, "note": "ms per Render, each Render preceded by rt_set_camera with a new pose and rendered as frame index 1; "
                                   "all per-view set-up inside the timed region; image resident in HBM"}

    def poses(kind, i):
        if kind == "rotate":
            a = np.float32(0.002 * (i + 1))
            c, s_ = np.float32(np.cos(a)), np.float32(np.sin(a))
            fwd = np.array([fwd0[0] * c + fwd0[2] * s_, fwd0[1], -fwd0[0] * s_ + fwd0[2] * c], np.float32)
            return pos0, fwd
        return (pos0 + fwd0 * np.float32(0.005 * (i + 1))).astype(np.float32), fwd0

    try:
        for kind in kinds:
            for i in range(3):                                   # warm: buffers of the per-view data are allocated here
                cam.SetPose(*poses(kind, 100 + i)); r.upload(scene, cam); r.render_only(1)
            r.sync(); torch.cuda.synchronize()
            r.reset_counters()
            t0 = time.perf_counter()
            for i in range(steps):
                cam.SetPose(*poses(kind, i)); r.upload(scene, cam); r.render_only(1)
            r.sync(); torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            cnt = r.read_counters()
            out[kind] = {"ms_per_render": dt / steps * 1e3, "value": cnt.rays / dt / 1e6, "unit": "Mrays/s",
                         "view_setups": int(cnt.view_setups)}
    finally:
        cam.SetPose(pos0, fwd0); r.upload(scene, cam)
    return out


def device_clock_hz(torch, device=0):
    """The engine clock the device reports (its maximum: the kernels here run at or near it; rocm-smi on the box shows the same)."""
    try:
        return float(torch.cuda.get_device_properties(device).clock_rate) * 1e3
    except Exception:
        return 2.4e9


def run_single(config_id, steps, warmup, order_name, torch, device=0):
    """One-GPU measurement of a configuration, one Render per launch, plus throughput mode.  Returns (record, scene, cam)."""
    import raytracer_amd as rt
    cfg = CONFIGS[config_id]
    w, h = cfg["w"], cfg["h"]
    scene = build_scene(cfg)
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    r = rt.Renderer(device)
    r.download = False
    st = r.GetSettings()
    st.SceneMode, st.Bounces, st.TraversalOrder = cfg["mode"], cfg["bounces"], ORDERS[order_name]
    r.OnResize(w, h)
    r.set_kernel_timing(timing_every(steps, 1, cfg["mode"]))
    r.upload(scene, cam)
    settle_gc()
    r.reset_accum()
    for f in range(1, warmup + 1):
        r.render_only(f + 1000)
    r.sync(); torch.cuda.synchronize()
    r.reset_accum(); r.reset_counters(); r.kernel_time(reset=True)
    frames = list(range(1, steps + 1))
    t0 = time.perf_counter()
    for f in frames:
        r.render_only(f)
    r.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cnt, kt = r.read_counters(), r.kernel_time()
    clock_hz = device_clock_hz(torch, device)
    roof = roofline(config_id, cfg, scene, r, st, order_name, frames, cnt, kt, 1, clock_hz=clock_hz)
    rec = {"config": config_id, "workload": cfg["name"], "value": cnt.rays / dt / 1e6, "unit": "Mrays/s", "steps": steps,
           "ms_per_step": dt / steps * 1e3, "rays_per_frame": cnt.rays / steps,
           "empty_unit_rays_per_frame": cnt.empty_unit_rays / steps, "roofline": roof,
           "throughput_mode": throughput_mode(r, cfg, torch, max(1, steps // FRAMES_PER_LAUNCH), cnt.rays / steps,
                                              roof.get("algorithmic_bytes_per_launch"), len(scene.spheres), config_id, clock_hz)}
    if cfg["mode"] != 1:
        rec["first_hit_reuse"] = first_hit_reuse(r, st, cfg, torch, steps, max(1, steps // FRAMES_PER_LAUNCH))
    rec["view_change"] = view_change(r, scene, cam, cfg, torch, min(steps, 32))
    return rec, scene, cam, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the tree configurations 3, 4 and 6 of the default one-GPU run")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU baseline sample (0 = auto)")
    ap.add_argument("--profile", default="", choices=["", "single", "throughput"],
                    help="for profiler passes: only one-Render-per-launch steps, or only launches of 16 Renders (timed launches of one kind "
                         "under the trace kernel's name; no hand-off, no other configurations)")
    ap.add_argument("--order", default="auto", choices=sorted(ORDERS),
                    help="KD tree walk order of mesh scenes (auto = front to back on a KD tree)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))

    import torch                                   # device sync, barrier, RCCL: plumbing only
    import torch.distributed as dist
    import raytracer_amd as rt
    from raytracer_amd import dist as rtd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    # Rehearsal on a box with fewer GPUs than ranks: RT_BENCH_DEVICE=<n> puts every rank on that device and RT_BENCH_BACKEND=gloo moves
    # the collectives through host memory (RCCL refuses two ranks on one GPU).  The numbers of such a run mean nothing; the code path
    # of every rank is the one `--gpus N` runs.
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if os.environ.get("RT_BENCH_DEVICE"):
        local_rank = int(os.environ["RT_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    # RT_BENCH_FORCE_DIST=1 runs the multi-GPU code path (process group, RCCL sum, rank-0 resolve) even with one rank,
    # so it can be rehearsed on a one-GPU box
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    cfg = CONFIGS[args.config]
    w, h, K, W = cfg["w"], cfg["h"], args.steps, args.warmup
    scene = build_scene(cfg)
    cam = rt.Camera(70.0, 0.05, 100.0)
    cam.OnResize(w, h)
    r = rt.Renderer(local_rank)
    r.download = False                               # frames stay in HBM; the host copy is priced separately (with_d2h)
    st = r.GetSettings()
    st.SceneMode, st.Bounces = cfg["mode"], cfg["bounces"]
    st.TraversalOrder = ORDERS[args.order]
    r.OnResize(w, h)
    stream = torch.cuda.Stream()                     # one stream for the kernels and the collective that follows them
    torch.cuda.set_stream(stream)
    r.set_stream(stream.cuda_stream)
    r.upload(scene, cam)
    settle_gc()

    def barrier():
        if use_dist:
            dist.barrier()

    tile_split = cfg.get("split") == "tile"
    if tile_split:
        # screen-tile split: every rank renders all K frames of its own 16-row bands; bands are gathered on rank 0 per frame
        r.set_bands(rank, world)
        warm = list(range(1, W + 1)); frames = list(range(1, K + 1))
        bands = rtd.BandGather(lambda: rtd.image_tensor(r), rank, world) if use_dist else None
        acc_t = None
    else:
        # sample split: rank r renders frame indices r+1, r+1+N, ... (world == 1: 1, 2, 3, ...)
        warm = rtd.frames_for_rank(rank, world, W)
        frames = rtd.frames_for_rank(rank, world, K, first=1)
        acc_t = rtd.accum_tensor(r) if use_dist else None
        bands = None

    r.reset_accum()
    for f in warm:
        r.render_only(f + 1000)                      # warmup only heats caches/clocks; its sum is discarded
        if bands is not None:
            bands.gather()
    if use_dist and not tile_split:
        rtd.sum_accumulation(acc_t)
    r.sync(); torch.cuda.synchronize(); barrier()
    r.reset_accum(); r.reset_counters(); r.kernel_time(reset=True)
    torch.cuda.synchronize(); barrier()

    per_launch = FRAMES_PER_LAUNCH if args.profile == "throughput" else 1
    r.set_kernel_timing(timing_every(K, per_launch, cfg["mode"]))
    t0 = time.perf_counter()
    for f in frames[::per_launch]:                   # every rank starts from a zeroed buffer (rt_reset_accum)
        r.render_only(f, per_launch)
        if bands is not None:
            bands.gather()                           # the displayed frame, assembled on rank 0
    if use_dist and not tile_split:
        rtd.sum_accumulation(acc_t)                  # RCCL reduce of the rgb sums onto rank 0
        if rank == 0:
            r.resolve(world * K)
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0

    dt_t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    cnt = r.read_counters()
    rays_t = torch.tensor([float(cnt.rays)], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays_t, op=dist.ReduceOp.SUM)
    dt_max, rays_total = float(dt_t.item()), float(rays_t.item())
    kt = r.kernel_time()

    if rank == 0:
        clock_hz = device_clock_hz(torch, local_rank)
        roof = roofline(args.config, cfg, scene, r, st, args.order, frames, cnt, kt, world, bool(args.profile), clock_hz)
        out = {
            "metric": "Mrays/s (primary+bounce)", "value": rays_total / dt_max / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt_max / K * 1e3, "higher_is_better": True,
            "scaling": "strong" if tile_split else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["name"], "width": w, "height": h, "bounces": cfg["bounces"],
                       "frames_per_gpu": K, "split": ("tile" if tile_split else "sample") if use_dist else "none",
                       "rays_per_frame": rays_total / (K if tile_split else K * world),
                       "empty_unit_rays_per_frame": cnt.empty_unit_rays / K, "image": "resident in HBM (no D2H in the timed region)"},
            "roofline": roof,
        }
        if os.environ.get("RT_BENCH_DEVICE") or backend != "nccl":
            out["rehearsal"] = "ranks share device %d, collectives over %s: a check of the code path, not a measurement" % (local_rank, backend)
        if args.profile:
            out["profile"] = args.profile + (": %d Renders per launch in the timed region" % per_launch)
        if world == 1 and not use_dist and not args.profile:
            out["throughput_mode"] = throughput_mode(r, cfg, torch, max(1, K // FRAMES_PER_LAUNCH), rays_total / K,
                                                     roof.get("algorithmic_bytes_per_launch"), len(scene.spheres), args.config, clock_hz)
            if cfg["mode"] != 1:
                out["first_hit_reuse"] = first_hit_reuse(r, st, cfg, torch, K, max(1, K // FRAMES_PER_LAUNCH))
            out["with_d2h"] = with_d2h(r, torch, min(K, 32))
            # what a caller of the drop-in Renderer::Render sees per call (the frame ends in host memory, Renderer.cpp:134), beside
            # ms_per_step (frames resident in HBM, what `value` is quoted on)
            out["ms_per_step_with_d2h"] = out["with_d2h"]["ms_per_step_map_image"]
            out["ms_per_step_with_d2h_pipelined"] = out["with_d2h"]["ms_per_step_map_image_pipelined"]
            r.set_stream(stream.cuda_stream)
            out["view_change"] = view_change(r, scene, cam, cfg, torch, min(K, 32))
        if not args.no_cpu_baseline and world == 1:             # the CPU oracle is timed beside the one-GPU run only
            n = args.cpu_frames or (16 if cfg["mode"] == 1 else 6)       # about 15-30 core-seconds of CPU work
            if cfg["mesh"] and cfg["mesh"][1] > 1000:
                n = 1                                                    # 10 M triangles at 4K: one frame of every 64th row
            out["cpu_baseline"] = cpu_baseline(cfg, scene, cam, n)
        if world == 1 and not use_dist and args.config == 2 and not args.no_other_configs and not args.profile:
            # the tree configurations in the same run, so that one driver-timed command covers them (one Render per launch and
            # throughput mode each; counters, event times and rooflines as for the main configuration)
            del r
            others = []
            for cid, steps in ((3, 32), (4, 16), (6, 32)):
                try:
                    rec, _s, _c, rr = run_single(cid, steps, 3, args.order, torch, local_rank)
                    del rr, _s
                    others.append(rec)
                except Exception as e:                                   # the main line must not be lost to a side measurement
                    others.append({"config": cid, "error": "%s: %s" % (type(e).__name__, e)})
            out["other_configs"] = others
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
