#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of tools/profile.sh from gpurun_out/ into profiles/ (tracked) under a round tag, and derive
profiles/traffic.json (HBM bytes per trace-kernel launch: 2 x FETCH_SIZE + WRITE_SIZE, FETCH_SIZE doubled because gfx950
reports half of the bytes of wide coalesced reads; both counters are in KiB) from the one-Render-per-launch passes.

    python tools/collect_profiles.py r2        # after tools/profile.sh N single|throughput runs on the GPU box"""
import glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
traffic_path = os.path.join(root, "profiles", "traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
# profiles/issue.json: wave-instructions per launch of the timed trace kernel, by kind (what bench.py prices against the rate at which
# the SIMDs can issue), from the same passes
issue_path = os.path.join(root, "profiles", "issue.json")
issue = json.load(open(issue_path)) if os.path.exists(issue_path) else {}
for cfg in (1, 2, 3, 4, 5, 6):
    for kind in ("single", "throughput"):
        d = os.path.join(root, "gpurun_out", "prof_cfg%d_%s" % (cfg, kind))
        summ = d + "_summary.json"
        if not os.path.exists(summ):
            continue
        name = "%s_cfg%d_%s" % (tag, cfg, kind)
        stats = sorted(glob.glob(d + "/trace/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)
        if stats:                                        # the newest: gpurun merges every run's files into the same directory
            shutil.copy(stats[-1], os.path.join(root, "profiles", name + "_kernel_stats.csv"))
        shutil.copy(summ, os.path.join(root, "profiles", name + "_pmc_summary.json"))
        if os.path.exists(d + "_bench.json") and os.path.getsize(d + "_bench.json"):
            shutil.copy(d + "_bench.json", os.path.join(root, "profiles", name + "_bench.json"))
        s = json.load(open(summ))
        # launches of the timed kind in the profiled command (timed + warm-up); the trace kernel of a scene that parks rays is
        # also launched, small, to finish the parked paths: the per-launch means are scaled to the launches that draw units
        n_main = None
        if os.path.exists(d + "_bench.json") and os.path.getsize(d + "_bench.json"):
            b = json.loads(open(d + "_bench.json").read())
            n_main = b["roofline"]["launches"] + (b["warmup"] if kind == "single" else 0)
        for k, cs in s["counters"].items():
            # the kernels the benchmark times: no counting, front-to-back walk (the reference-order walk runs once, for the counts)
            # (or, where RT_ORDER_AUTO picks the reference's order -- config 6 --, the timed reference-order kernel)
            timed = ("rt_trace_spheres_kernel<false" in k or "rt_trace_kernel<0, false, true>" in k or "rt_trace_kernel<2, false, true>" in k or
                     "rt_trace_kernel<0, false, false>" in k or "rt_trace_kernel<2, false, false>" in k)
            if timed and "SQ_INSTS_VALU" in cs and "SQ_INSTS_SALU" in cs:
                key = "config%d" % cfg if kind == "single" else "config%d_x16" % cfg
                calls = cs["SQ_INSTS_VALU"]["launches"]
                scale = calls / n_main if n_main and calls > n_main else 1.0
                get = lambda name: cs[name]["mean"] * scale if name in cs else None
                vmem = (get("SQ_INSTS_VMEM_RD") or 0.0) + (get("SQ_INSTS_VMEM_WR") or 0.0)
                rec = {"kernel": k, "valu": get("SQ_INSTS_VALU"), "salu": get("SQ_INSTS_SALU"), "smem": get("SQ_INSTS_SMEM"), "vmem": vmem,
                       "lds": get("SQ_INSTS_LDS"), "waves": get("SQ_WAVES"),
                       "launch": "one Render" if kind == "single" else "16 Renders",
                       "source": "rocprofv3 --pmc SQ_INSTS_* (tools/profile.sh %d %s), wave-instructions per launch, round tag %s" % (cfg, kind, tag)}
                if "SQ_THREAD_CYCLES_VALU" in cs and "SQ_ACTIVE_INST_VALU" in cs and cs["SQ_ACTIVE_INST_VALU"]["mean"]:
                    # thread-cycles over 64 x instruction-cycles: the share of the lanes a VALU instruction had switched on
                    rec["valu_lane_use"] = cs["SQ_THREAD_CYCLES_VALU"]["mean"] / (64.0 * cs["SQ_ACTIVE_INST_VALU"]["mean"])
                if "SQ_WAIT_ANY" in cs and "SQ_WAVE_CYCLES" in cs and cs["SQ_WAVE_CYCLES"]["mean"]:
                    rec["wave_cycles_waiting"] = cs["SQ_WAIT_ANY"]["mean"] / cs["SQ_WAVE_CYCLES"]["mean"]
                rec["instructions"] = sum(v for v in (rec["valu"], rec["salu"], rec["smem"], rec["vmem"], rec["lds"]) if v)
                issue[key] = rec
            if timed and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                key = "config%d" % cfg if kind == "single" else "config%d_x16" % cfg
                calls = cs["FETCH_SIZE"]["launches"]
                scale = calls / n_main if n_main and calls > n_main else 1.0
                traffic[key] = {
                    "kernel": k, "fetch_size_KiB": cs["FETCH_SIZE"]["mean"], "write_size_KiB": cs["WRITE_SIZE"]["mean"],
                    "kernel_launches_in_pass": calls, "of_which_draw_units": n_main if n_main else calls,
                    "hbm_bytes_per_launch": (2.0 * cs["FETCH_SIZE"]["mean"] + cs["WRITE_SIZE"]["mean"]) * 1024.0 * scale,
                    "launch": "one Render" if kind == "single" else "16 Renders",
                    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/profile.sh %d %s, round tag %s" % (cfg, kind, tag)}
json.dump(traffic, open(traffic_path, "w"), indent=1)
json.dump(issue, open(issue_path, "w"), indent=1)
print(json.dumps(traffic, indent=1))
