#!/usr/bin/env python3
"""Prints the figures of bench.py JSON lines side by side: tools/show_bench.py FILE..."""
import json, sys
def r(v): return round(v, 4) if isinstance(v, float) else v
def show(d, tag):
    print(tag, "Mrays/s", round(d["value"]), "ms/frame", r(d["ms_per_step"]), "kernel ms", r(d["roofline"].get("kernel_avg_ms")),
          "frac", r(d["roofline"]["frac"]), "frac_hbm_moved", r(d["roofline"].get("frac_hbm_moved")))
    t = d.get("throughput_mode")
    if t: print("   x16: ms/frame", r(t["ms_per_frame"]), "Mrays/s", round(t["value"]), "kernel", r(t["kernel_ms_per_frame"]), "frac", r(t.get("roofline_frac")))
    f = d.get("first_hit_reuse")
    if f:
        for k in ("one_render_per_launch", "throughput_mode"):
            print("   reuse %-22s ms/frame" % k, r(f[k]["ms_per_frame"]), "Mrays/s", round(f[k]["value"]), "traced", round(f[k]["traced_value"]), "kernel", r(f[k]["kernel_ms_per_frame"]))
    w = d.get("with_d2h")
    if w: print("   d2h:", {k: r(v) for k, v in w.items() if k != "note"})
    c = d.get("cpu_baseline")
    if c: print("   cpu:", {k: r(v) for k, v in c.items() if k not in ("sample", "note")})
for p in sys.argv[1:]:
    lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
    if not lines: print(p, "no JSON line"); continue
    d = json.loads(lines[-1])
    show(d, p + ":")
    for o in d.get("other_configs", []):
        if "error" in o: print("  config", o["config"], "ERROR", o["error"])
        else: show(o, "  config %d:" % o["config"])
