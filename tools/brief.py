"""Reads bench.py JSON lines on stdin, prints one compact line each (label = argv[1])."""
import json
import sys
label = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d["roofline"]
    print(label, "value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), "roof GB/s", round(r["achieved"], 1),
          "launch_ms", round(r["avg_launch_ms"], 3), "share", round(r["intersect_share_of_pass"], 3),
          "nodes", round(r["kd_nodes_per_path_bounce"], 1), "tris", round(r["tris_per_path_bounce"], 1),
          "non-kd ms/step", round(d["ms_per_step"] * (1 - r["intersect_share_of_pass"]), 3))
