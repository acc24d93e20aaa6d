#!/usr/bin/env python
"""one line per bench JSON file given on the command line (value, ms/step, roofline headline, cpu baseline)"""
import json, sys
for f in sys.argv[1:]:
    for ln in open(f):
        if not ln.startswith("{"):
            continue
        d = json.loads(ln); r = d["roofline"]; c = d.get("cpu_baseline", {})
        print(f.split("/")[-1], "|", d["config"]["workload"][:60], "|", d["config"]["branch"][:12], "| B", d["config"]["segments_per_gpu"], "streams", d["config"]["sub_batch_streams"],
              "| value", d["value"], "ms/step", d["ms_per_step"], "| kernel", r["kernel"], "frac", r["frac"], "step", r["step_executed_frac"],
              "non-wino", r["non_winograd_conv_time_fraction_single_stream"], "| cpu", c.get("value"), c.get("seconds_per_evaluation"))
