"""Print the roofline-relevant fields of a bench.py JSON line (developer convenience)."""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], d["unit"], "ms/step", d["ms_per_step"], "| roofline achieved", r["achieved"], "frac", r["frac"],
      "traffic", r["traffic"], "| algorithmic", r["algorithmic"])
for k, v in d.get("extras", {}).items():
    if isinstance(v, dict):
        print(" ", k, "ms", v.get("ms"), "Gpair/s", v.get("Gpair_per_s"), "roofline.frac", (v.get("roofline") or {}).get("frac"),
              "valu.frac", (v.get("valu_issue") or {}).get("frac"))
print("  cpu_baseline", d.get("cpu_baseline", {}).get("value"), "fft_path", d.get("fft_path", {}).get("ms"))
