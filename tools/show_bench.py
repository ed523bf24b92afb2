"""Print the roofline-relevant fields of a bench.py JSON line (developer convenience)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], d["unit"], "| n_gpus", d["n_gpus"], "| ms/step", d["ms_per_step"], "kernel_ms (median)", r["kernel_ms"])
print("  executed_issue_frac", r.get("executed_issue_frac"), "at profiled clock",
      (r.get("executed_issue") or {}).get("frac_at_profiled_clock"), "@", (r.get("executed_issue") or {}).get("profiled_clock_GHz"), "GHz",
      "| algorithmic_frac", r.get("algorithmic_frac"), "| traffic", r.get("traffic"), "vs", r["algorithmic"]["bytes_per_launch"])
if d.get("rccl"):
    print("  rccl", d["rccl"], "| end_to_end_sharded", (d.get("end_to_end_sharded") or {}).get("ms"))
for k, v in d.get("extras", {}).items():
    if isinstance(v, dict):
        ms = v.get("ms", v.get("kernel_ms_slowest_slot", v.get("kernel_ms_slowest_rank")))
        print(f"  {k:28s} ms {ms!s:10s} executed {v.get('executed_issue_frac')!s:7s} algorithmic "
              f"{v.get('algorithmic_frac', v.get('algorithmic_frac_per_slot', v.get('algorithmic_frac_per_rank')))!s:7s}"
              + (f" end_to_end_ms {v['end_to_end_ms']}" if "end_to_end_ms" in v else ""))
cb = d.get("cpu_baseline") or {}
print("  cpu_baseline", cb.get("value"), cb.get("unit"), "| direct", (cb.get("direct_sum") or {}).get("value"),
      "| fft_path (GPU) ms", (d.get("fft_path") or {}).get("ms"))
