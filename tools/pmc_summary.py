"""Merge rocprofv3 --pmc passes into one JSON summary that bench.py reads for its roofline block.

    python tools/pmc_summary.py gpurun_out/r02_pmc --out profiles/r02_pmc_summary.json \
        --command "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"

Every sub-directory of the first argument is one pass (rocprofv3 cannot collect all counters at
once; FETCH_SIZE and WRITE_SIZE need a pass each).  Per kernel (name + grid size, so that the same
template run on two workloads stays apart) the mean over dispatches of every counter, the mean
dispatch duration, and - when both FETCH_SIZE and WRITE_SIZE were collected - the HBM-side bytes per
launch with the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE counts 64 B
per 128-B request: doubled; both are in KB).  The sha256 of every kernel source file is stored so that
bench.py can refuse the entries of a kernel whose sources changed since.

Without --out: prints the per-kernel means (the round-1 behaviour), optionally filtered by a
kernel-name substring given as second positional argument.
"""
import argparse
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def collect(top, substr=None):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(top, "**", "*counter_collection.csv"), recursive=True)):
        seen = set()
        for r in csv.DictReader(open(f)):
            if substr and substr not in r["Kernel_Name"]:
                continue
            key = f"{short(r['Kernel_Name'])} grid={r['Grid_Size']}"
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            did = (f, r["Dispatch_Id"])
            if did not in seen:
                seen.add(did)
                dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return agg, dur


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("top")
    ap.add_argument("substr", nargs="?")
    ap.add_argument("--out")
    ap.add_argument("--command", default="")
    args = ap.parse_args()
    agg, dur = collect(args.top, args.substr)
    if not args.out:
        for k, v in agg.items():
            print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, f"ms={sum(dur[k]) / len(dur[k]):.3f}")
        return
    import bench
    kernels = {}
    for k, v in agg.items():
        e = {c: round(sum(x) / len(x)) for c, x in v.items()}
        e["dispatches_per_pass"] = max(len(x) for x in v.values())
        e["ms"] = round(sum(dur[k]) / len(dur[k]), 4)
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes"] = int((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)
        kernels[k] = e
    out = {"src_sha": bench.source_hashes(), "generated_by": "tools/pmc_summary.py",
           "command": args.command,
           "note": "means per dispatch; ms = mean dispatch duration under the profiler (PMC passes run "
                   "a few % slower than un-profiled); FETCH_SIZE/WRITE_SIZE in KB; hbm_bytes = "
                   "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE tallies 64 B per 128-B request)",
           "kernels": kernels}
    json.dump(out, open(args.out, "w"), indent=1, sort_keys=True)
    print(f"wrote {args.out}: {len(kernels)} kernels")


if __name__ == "__main__":
    main()
