#!/bin/bash
# Run on the GPU box from the repo root: two separate rocprofv3 --pmc passes (FETCH_SIZE costs 3 TCC counters, WRITE_SIZE 2:
# they do not fit one pass, MI355X_MICROARCH.md) over the default bench command, reduced to profiles-style JSON on stdout-file.
# usage: tools/pmc_traffic.sh gpurun_out/rNN_pmc_hbm_traffic.json
OUT=${1:-gpurun_out/pmc_hbm_traffic.json}
R=$PWD; cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$C
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$C -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > /dev/null 2>&1
done
cd $R
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections, statistics
out = {"command": "rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline (one pass per counter)",
       "note": "FETCH_SIZE/WRITE_SIZE are KiB per dispatch as rocprofv3 reports them. MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read, so the corrected read traffic is up to 2x the raw value; other access widths and WRITE_SIZE are uncalibrated. Both raw and x2 are listed. A multi-pass kernel dispatch covers passes_per_launch passes.",
       "kernels": {}}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{C}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == C:
            per[r["Kernel_Name"].split("(")[0]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, d in per.items():
        if not any(s in k for s in ("pass_kernel", "multipass", "fit_planes", "prepare", "cov_update")):
            continue
        v = list(d.values())
        e = out["kernels"].setdefault(k, {})
        e[f"{C}_KiB_mean"] = statistics.mean(v); e[f"{C}_KiB_median"] = statistics.median(v); e[f"{C}_dispatches"] = len(v)
for k, e in out["kernels"].items():
    if "FETCH_SIZE_KiB_median" in e:
        e["read_bytes_raw"] = e["FETCH_SIZE_KiB_median"] * 1024; e["read_bytes_x2_corrected"] = 2 * e["read_bytes_raw"]
    if "WRITE_SIZE_KiB_median" in e:
        e["write_bytes_raw"] = e["WRITE_SIZE_KiB_median"] * 1024
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: {a: round(b) for a, b in e.items() if a.endswith("raw") or a.endswith("corrected")} for k, e in out["kernels"].items()}))
PY
