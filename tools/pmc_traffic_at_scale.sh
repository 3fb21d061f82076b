#!/bin/bash
# Run on the GPU box from the repo root: FETCH_SIZE / WRITE_SIZE (the counters that are safe on this pool; separate --pmc passes, as
# MI355X_MICROARCH.md prescribes) for the AT-SCALE kernels of the bench line -- lio18_pass_kernel<0> over 32 M (and 8 M) points,
# vio_pass_kernel<0, 1> over 1 M patches -- reduced to a profiles-style JSON that bench.py attaches to roofline.at_scale[*].traffic.
# usage: tools/pmc_traffic_at_scale.sh gpurun_out/rNN_pmc_hbm_traffic_at_scale.json
OUT=${1:-gpurun_out/pmc_hbm_traffic_at_scale.json}
R=$PWD; cd /tmp && export TMPDIR=/tmp
run() {   # tag, bench args...
  local tag=$1; shift
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmcs_${tag}_$C
    timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_${tag}_$C -- python $R/bench.py "$@" > /dev/null 2>&1
  done
}
run lio32M --only at_scale --at-scale-points 32000000
run lio8M --only at_scale --at-scale-points 8000000
run vio1M --only vio_sweep --vio-sweep-patches 1000000
cd $R
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections, statistics
out = {"command": "rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python bench.py --only at_scale --at-scale-points N | --only vio_sweep --vio-sweep-patches M (one pass per counter)",
       "note": "FETCH_SIZE/WRITE_SIZE are KiB per dispatch as rocprofv3 reports them. MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read, so the corrected read traffic is 2x the raw value for these streaming kernels; WRITE_SIZE is uncalibrated (raw). One dispatch = one pass. Median over the dispatches of the timed loop.",
       "kernels": {}}
for tag, want, size in (("lio32M", "lio18_pass_kernel", 32000000), ("lio8M", "lio18_pass_kernel", 8000000), ("vio1M", "vio_pass_kernel", 1000000)):
    e = {}
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(f"gpurun_out/pmcs_{tag}_{C}/**/*counter_collection.csv", recursive=True)
        if not fs:
            continue
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(fs[0])):
            if r["Counter_Name"] == C and want in r["Kernel_Name"]:
                per[r["Kernel_Name"].split("(")[0]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for k, d in per.items():
            v = list(d.values())
            ee = e.setdefault(k, {})
            ee[f"{C}_KiB_median"] = statistics.median(v); ee[f"{C}_dispatches"] = len(v)
    for k, ee in e.items():
        if "FETCH_SIZE_KiB_median" in ee:
            ee["read_bytes_raw"] = ee["FETCH_SIZE_KiB_median"] * 1024; ee["read_bytes_x2_corrected"] = 2 * ee["read_bytes_raw"]
        if "WRITE_SIZE_KiB_median" in ee:
            ee["write_bytes_raw"] = ee["WRITE_SIZE_KiB_median"] * 1024
        ee["units"] = size
        out["kernels"][f"{k}@{size}"] = ee
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: {a: round(b) for a, b in e.items() if a.endswith("raw") or a.endswith("corrected")} for k, e in out["kernels"].items()}))
PY
