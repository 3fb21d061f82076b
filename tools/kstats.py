"""Per-kernel average / min / max (us) out of a rocprofv3 *_kernel_stats.csv, filtered by substrings:  python tools/kstats.py file.csv mapinc knn_"""
import csv, sys
keys = sys.argv[2:]
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0][:48]
    if not keys or any(k in n for k in keys):
        print("%-48s calls %5s avg %9.1f us  min %8.1f max %8.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
