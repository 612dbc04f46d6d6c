"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs.

    python tools/pmc_summary.py gpurun_out/prof_r01/pmc_*/bench_counter_collection.csv

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB... on gfx950 FETCH_SIZE counts 128-B
requests at 64 B (MI355X_MICROARCH.md, HBM section): the corrected column doubles it.
"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0.0, 0])
for path in sys.argv[1:]:
    with open(path) as fp:
        for row in csv.DictReader(fp):
            k = (row["Kernel_Name"].split("(")[0][-60:], row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"])
            acc[k][1] += 1
print(f"{'kernel':<62} {'counter':<30} {'dispatches':>10} {'avg':>16} {'sum':>18}")
for (k, c), (s, n) in sorted(acc.items(), key=lambda kv: (kv[0][1], -kv[1][0])):
    if "ammc" not in k:
        continue
    print(f"{k:<62} {c:<30} {n:>10} {s / n:>16.2f} {s:>18.1f}")
