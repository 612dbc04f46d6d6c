"""sum the per-launch times of tools/layer_times.py by kernel label: python tools/layer_times.py ... | python tools/layer_sum.py"""
import re
import sys
tot = {}
for line in sys.stdin:
    if line.startswith("total"):
        print(line.strip())
        continue
    m = re.match(r"\s*\d+\s+(\S+)\s+(.*?)\s+([\d.]+) us", line)
    if m:
        k = m.group(2).strip()
        t = tot.setdefault(k, [0, 0.0])
        t[0] += 1
        t[1] += float(m.group(3))
for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:40s} {n:3d} launches {us:9.1f} us")
