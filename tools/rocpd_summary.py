"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max.

    python tools/rocpd_summary.py gpurun_out/prof_r1/bench_results.db > profiles/r01_bench_kernel_stats.txt
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                  "from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
print(f"# source: {sys.argv[1]}   (rocprofv3 --kernel-trace --stats, rocpd format)")
print(f"{'kernel':<100} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}")
for name, n, tot, avg, mn, mx in rows:
    print(f"{name[:100]:<100} {n:>6} {tot / 1e6:>10.3f} {avg / 1e3:>10.2f} {mn / 1e3:>10.2f} {mx / 1e3:>10.2f} "
          f"{100 * tot / total:>6.2f}")
if "--pmc" in sys.argv:
    q = ("select k.name, p.name, avg(e.value), count(*) from pmc_events e join kernels k on k.id = e.event_id "
         "join pmc_info p on p.id = e.pmc_id group by 1, 2")
    try:
        for r in db.execute(q):
            print(r)
    except Exception as exc:  # schema differs between rocprof versions
        print("pmc query failed:", exc)
