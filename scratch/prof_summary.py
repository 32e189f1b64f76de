"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel table: calls, total, avg, % of GPU time."""
import sqlite3, sys
db, out = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
lines = [f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}"]
for n, k, t, a, mn, mx in rows:
    lines.append(f"{n[:70]:70s} {k:6d} {t/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*t/tot:6.2f}")
lines.append(f"TOTAL GPU kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
txt = "\n".join(lines)
print(txt)
if out: open(out, 'w').write(txt + "\n")
