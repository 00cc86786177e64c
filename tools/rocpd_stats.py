"""Summarise a rocprofv3 rocpd sqlite database: per-kernel calls / total / avg / min / max (us)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
symcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
namecol = "kernel_name" if "kernel_name" in symcols else ("display_name" if "display_name" in symcols else symcols[-1])
q = f"""select s.{namecol}, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3
        from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.{namecol} order by 3 desc"""
rows = cur.execute(q).fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
for r in rows:
    print(f"{r[0][:70]:70s} {r[1]:6d} {r[2]:12.1f} {r[3]:10.1f} {r[4]:10.1f} {r[5]:10.1f} {100*r[2]/tot:6.2f}")
