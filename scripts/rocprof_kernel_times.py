"""Per-kernel dispatch count and mean duration from a rocprofv3 --kernel-trace database (rocpd sqlite)."""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute(f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3 from {kd} d "
                 f"join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
print("kernel,calls,avg_us,min_us")
for r in rows:
    if 'lz4mi' in r[0]:
        print(f"{r[0]},{r[1]},{r[2]:.1f},{r[3]:.1f}")
