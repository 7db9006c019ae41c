"""Timeline of one step from a rocprofv3 --kernel-trace database: start (us from the step's first kernel), duration, hardware queue, grid, kernel.
usage: python tools/trace_timeline.py <dir with *_results.db> [step_from_end=2]"""
import glob, re, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = c.execute(f"select d.start,d.end,d.queue_id,s.kernel_name,d.grid_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
st = [i for i, r in enumerate(rows) if "stage_images" in r[3]]
a, b = st[-2 * back], st[-2 * back + 2]
t0 = rows[a][0]
def short(n):
    n = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", n)
    n = re.sub(r"^_ZN2at6native", "at::", n)
    return n[:72]
for r in rows[a:b]:
    print("%8.1f %7.1f q%d g%-9d %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[2], r[4], short(r[3])))
