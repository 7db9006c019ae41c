"""kernel list of the LAST inference pass in a rocprofv3 --kernel-trace database of tools/teacher_tail_bench.py (from its stem kernel on)"""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = c.execute(f"select s.kernel_name, d.end-d.start, d.start from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if "stem_pool" in r[0]]
last = rows[idx[-1]:]
t0 = last[0][2]
started = False
for n, t, st in last:
    if "rpn_keys" in n:
        started = True
    if started:
        print("%8.1f  %6.1f us  %s" % ((st - t0) / 1e3, t / 1e3, n[:100]))
print("pass total %.1f us (stem .. last kernel), %d launches" % ((last[-1][2] + last[-1][1] - t0) / 1e3, len(last)))
