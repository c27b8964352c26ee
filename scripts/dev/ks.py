import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end, duration from kernels order by start"))
t0 = rows[0][1]
ks = [(r[1]-t0, r[3]) for r in rows if 'k_schedule' in r[0]]
es = [(r[1]-t0, r[2]-t0, r[3]) for r in rows if 'k_env_step' in r[0]]
print("k_schedule durations us:", [round(d/1e3) for _, d in ks[:40]])
# timeline of the last few launches
for r in rows[-14:]:
    print("%-22s start %.3f ms end %.3f ms dur %.3f ms" % (r[0][:22], (r[1]-t0)/1e6, (r[2]-t0)/1e6, r[3]/1e6))
