import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where (kernel_name like '%k_weights_L<0>%' or kernel_name like '%k_rays%') group by kernel_name, counter_name").fetchall()
for r in rows: print("%-40s %-24s %4d %16.0f  per-64-rays %8.1f" % (r[0][:40], r[1], r[2], r[3], r[3] / 562500))
