import sqlite3, sys
db=sqlite3.connect(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/prof_kb/pmc_sq/kb_results.db'); cur=db.cursor()
rows=cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where (kernel_name like '%k_weights_L<0>%' or kernel_name like '%k_rays_pl%') and counter_name in ('SQ_INSTS_VALU','SQ_INSTS_VMEM_RD','SQ_ACTIVE_INST_VALU','SQ_WAVE_CYCLES','SQ_WAIT_INST_ANY','SQ_WAIT_ANY') group by kernel_name, counter_name").fetchall()
for r in rows: print("%-50s %-22s %4d %16.0f  per-64-rays %8.1f"%(r[0][:50],r[1],r[2],r[3], r[3]/562500))
