cd /root/repo; export TMPDIR=/tmp; R=$(pwd)
for i in 1 2 3; do
( cd /tmp && rm -rf /tmp/prof_kt && rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-em-run > /tmp/b.json 2> /dev/null )
DB=$(find /tmp/prof_kt -name '*_results.db' | head -1)
python profiles/summarize.py $DB gpurun_out/r06i_kernel_stats_$i.csv > /dev/null
head -2 gpurun_out/r06i_kernel_stats_$i.csv | tail -1 | awk -F, '{print "stats run: k_seg_fb calls", $(NF-3), "avg", $(NF-1)}'
python -c "
import json; d=json.loads([l for l in open('/tmp/b.json') if l.startswith('{\"metric\"')][-1]); r=d['roofline']; print('  same run, in-bench median %.2f us (min %.2f max %.2f)' % (r['kernel_ms_timed']*1e3, r['kernel_ms_samples']['min']*1e3, r['kernel_ms_samples']['max']*1e3))"
python - <<'PY'
import sqlite3,glob
db=glob.glob('/tmp/prof_kt/**/*_results.db', recursive=True)[0]
con=sqlite3.connect(db); cur=con.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
k=[t for t in tabs if 'kernel' in t.lower()]
print('  tables:', k[:8])
PY
done
