cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_train2; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/tools/train_bench.py --shapes 32 --steps 3 > $O/train.log 2>&1
tail -2 $O/train.log
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/kt/kt_kernel_stats.csv')))
for r in rows[:22]: print(r['Name'][:90].ljust(90), r['Calls'].rjust(6), '%10.3f ms'%(float(r['TotalDurationNs'])/1e6), r['Percentage'])
PY
