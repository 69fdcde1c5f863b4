# usage: bash tools/profile_train.sh [storage=f32|bf16] -> gpurun_out/prof_train_<storage>
cd /tmp && export TMPDIR=/tmp
S=${1:-f32}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_train_$S; mkdir -p $O
GV_NO_TUNE=1 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/tools/train_bench.py --shapes 32 --steps 3 --storage $S > $O/train.log 2>&1
tail -2 $O/train.log
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/kt/kt_kernel_stats.csv')))
for r in rows[:28]: print(r['Name'][:100].ljust(100), r['Calls'].rjust(6), '%10.3f ms'%(float(r['TotalDurationNs'])/1e6), r['Percentage'])
PY
