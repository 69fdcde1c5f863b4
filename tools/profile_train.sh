# usage: bash tools/profile_train.sh [storage=f32|bf16] [tune=0|1] -> gpurun_out/prof_train_<storage>
cd /tmp && export TMPDIR=/tmp
S=${1:-f32}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_train_$S; mkdir -p $O
if [ "${2:-0}" = "0" ]; then export GV_NO_TUNE=1; fi
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/tools/train_bench.py --shapes 32 --steps 3 --storage $S > $O/train.log 2>&1
grep -a "views/s" $O/train.log
python3 $R/tools/train_profile_summary.py $O/kt/kt_kernel_trace.csv | tee $O/summary.txt
