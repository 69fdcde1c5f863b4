cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_train_bf16_n8; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/tools/train_bench.py --shapes 8 --steps 3 --storage bf16 > $O/train.log 2>&1
grep -a "views/s" $O/train.log
python3 $R/tools/train_profile_summary.py $O/kt/kt_kernel_trace.csv | tee $O/summary.txt
