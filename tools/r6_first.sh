# round 6, first GPU call: the new tests of this round + a baseline line per preset on this box
mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_rccl_one_rank.py -x -q -m gpu -s > gpurun_out/r6/t_rccl.txt 2>&1; echo "rccl rc $?"
python -m pytest tests/test_gpu_lowp.py -x -q -m gpu -k "fast_epilogue or test_lp_pools" > gpurun_out/r6/t_fast.txt 2>&1; echo "fast rc $?"
python -m pytest tests/test_gpu_p3.py -x -q -m gpu -k "one_launch or rebuilds" > gpurun_out/r6/t_p3.txt 2>&1; echo "p3 rc $?"
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "c3" > gpurun_out/r6/t_cfg.txt 2>&1; echo "cfg rc $?"
tail -3 gpurun_out/r6/t_*.txt
for p in c4; do
python bench.py --preset $p --no-cpu-baseline --no-traffic --no-exact > gpurun_out/r6/base_$p.json 2> gpurun_out/r6/base_$p.err; tail -c 1500 gpurun_out/r6/base_$p.json
done
python tools/seq_vs_warm.py --preset c4 > gpurun_out/r6/seq_c4.txt 2>&1; tail -70 gpurun_out/r6/seq_c4.txt
