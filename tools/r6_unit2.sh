mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q -m gpu > gpurun_out/r6/t_unit.txt 2>&1; echo "unit rc $?"; tail -n 12 gpurun_out/r6/t_unit.txt
timeout 600 python -m pytest tests/test_gpu_lowp.py tests/test_gpu_model.py tests/test_gpu_configs.py -x -q -m gpu -k "resnet or c4" > gpurun_out/r6/t_resnet.txt 2>&1; echo "resnet rc $?"; tail -n 5 gpurun_out/r6/t_resnet.txt
timeout 300 python tools/seq_vs_warm.py --preset c4 > gpurun_out/r6/seq_c4_pair.txt 2>&1; grep -E "unit_1|block1|conv launches|other ops" gpurun_out/r6/seq_c4_pair.txt
GV_NO_PAIR=1 timeout 300 python tools/seq_vs_warm.py --preset c4 > gpurun_out/r6/seq_c4_nopair.txt 2>&1; grep -E "conv launches|other ops" gpurun_out/r6/seq_c4_nopair.txt
timeout 600 python bench.py --preset c4 --no-cpu-baseline --no-traffic --no-exact > gpurun_out/r6/bench_c4_units.json 2> gpurun_out/r6/bench_c4_units.err; tail -c 1800 gpurun_out/r6/bench_c4_units.json
