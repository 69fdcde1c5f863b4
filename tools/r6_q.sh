mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q -m gpu > gpurun_out/r6/t_unit.txt 2>&1; echo "unit rc $?"; tail -n 6 gpurun_out/r6/t_unit.txt
timeout 300 python tools/seq_vs_warm.py --preset c4 > gpurun_out/r6/seq_c4_final.txt 2>&1; grep -E "conv launches|other ops" gpurun_out/r6/seq_c4_final.txt
