mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q -m gpu > gpurun_out/r6/t_unit.txt 2>&1; echo "unit rc $?"; tail -n 4 gpurun_out/r6/t_unit.txt
for i in 1 2; do
timeout 300 python tools/seq_vs_warm.py --preset c4 > gpurun_out/r6/seq_c4_fd3_$i.txt 2>&1; grep -E "conv2\+|conv launches" gpurun_out/r6/seq_c4_fd3_$i.txt | cut -c60-200
done
