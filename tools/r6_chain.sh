mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_chain.py -x -q -m gpu > gpurun_out/r6/t_chain.txt 2>&1; echo "chain rc $?"; tail -n 15 gpurun_out/r6/t_chain.txt
python -m pytest tests/test_gpu_p3.py -x -q -m gpu -k "rebuilds" > gpurun_out/r6/t_p3b.txt 2>&1; echo "p3 rc $?"; tail -n 3 gpurun_out/r6/t_p3b.txt
python tools/seq_vs_warm.py --preset c4 > gpurun_out/r6/seq_c4_chain.txt 2>&1; grep -E "block1|block2/unit_[123]|conv launches|other ops" gpurun_out/r6/seq_c4_chain.txt
GV_NO_CHAIN=1 python tools/seq_vs_warm.py --preset c4 > gpurun_out/r6/seq_c4_nochain.txt 2>&1; grep -E "conv launches|other ops" gpurun_out/r6/seq_c4_nochain.txt
