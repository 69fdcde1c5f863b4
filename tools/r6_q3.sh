mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q -m gpu > gpurun_out/r6/t_unit.txt 2>&1; echo "unit rc $?"; tail -n 6 gpurun_out/r6/t_unit.txt
timeout 600 python -m pytest tests/test_gpu_lowp.py -x -q -m gpu -k "xpre or resnet or preact" > gpurun_out/r6/t_xpre.txt 2>&1; echo "xpre rc $?"; tail -n 3 gpurun_out/r6/t_xpre.txt
for i in 1 2; do
timeout 300 python tools/seq_vs_warm.py --preset c4 > gpurun_out/r6/seq_c4_tail_$i.txt 2>&1; grep -E "block3/unit_[2-6]/bottleneck_v2/conv1|conv launches" gpurun_out/r6/seq_c4_tail_$i.txt | cut -c40-200
done
