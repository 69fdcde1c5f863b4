mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_chain.py -x -q -m gpu > gpurun_out/r6/t_chain.txt 2>&1; echo "chain rc $?"; tail -n 12 gpurun_out/r6/t_chain.txt
python tools/chain_probe.py 2>&1 | grep "^d " | tee gpurun_out/r6/chain_probe2.txt
gvcnn-tf_amd/build/mfma_shape_ab 40000 > gpurun_out/r6/mfma_shape_ab.txt 2>&1; cat gpurun_out/r6/mfma_shape_ab.txt
