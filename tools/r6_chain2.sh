mkdir -p gpurun_out/r6
python tools/chain_probe.py > gpurun_out/r6/chain_probe.txt 2>&1; cat gpurun_out/r6/chain_probe.txt
bash tools/pmc_chain.sh 64 > gpurun_out/r6/pmc_chain_64.txt 2>&1; cat gpurun_out/r6/pmc_chain_64.txt
