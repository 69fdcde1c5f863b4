mkdir -p gpurun_out/r6
for i in 1 2; do
python tools/chain_probe.py --d 64 --only chain; python tools/chain_probe.py --d 64 --only chain --dbg 8
done 2>&1 | grep "^d " | tee gpurun_out/r6/chain_exp_full_lines.txt
