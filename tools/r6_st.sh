mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_train_lp.py tests/test_gpu_wgrad_det.py -x -q -m gpu -k "stem" > gpurun_out/r6/t_stem.txt 2>&1; echo "stem rc $?"; tail -5 gpurun_out/r6/t_stem.txt
python tools/step_times.py --backbone resnet_v2_50 --tune > gpurun_out/r6/step_times_train_c4.txt 2>&1; head -3 gpurun_out/r6/step_times_train_c4.txt | cut -c1-220; tail -1 gpurun_out/r6/step_times_train_c4.txt | cut -c1-200
python bench.py --train --preset c4 --no-traffic > gpurun_out/r6/train_c4.json 2> gpurun_out/r6/train_c4.err; cut -c1-300 gpurun_out/r6/train_c4.json
