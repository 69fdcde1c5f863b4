for t in 0 5 2; do for d in 0 8192 16384 16388; do python tools/ws_one.py $t 1 7 192 192 17 17 640 $d 1 2>&1 | grep -v amdgpu | tail -1; done; done
