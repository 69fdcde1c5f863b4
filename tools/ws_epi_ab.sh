# A/B of the wave-specialised kernel's epilogues on one box: dbg 0 = pipelined fast epilogue, 1048576 = the general staged one, 4 = none
for shape in "1 7 192 192 17 17 640" "3 3 64 96 35 35 640" "1 7 192 192 12 12 384" "3 3 96 96 25 25 384" "1 7 128 128 12 12 384"; do
  echo "== shape (KH KW CIN COUT H W NB): $shape"
  for t in 0 5 2 1; do for d in 0 1048576 4; do python tools/ws_one.py $t $shape $d 1 2>&1 | grep -v amdgpu | tail -1; done; done
done
