# Where does the wave-specialised kernel lose against its consumers running alone?  (timing only; results garbage)
#   dbg 0 product | 16384 consumers alone (no loads, no barriers) | 32768 barriers, no loads | 65536 loads, no barriers | +4 no epilogue
for shape in "1 7 192 192 17 17 640" "3 3 64 96 35 35 640" "1 7 192 192 12 12 384"; do
  echo "== shape (KH KW CIN COUT H W NB): $shape"
  for t in 0 5 2; do for d in 0 16384 32768 65536 4 16388; do python tools/ws_one.py $t $shape $d 1 2>&1 | grep -v amdgpu | tail -1; done; done
done
