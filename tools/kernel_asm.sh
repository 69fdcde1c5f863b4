#!/bin/bash
# Instruction counts of ONE instantiation of the LDS-DMA convolution kernel (csrc/conv_dma.hip), split at the phase
# stamps of the profiling build: setup | prologue issue | main loop | epilogue.  A wave issues at most one instruction per
# four clocks, so the once-per-workgroup phases cost (instructions x 4) clocks at best — count them here, in seconds,
# instead of guessing from noisy timings.
#   tools/kernel_asm.sh "__bf16, 1, 2, 2, 2, 3, 4, 0, 3"  [more instantiations ...]
# (template arguments: T, NP, WM, WN, TM, TN, ST, EPI, STATS; STATS 3 = lean epilogue, 1 / 2 = with forward / backward sums)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=${GV_ASM_OUT:-/tmp/gv_kernel_asm}
mkdir -p "$OUT"
SRC="$OUT/one.hip"
{
  echo '#define GV_KERNEL_ONLY 1'
  echo '#define GV_PHASE_TIMES 1'
  echo "#include \"$ROOT/gvcnn-tf_amd/csrc/conv_dma.hip\""
  echo 'namespace {'
  for inst in "$@"; do echo "template __global__ void conv_dma<$inst>(const ConvArgs);"; done
  echo '}'
} > "$SRC"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I"$ROOT/include" -I"$ROOT/gvcnn-tf_amd/csrc" -S --cuda-device-only -o "$OUT/one.s" "$SRC" 2>/dev/null
python3 - "$OUT/one.s" <<'PY'
import re, sys, collections
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\S*conv_dma\S*):\s*; @', s, re.M):
    body = s[m.end():s.index('.end_amdhsa_kernel', m.end())]
    lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
    idx = [i for i, l in enumerate(lines) if 's_memtime' in l]
    vg = re.search(r'\.amdhsa_next_free_vgpr (\d+)', body)
    ag = re.search(r'; NumAgprs: (\d+)', body)
    sc = re.search(r'; ScratchSize: (\d+)', body)
    print(m.group(1)[:110])
    print("   %d instructions, vgprs %s agprs %s scratch %s; stamps at %s" % (len(lines), vg and vg.group(1), ag and ag.group(1), sc and sc.group(1), idx))
    if len(idx) >= 2:
        a, b = idx[-3], idx[-2]          # main-loop end .. epilogue end
        c = collections.Counter(l.split()[0] for l in lines[a:b])
        print("   epilogue: %d instructions: %s" % (b - a, ", ".join("%s %d" % kv for kv in c.most_common(14))))
PY
