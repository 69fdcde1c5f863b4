"""Instruction mix / register use of one kernel in a hipcc -save-temps .s file.  Usage: asm_stats.py FILE.s SUBSTRING [--dump]"""
import sys, re, collections
s = open(sys.argv[1]).read()
key = sys.argv[2]
start = None
for m in re.finditer(r'^(\S+):\s*; @(\S+)', s, re.M):
    if key in m.group(1):
        start = m
        break
if start is None:
    sys.exit("kernel not found")
end = s.index('.end_amdhsa_kernel', start.end())
body = s[start.end():end]
for l in body.split('\n'):
    if any(k in l for k in ('.amdhsa_next_free_vgpr', '.amdhsa_accum_offset', '.amdhsa_group_segment_fixed', 'ScratchSize', 'Occupancy', 'NumVgprs', 'NumAgprs', '.amdhsa_private_segment_fixed_size')):
        print(l.strip())
code = body.split('s_endpgm')[0]
lines = [l.strip() for l in code.split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
print(len(lines), 'instrs')
c = collections.Counter(l.split()[0] for l in lines)
for k, v in c.most_common(28):
    print('  ', k, v)
if '--dump' in sys.argv:
    print(code)
