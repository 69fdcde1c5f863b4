#!/bin/bash
# PMC passes over ONE conv launch of the Inception plan (tools/one_conv.py).  Usage: bash tools/pmc_conv.sh OUTDIR MATH STORAGE OP TILE
O=$1; MATH=$2; STORAGE=$3; OP=$4; TILE=$5
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$O/op${OP}_p1 -o p --output-format csv -- python3 $R/tools/one_conv.py --math $MATH --storage $STORAGE --op $OP --tile $TILE --reps 20 > $R/$O/op${OP}_p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS -d $R/$O/op${OP}_p2 -o p --output-format csv -- python3 $R/tools/one_conv.py --math $MATH --storage $STORAGE --op $OP --tile $TILE --reps 20 > $R/$O/op${OP}_p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM TCC_HIT_sum TCC_MISS_sum -d $R/$O/op${OP}_p3 -o p --output-format csv -- python3 $R/tools/one_conv.py --math $MATH --storage $STORAGE --op $OP --tile $TILE --reps 20 > $R/$O/op${OP}_p3.log 2>&1
grep "^op " $R/$O/op${OP}_p1.log
python3 - <<PY
import csv,collections,glob
acc={}
for f in glob.glob('$R/$O/op${OP}_p*/**/*counter_collection.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    sel=[r for r in rows if 'conv' in r['Kernel_Name']]
    byc=collections.defaultdict(list)
    for r in sel: byc[r['Counter_Name']].append((int(r['Dispatch_Id']),float(r['Counter_Value']),r['Kernel_Name']))
    for c,v in byc.items():
        v.sort(); v=v[-20:]
        acc[c]=sum(x[1] for x in v)/len(v); acc['kernel']=v[-1][2][:90]
print(acc.get('kernel'))
for c in sorted(k for k in acc if k!='kernel'): print('%-32s %.5g'%(c,acc[c]))
g=acc.get('GRBM_GUI_ACTIVE',0)/8
if g and 'SQ_VALU_MFMA_BUSY_CYCLES' in acc:
    print('MFMA pipe busy = %.3f of SIMD-cycles (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs))' % (acc['SQ_VALU_MFMA_BUSY_CYCLES']/(g*1024)))
    print('VALU instructions per MFMA (32x32x16): %.2f' % (acc['SQ_INSTS_VALU']/(acc['SQ_VALU_MFMA_BUSY_CYCLES']/32)))
if 'SQ_LDS_IDX_ACTIVE' in acc and acc['SQ_LDS_IDX_ACTIVE']:
    print('LDS bank conflict share of LDS-active cycles: %.4f' % (acc['SQ_LDS_BANK_CONFLICT']/acc['SQ_LDS_IDX_ACTIVE']))
if 'SQ_WAVE_CYCLES' in acc:
    w=acc['SQ_WAVE_CYCLES']
    print('of wave-cycles: WAIT_ANY %.3f  WAIT_INST_ANY %.3f  ACTIVE_INST_ANY %.3f' % (acc['SQ_WAIT_ANY']/w, acc['SQ_WAIT_INST_ANY']/w, acc['SQ_ACTIVE_INST_ANY']/w))
if 'TCC_HIT_sum' in acc:
    print('L2 hit rate %.3f' % (acc['TCC_HIT_sum']/(acc['TCC_HIT_sum']+acc['TCC_MISS_sum'])))
PY
