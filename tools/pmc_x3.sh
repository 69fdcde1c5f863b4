cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_x3
for op in 5; do
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmc_x3/op${op}_p1 -o p --output-format csv -- python3 $R/tools/one_conv.py --math bf16x3 --op $op --tile 3 --reps 20 > $R/gpurun_out/pmc_x3/op${op}_p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_x3/op${op}_p2 -o p --output-format csv -- python3 $R/tools/one_conv.py --math bf16x3 --op $op --tile 3 --reps 20 > $R/gpurun_out/pmc_x3/op${op}_p2.log 2>&1
tail -1 $R/gpurun_out/pmc_x3/op${op}_p1.log
python3 - <<PY
import csv,collections
acc=collections.defaultdict(list)
import glob
for f in glob.glob('$R/gpurun_out/pmc_x3/op${op}_p*/p_counter_collection.csv'):
    rows=list(csv.DictReader(open(f)))
    # only the repeated kernel: conv_igemm_bf16s<4, 1, 1, 3, 3, false> launches at the end (last 20 dispatches of that kernel)
    sel=[r for r in rows if 'conv_igemm_bf16sILi4ELi1ELi1ELi3ELi3ELb0' in r['Kernel_Name'] or 'conv_igemm_bf16s<4, 1, 1, 3, 3, false>' in r['Kernel_Name']]
    byc=collections.defaultdict(list)
    for r in sel: byc[r['Counter_Name']].append((int(r['Dispatch_Id']),float(r['Counter_Value'])))
    for c,v in byc.items():
        v.sort(); v=v[-20:]
        acc[c]=sum(x[1] for x in v)/len(v)
for c in sorted(acc): print('%-28s %.4g'%(c,acc[c]))
g=acc.get('GRBM_GUI_ACTIVE',0)/8
print('mfma util = %.3f' % (acc['SQ_VALU_MFMA_BUSY_CYCLES']/(g*1024)) if g else '')
print('valu per mfma: %.2f' % (acc['SQ_INSTS_VALU']/(acc['SQ_VALU_MFMA_BUSY_CYCLES']/32)))
PY
done
