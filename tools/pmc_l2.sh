# L2 (TCC) hit rate of the bf16 training step's MFMA kernels.  usage: bash tools/pmc_l2.sh -> gpurun_out/pmc_l2/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_l2; mkdir -p $O
export GV_NO_TUNE=1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $O/p1 -o p --output-format csv -- python3 $R/tools/train_bench.py --shapes 32 --steps 1 --storage bf16 > $O/p1.log 2>&1
tail -3 $O/p1.log
python3 - <<PY > $O/summary.txt
import csv, glob, collections
FAM = [("conv fwd/dgrad", ("conv_igemm_lp", "conv3x3_halo", "conv_stem_patch")), ("wgrad per tap", ("conv_wgrad_lp",)),
       ("wgrad strip", ("conv_wgrad_strip",)), ("BN sums", ("grouped_sums",)), ("BN apply", ("bn_stream",))]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for path in glob.glob("$O/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        fam = next((f for f, keys in FAM if any(k in r["Kernel_Name"] for k in keys)), None)
        if fam:
            acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
print("rocprofv3 --pmc TCC_* over tools/train_bench.py --shapes 32 --steps 1 --storage bf16")
for fam, _ in FAM:
    c = acc[fam]
    if c:
        print("%-16s TCC_REQ %.3g  HIT %.3g  MISS %.3g  hit rate %.1f %%  EA0_RDREQ %.3g"
              % (fam, c["TCC_REQ_sum"], c["TCC_HIT_sum"], c["TCC_MISS_sum"],
                 100 * c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1), c["TCC_EA0_RDREQ_sum"]))
PY
cat $O/summary.txt
