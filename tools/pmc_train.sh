# MFMA-pipe utilisation and LDS bank conflicts of the bf16 training step's kernels (rocprofv3 --pmc, two passes).
# usage: bash tools/pmc_train.sh  -> gpurun_out/pmc_train/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_train; mkdir -p $O
export GV_NO_TUNE=1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $O/p1 -o p --output-format csv -- python3 $R/tools/train_bench.py --shapes 32 --steps 1 --storage bf16 > $O/p1.log 2>&1
python3 - <<PY > $O/summary.txt
import csv, glob, collections
FAM = [("conv fwd/dgrad, LDS-DMA tiles (conv_dma)", ("conv_dma",)),
       ("conv fwd/dgrad, register-staged / halo / stem (conv_igemm_lp, conv3x3_halo_lp, conv_stem_patch_lp)", ("conv_igemm_lp", "conv3x3_halo", "conv_stem_patch")),
       ("filter gradient, LDS-DMA (conv_wgrad_dma)", ("conv_wgrad_dma",)),
       ("filter gradient, tap per workgroup (conv_wgrad_lp)", ("conv_wgrad_lp",)),
       ("filter gradient, strip / stem forms", ("conv_wgrad_strip", "conv_wgrad_stem", "conv_wgrad_direct")),
       ("BN sums (grouped_sums_v8)", ("grouped_sums",)), ("BN apply (bn_stream_v8)", ("bn_stream",)),
       ("max pools (argmax forward / backward, with the BatchNorm tail)", ("maxpool",)), ("other pools", ("pool2d", "avgpool"))]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for path in glob.glob("$O/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        fam = next((f for f, keys in FAM if any(k in r["Kernel_Name"] for k in keys)), None)
        if fam is None:
            continue
        acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[fam] += 1
print("rocprofv3 --pmc over tools/train_bench.py --shapes 32 --steps 1 --storage bf16 (heuristic tiles, 2 steps incl. warm-up)")
print("MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); bank conflicts / LDS active cycles")
for fam, _ in FAM:
    c = acc[fam]
    if not c:
        continue
    gui = c["GRBM_GUI_ACTIVE"] / 8.0
    print("%-100s launches %4d  MFMA pipe %5.1f %%  LDS conflicts %5.1f %% of LDS-active  busy cycles %.3g"
          % (fam, cnt[fam], 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024) if gui else 0,
             100 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"] if c["SQ_LDS_IDX_ACTIVE"] else 0, c["SQ_BUSY_CYCLES"]))
PY
cat $O/summary.txt
