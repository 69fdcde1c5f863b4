# Round 6: FETCH_SIZE calibration on this library's read patterns (tools/micro/fetch_calib.hip) -> profiles/r6_fetch_calib.txt
#   bash tools/r6_calib.sh        (on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r6
cd /tmp && export TMPDIR=/tmp
$R/gvcnn-tf_amd/build/fetch_calib > $R/gpurun_out/r6/fetch_calib_run.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r6/calib_f -o p --output-format csv -- $R/gvcnn-tf_amd/build/fetch_calib > /dev/null 2> $R/gpurun_out/r6/calib_f.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $R/gpurun_out/r6/calib_r -o p --output-format csv -- $R/gvcnn-tf_amd/build/fetch_calib > /dev/null 2> $R/gpurun_out/r6/calib_r.err
cd $R
python3 - <<'PY' | tee gpurun_out/r6/fetch_calib.txt
import csv, glob
print(open("gpurun_out/r6/fetch_calib_run.txt").read())
want = {"calib_wide16": 1536 * 2**20, "calib_planes16": 1536 * 2**20, "calib_half64<0>": 768 * 2**20, "calib_half64<1>": 768 * 2**20,
        "calib_dword4": 1536 * 2**20}
for d, names in (("calib_f", ("FETCH_SIZE",)), ("calib_r", ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"))):
    fs = glob.glob("gpurun_out/r6/%s/**/p_counter_collection.csv" % d, recursive=True)
    if not fs:
        print(d, ": no counter file"); continue
    rows = list(csv.DictReader(open(fs[0])))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for cn in names:
        seen = {}
        for r in rows:
            if r["Counter_Name"] != cn: continue
            k = next((k for k in want if k.replace("<", "").replace(">", "") in r["Kernel_Name"].replace("<", "").replace(">", "")), None)
            if k: seen[k] = float(r["Counter_Value"])          # (the last launch of each)
        for k, v in seen.items():
            if cn == "FETCH_SIZE":
                print("%-16s FETCH_SIZE %12.0f KiB = %8.1f MB; bytes read %8.1f MB: reported / read = %.3f" % (k, v, v * 1024 / 1e6, want[k] / 1e6, v * 1024 / want[k]))
            else:
                print("%-16s %-22s %12.0f requests; bytes read / requests = %.1f" % (k, cn, v, want[k] / max(v, 1)))
PY
[ "$1" = "calib-only" ] && exit 0
# the same counters over the c2 step (single lane, tuned tiles): FETCH_SIZE next to the request-size split
O=gpurun_out/r6/pmc_req_c2
mkdir -p $R/$O
cd /tmp
python3 $R/bench.py --pmc-child tune --tile-cache $R/$O/tiles.json > /dev/null 2> $R/$O/tune.err
CMD="python3 $R/bench.py --pmc-child 1 --no-lanes --tile-cache $R/$O/tiles.json --steps 2 --warmup 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/f -o p --output-format csv -- $CMD > /dev/null 2> $R/$O/f.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $R/$O/q -o p --output-format csv -- $CMD > /dev/null 2> $R/$O/q.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/w -o p --output-format csv -- $CMD > /dev/null 2> $R/$O/w.err
cd $R
ls -la $O/q | head; tail -3 $O/q.err
