cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/m_pmc -o sq -- python3 $R/tools/spec_bench.py 256 6 > $R/gpurun_out/m_pmc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $R/gpurun_out/m_pmc2 -o sq -- python3 $R/tools/spec_bench.py 256 6 > $R/gpurun_out/m_pmc2.log 2>&1
cd $R
python3 - <<'PY'
import csv, collections, glob
for d in ("gpurun_out/m_pmc", "gpurun_out/m_pmc2"):
    f = glob.glob(d + "/*counter_collection.csv")
    if not f: print("no csv in", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float)
    seen=set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("spec_"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); cnt[k] += 1; dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for k in acc:
        print(k, "launches", cnt[k], "avg_us", round(dur[k] / cnt[k] / 1e3, 1))
        for c, v in acc[k].items(): print("   %-24s %.4g per launch" % (c, v / cnt[k]))
PY
rm -rf gpurun_out/m_pmc gpurun_out/m_pmc2
