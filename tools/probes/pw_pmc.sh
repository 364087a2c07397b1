# rocprofv3 evidence for the flat 1x1 kernel: kernel-trace stats and the two HBM counters (separate passes) on tools/probes/pw_one.py
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pw_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export IDEAS_B3_PW=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/probes/pw_one.py > $O/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python $R/tools/probes/pw_one.py > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python $R/tools/probes/pw_one.py > $O/write.log 2>&1
python - <<PY
import csv, glob, collections
O="$O"
st = glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(st)):
    if "conv_b3_pw" in r["Name"]:
        print("stats", r["Name"][:90], r["Calls"], "avg us", float(r["AverageNs"]) / 1e3)
for what in ("fetch", "write"):
    f = glob.glob(O + "/%s/**/*counter_collection.csv" % what, recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "conv_b3_pw_kernel" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:70], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(what, k, len(v), "avg KB", sum(v) / len(v))
PY
