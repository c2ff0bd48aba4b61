#!/bin/bash
# usage: bash profiles/profile_round.sh vNN   (on the GPU box, from the repo root)
set -e
V=$1
R=$PWD
O=$R/gpurun_out/prof_$V   # copy what is to be kept into profiles/ afterwards
mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json     # the driver's contract command
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-large-state > $O/bench_under_rocprof.json 2> $O/kt.err
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  D=$O/pmc_$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-large-state --secondary-maxiter 0 > $D.json 2> $D.err
done
cd $R
python - <<PY
import glob, csv, collections, json, re
O = "$O"
ks = glob.glob(O + "/kt/**/*kernel_stats.csv", recursive=True)
import shutil
if ks: shutil.copy(ks[0], O + "/kernel_stats.csv")
def short(n):
    n = re.sub(r"\(.*", "", n); n = n.replace("void ", "")
    return n
res = collections.defaultdict(dict)
for d in glob.glob(O + "/pmc_*/"):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                res[k][c] = sum(v) / len(v)
json.dump(res, open(O + "/pmc_per_launch.json", "w"), indent=1)
for k, cs in sorted(res.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:12]:
    print(k, {c: round(v, 1) for c, v in cs.items()})
PY
