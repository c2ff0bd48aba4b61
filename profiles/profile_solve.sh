#!/bin/bash
# usage: bash profiles/profile_solve.sh <tag>   (on the GPU box, from the repo root)
# MFMA-busy of the kernels of the complete iLQR solve (bench.py's secondary.solve: maxiter 100, 1024
# trajectories): one rocprofv3 --pmc pass with kernel trace only; the program itself follows `--`.
set -e
TAG=$1
R=$PWD
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-large-state --secondary-maxiter 1 --solve-maxiter 100"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -- python3 $R/bench.py $ARGS > $O/bench.json 2> $O/pmc.err
cd $R
python - <<PY
import glob, csv, collections, re
O = "$O"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
rows = []
for k, cs in acc.items():
    gui = sum(cs.get("GRBM_GUI_ACTIVE", [0])); busy = sum(cs.get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))
    n = len(cs.get("GRBM_GUI_ACTIVE", []))
    if gui > 0: rows.append((gui, k, n, busy))
rows.sort(reverse=True)
out = ["| kernel | launches | GUI_ACTIVE (sum XCDs, all launches) | MFMA_BUSY cyc (sum SIMDs) | MFMA-busy % |", "|---|---:|---:|---:|---:|"]
for gui, k, n, busy in rows[:10]:
    out.append(f"| \`{k}\` | {n} | {gui:.4g} | {busy:.4g} | {100 * busy / (1024 * gui / 8):.1f} |")
open(O + "/pmc_summary_table.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
