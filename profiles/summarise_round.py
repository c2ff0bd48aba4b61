"""Turn the output of profile_round.sh (gpurun_out/prof_<v>/) into the artefacts kept under profiles/:
<round>_<v>_bench.json, <round>_<v>_bench_kernel_stats.csv, <round>_<v>_bench_under_rocprof.json,
<round>_<v>_pmc_summary.md and traffic_latest.json.
usage: python profiles/summarise_round.py v12 [r01]      (round prefix, default r01; the profile directory is
gpurun_out/prof_<v> for r01 and gpurun_out/prof_<round><v> otherwise)"""
import json
import shutil
import sys

v = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
src = f"gpurun_out/prof_{v}" if rnd == "r01" else f"gpurun_out/prof_{rnd}{v}"
if not __import__("os").path.isdir(src):           # profile_round.sh r03_v1 -> gpurun_out/prof_r03_v1
    src = f"gpurun_out/prof_{rnd}_{v}"
shutil.copy(f"{src}/bench.json", f"profiles/{rnd}_{v}_bench.json")
shutil.copy(f"{src}/kernel_stats.csv", f"profiles/{rnd}_{v}_bench_kernel_stats.csv")
shutil.copy(f"{src}/bench_under_rocprof.json", f"profiles/{rnd}_{v}_bench_under_rocprof.json")
r = json.load(open(f"{src}/pmc_per_launch.json"))
rows = sorted(r, key=lambda k: -r[k].get("GRBM_GUI_ACTIVE", 0))[:10]
out = [f"# {rnd} {v} PMC summary (rocprofv3 --pmc, three separate passes; per launch averages)", "",
       "Workload: `bench.py --steps 3 --warmup 1 --no-cpu-baseline --secondary-maxiter 0` (C3 shape, B=1024), one",
       "`rocprofv3 --kernel-trace --pmc <counters> --output-format csv` run per counter set (FETCH_SIZE;",
       "WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE), collected by `profiles/profile_round.sh`.",
       "FETCH_SIZE / WRITE_SIZE are KiB; MFMA-busy % = MFMA_BUSY / (1,024 SIMDs x GUI_ACTIVE / 8 XCDs).", "",
       "| kernel | FETCH KiB | WRITE KiB | MFMA_BUSY cyc (sum SIMDs) | GUI_ACTIVE (sum XCDs) | MFMA-busy % |",
       "|---|---:|---:|---:|---:|---:|"]
for k in rows:
    c = r[k]
    busy = 100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
    out.append(f"| `{k}` | {c.get('FETCH_SIZE', 0):.0f} | {c.get('WRITE_SIZE', 0):.0f} | "
               f"{c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.4g} | {c['GRBM_GUI_ACTIVE']:.4g} | {busy:.1f} |")
lin = r[[k for k in r if k.startswith("k_linearize_regs")][0]]
traffic = int((2 * lin["FETCH_SIZE"] + lin["WRITE_SIZE"]) * 1024)
out += ["", f"Dominant kernel HBM traffic = (2 x {lin['FETCH_SIZE']:.0f} + {lin['WRITE_SIZE']:.0f}) KiB = "
        f"**{traffic / 1e6:.1f} MB per launch** (FETCH_SIZE doubled: 16 B/lane loads on gfx950) against 84.7 MB",
        "algorithmic (`AB` written once, masks and weights read once)."]
open(f"profiles/{rnd}_{v}_pmc_summary.md", "w").write("\n".join(out) + "\n")
json.dump({"k_linearize_hbm_bytes_per_launch": traffic,
           "source": f"profiles/{rnd}_{v}_pmc_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                     "FETCH_SIZE doubled: 16 B/lane loads on gfx950)",
           "algorithmic_bytes_per_launch": 84700000}, open("profiles/traffic_latest.json", "w"), indent=1)
print("\n".join(out))
