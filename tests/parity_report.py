#!/usr/bin/env python3
"""Turn the records tests/gpu_util.py:assert_parity leaves behind (gpurun_out/parity_records.jsonl)
into the per-stage table committed under profiles/ (TEST INFRASTRUCTURE).

  python tests/parity_report.py [records.jsonl] > profiles/parity_rNN.md
"""

import json
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fmt(v):
    return "—" if v is None else f"{v:.2e}"


# why a stage states a tolerance above the north star's 1e-5 (first match wins; keyed on test / stage text)
REASONS = [
    ("share of trajectories", "not an error figure: the share of trajectories whose control flow is decided (fp32 / fp64 / "
                              "perturbed fp32 oracles agree, no criterion within 2 % of its threshold); `tol` = the share required"),
    ("Hessian solve residual", "backward error of the structured Hessian solve against the fp64 operator: the m x m pivoted "
                                 "solves inherit cond(R) ~ 1e4 (alpha = 1e-2 in the stage cost); bar 1e-4 or 10 x the fp32 oracle's"),
    ("Hessian-solve residual", "as above, at the full shapes"),
    ("gradient response", "information only: what a right-hand-side perturbation of the size of HIP's residual does to the "
                          "gradient in fp64 (sets the end-to-end bar)"),
    ("round-2 fixed bar", "the round-2 fixed bar of the end-to-end gradient, kept as a recorded check (information)"),
    ("bilevel grad end-to-end", "forward error of the gradient = the Hessian solve's backward error seen through cond(A): bar "
                                "= max(1e-4, 4 x the fp64 response to a residual of HIP's size), never above 1e-3; the round-2 "
                                "fixed 1e-4 bar is recorded beside it"),
    ("bilevel Bvec", "loss adjoint at the full shapes through T = 50 .. 100 Jacobian products; achieved 1e-5 .. 1e-4"),
    ("tangent roll", "dX rolled forward from the GPU's own H through T Jacobians (fp64 roll of the same H as reference)"),
    ("cost_vjp", "sum over the batch of per-trajectory vector-Jacobian products taken at the GPU's own (H, dX)"),
    ("(<= 1 iteration)", "one iLQR step: the iterate carries the Riccati gains' conditioning-limited error (cond(R) ~ 1e4: the "
                         "NumPy fp32 oracle shows the same size), GAIN_CEILING applies"),
    ("ilqr ", "one iLQR iteration at the full shapes: gains' conditioning (see gain rows), bar 1e-4 + the 4 x fp32-oracle rule"),
    ("ls16 ", "iterate after 1 - 2 iLQR iterations through the 16-candidate line search: gains' conditioning"),
    (" obj", "objective after several chaotic, ill-conditioned Newton steps (1 .. 12 iterations); bar 1e-5 .. 3e-4 by scenario"),
    ("test_ilqr_single_iteration_teacher_forced", "one iLQR iteration from identical starts: gains' conditioning"),
    ("test_loss_and_grad", "batch-mean bilevel gradient through the host mirror: the Hessian solve's conditioning"),
    ("lstm-dynamics", "LSTM dynamics variant, whole policy step: gains' and Hessian solve's conditioning through the smooth cell"),
]


def reason(r):
    key = r.get("test", "") + " " + r["stage"]
    for pat, why in REASONS:
        if pat in key:
            return why
    return "chained fp32 stage downstream of the Riccati gains / Hessian solve (conditioning-limited)"


def loose_table(recs):
    loose = [r for r in recs if r["tol"] > 1.0001e-5 and "share of trajectories" not in r["stage"]]
    print("## Assertions whose STATED tolerance is above 1e-5\n")
    print(f"{len(loose)} of {len(recs)} assertions state a tolerance above the north star's 1e-5; grouped by test and stage "
          "(`n`: assertions in the group, over shapes / parameters; errors: the group's maxima).\n")
    print("| test | stage | n | stated tol | max e_hip | max e_o32 | why the stage cannot be asked for 1e-5 |")
    print("|---|---|---:|---:|---:|---:|---|")
    groups = OrderedDict()
    for r in loose:
        t = r.get("test", "").split("::")[-1].split("[")[0]
        groups.setdefault((t, r["stage"]), []).append(r)
    for (t, stage), rows in groups.items():
        tols = sorted({float(f"{x['tol']:.1e}") for x in rows})
        tol_s = fmt(tols[0]) if len(tols) == 1 else f"{fmt(tols[0])} .. {fmt(tols[-1])}"
        print(f"| {t} | {stage} | {len(rows)} | {tol_s} | {fmt(max(x['e_hip'] for x in rows))} | "
              f"{fmt(max((x.get('e_o32') or 0.0) for x in rows))} | {reason(rows[0])} |")
    print()
    shares = [r for r in recs if "share of trajectories" in r["stage"]]
    if shares:
        print("## Control-flow tests: share of trajectories compared\n")
        print("| scenario | observed share | required |")
        print("|---|---:|---:|")
        for r in shares:
            print(f"| {r['stage'].split(':')[0]} ({r.get('config', '')}) | {r['e_hip']:.2f} | {r['tol']:.2f} |")
        print()


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_records.jsonl")
    recs = [json.loads(line) for line in open(path) if line.strip()]
    # the last record of a (test, stage, config) wins: a re-run replaces the earlier one
    last = OrderedDict()
    for r in recs:
        last[(r.get("test", ""), r["stage"], r.get("config", ""))] = r
    recs = list(last.values())
    on_tol = sum(1 for r in recs if r["branch"] == "tol")
    print("# Parity report: HIP fp32 vs the fp64 oracle, per assertion\n")
    print("Produced by `tests/parity_report.py` from the records every `assert_parity` call of the `-m gpu` "
          "suite appends (`tests/gpu_util.py`).  `e_hip` / `e_o32`: max-norm relative error of the HIP result / "
          "of the NumPy fp32 oracle against the fp64 oracle.  `branch = tol`: HIP is within the stated tolerance "
          "of fp64 (the north star's 1e-5 unless the column says otherwise); `branch = slack`: it passed as "
          "\"no worse than 4x the fp32 oracle's own error\", capped at 1e-3 (1e-2 for the Riccati gains and the iLQR iterate, whose forward error carries cond(G) ~ 1e4, see tests/gpu_util.py GAIN_CEILING; their backward error is reported beside them as \"gain equation residual\").  `el_*`: largest per-entry relative "
          "error with the denominator floored at 1e-6 x max|ref| (p99.9 beside it).\n")
    info = [r for r in recs if r["branch"] == "info"]
    n_slack = sum(1 for r in recs if r["branch"] == "slack")
    print(f"{len(recs)} records: {on_tol} assertions on the tolerance branch, {n_slack} on the slack branch, "
          f"{sum(1 for r in recs if r['branch'] != 'info' and not r.get('passed', True))} failed; {len(info)} recorded for "
          f"information (not asserted), of which {sum(1 for r in info if not r.get('passed', True))} are above their "
          "recorded bar (the round-2 fixed 1e-4 bar of the end-to-end bilevel gradient: listed as **above bar** below).\n")
    loose_table(recs)
    by_cfg = OrderedDict()
    for r in recs:
        by_cfg.setdefault(r.get("config", ""), []).append(r)
    for cfg, rows in by_cfg.items():
        print(f"## {cfg}\n")
        print("| test | stage | e_hip | e_o32 | tol | tol_used | branch | el_hip (p99.9) | el_o32 (p99.9) | entries |")
        print("|---|---|---:|---:|---:|---:|---|---:|---:|---:|")
        for r in rows:
            test = r.get("test", "").split("::")[-1]
            el_h = f"{fmt(r.get('el_hip'))} ({fmt(r.get('el_hip_p999'))})" if "el_hip" in r else "—"
            el_o = f"{fmt(r.get('el_o32'))} ({fmt(r.get('el_o32_p999'))})" if "el_o32" in r else "—"
            flag = "" if r.get("passed", True) else (" **above bar**" if r["branch"] == "info" else " **FAILED**")
            print(f"| {test} | {r['stage']}{flag} | {fmt(r['e_hip'])} | {fmt(r.get('e_o32'))} | {fmt(r['tol'])} | "
                  f"{fmt(r['tol_used'])} | {r['branch']} | {el_h} | {el_o} | {r.get('entries', '')} |")
        print()


if __name__ == "__main__":
    main()
