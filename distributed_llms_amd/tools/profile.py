"""rocprofv3 wrapper: per-kernel time plus hardware counters for any script of this repository.

    python -m distributed_llms_amd.tools.profile --out profiles/10_bench -- bench.py --steps 2 --warmup 1
    python -m distributed_llms_amd.tools.profile --passes time,mfma,lds --out profiles/x -- bench/kernel_bench.py

Passes (each is its own rocprofv3 run - the profiler does not split counters over passes, and counter runs
are never combined with trace domains other than the kernel trace):

  time   --kernel-trace --stats               -> <out>_kernel_stats.csv      (calls, total ms, mean us, share)
  mfma   SQ busy / MFMA busy / MFMA bf16 ops  -> <out>_pmc_mfma.csv          (MFMA busy share per kernel)
  lds    LDS bank conflicts / LDS active      -> <out>_pmc_lds.csv           (conflict cycles per active cycle)
  fetch  FETCH_SIZE                           -> <out>_pmc_fetch.csv         (KB read from HBM/fabric per launch)
  write  WRITE_SIZE + L2 hit / miss           -> <out>_pmc_write.csv

The program is started as ``python3 <script> ...`` directly after ``--`` (no shell, no env wrapper: the
profiler's preloaded library initialises the GPU, so every extra exec hop is forbidden on this pool), from
/tmp with TMPDIR=/tmp.  This module itself never touches the GPU.
"""
from __future__ import annotations

import argparse
import csv
import glob
import os
import sqlite3
import subprocess
import sys
from collections import defaultdict
from pathlib import Path
from typing import Dict, List

REPO = Path(__file__).resolve().parents[2]

N_XCD, N_CU = 8, 256        # MI355X

PMC_SETS: Dict[str, List[str]] = {
    # <= 8 SQ counters, <= 2 GRBM per run
    "mfma": ["SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU",
             "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"],
    "lds": ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_LDS_ADDR_CONFLICT", "SQ_BUSY_CYCLES"],
    # FETCH_SIZE alone uses 3 of the 4 TCC slots; WRITE_SIZE uses 2
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"],
}


def _run(cmd: List[str], timeout: int, log: Path) -> int:
    env = dict(os.environ, TMPDIR="/tmp", HSA_ENABLE_IPC_MODE_LEGACY="0")
    with open(log, "w") as f:
        try:
            return subprocess.run(cmd, cwd="/tmp", env=env, stdout=f, stderr=subprocess.STDOUT, timeout=timeout).returncode
        except subprocess.TimeoutExpired:
            return 124


def kernel_stats_from_db(db_path: str) -> List[dict]:
    db = sqlite3.connect(db_path)
    rows = db.execute("SELECT name, COUNT(*), SUM(end - start) FROM kernels GROUP BY name ORDER BY 3 DESC").fetchall()
    total = sum(r[2] for r in rows) or 1
    return [{"kernel": r[0], "calls": r[1], "total_ms": r[2] / 1e6, "mean_us": r[2] / r[1] / 1e3, "pct": 100.0 * r[2] / total}
            for r in rows]


def pmc_by_kernel(csv_paths: List[str]) -> Dict[str, Dict[str, float]]:
    """{kernel: {counter: sum over launches, '_launches': n}} from rocprofv3 counter_collection CSVs."""
    acc: Dict[str, Dict[str, float]] = defaultdict(lambda: defaultdict(float))
    seen = defaultdict(set)
    for p in csv_paths:
        with open(p, newline="") as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name") or row.get("kernel_name") or "?"
                c = row.get("Counter_Name") or row.get("counter_name")
                v = row.get("Counter_Value") or row.get("counter_value") or "0"
                if c is None:
                    continue
                acc[k][c] += float(v)
                seen[k].add(row.get("Dispatch_Id") or row.get("dispatch_id"))
    for k in acc:
        acc[k]["_launches"] = float(len(seen[k]))
    return acc


def _short(name: str, n: int = 70) -> str:
    return name if len(name) <= n else name[: n - 3] + "..."


def write_pmc_summary(name: str, data: Dict[str, Dict[str, float]], out: Path) -> None:
    counters = PMC_SETS[name]
    derived = {"mfma": "mfma_busy_share", "lds": "bank_conflict_per_lds_cycle", "fetch": "fetch_kb_per_launch",
               "write": "l2_hit_rate"}[name]
    key = counters[0] if name in ("mfma", "lds") else counters[0]
    rows = sorted(data.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", kv[1].get(key, 0.0)))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", *counters, derived])
        for k, d in rows:
            if name == "mfma":
                # MFMA busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs: share of the
                # kernel's wall cycles in which an average SIMD had its MFMA pipe busy
                ga = d.get("GRBM_GUI_ACTIVE", 0.0) / N_XCD
                val = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (ga * 4 * N_CU) if ga else 0.0
            elif name == "lds":
                a = d.get("SQ_LDS_IDX_ACTIVE", 0.0)
                val = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / a if a else 0.0
            elif name == "fetch":
                val = d.get("FETCH_SIZE", 0.0) / max(d.get("_launches", 1.0), 1.0)
            else:
                h, m = d.get("TCC_HIT_sum", 0.0), d.get("TCC_MISS_sum", 0.0)
                val = h / (h + m) if h + m else 0.0
            w.writerow([_short(k), int(d.get("_launches", 0)), *[f"{d.get(c, 0.0):.0f}" for c in counters], f"{val:.4f}"])


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", default="profiles/profile", help="output prefix (tracked summaries)")
    ap.add_argument("--scratch", default="gpurun_out/profile_raw", help="raw rocprofv3 output directory")
    ap.add_argument("--passes", default="time,mfma,lds,fetch")
    ap.add_argument("--timeout", type=int, default=240, help="seconds per pass")
    ap.add_argument("cmd", nargs=argparse.REMAINDER, help="-- script.py [args]")
    args = ap.parse_args(argv)
    cmd = [c for c in args.cmd if c != "--"]
    if not cmd:
        ap.error("give the script to profile after --")
    script = str((REPO / cmd[0]).resolve()) if not os.path.isabs(cmd[0]) else cmd[0]
    prog = [sys.executable if os.path.basename(sys.executable).startswith("python") else "python3", script, *cmd[1:]]
    out = (REPO / args.out) if not os.path.isabs(args.out) else Path(args.out)
    out.parent.mkdir(parents=True, exist_ok=True)
    raw = (REPO / args.scratch) if not os.path.isabs(args.scratch) else Path(args.scratch)
    raw.mkdir(parents=True, exist_ok=True)
    rc_all = 0
    for name in [p.strip() for p in args.passes.split(",") if p.strip()]:
        d = raw / name
        log = raw / f"{name}.log"
        if name == "time":
            rc = _run(["rocprofv3", "--kernel-trace", "--stats", "-d", str(d), "--", *prog], args.timeout, log)
            dbs = glob.glob(str(d / "**" / "*_results.db"), recursive=True)
            if rc == 0 and dbs:
                rows = kernel_stats_from_db(dbs[0])
                with open(f"{out}_kernel_stats.csv", "w", newline="") as f:
                    w = csv.writer(f)
                    w.writerow(["kernel", "calls", "total_ms", "mean_us", "pct"])
                    for r in rows:
                        w.writerow([r["kernel"], r["calls"], f"{r['total_ms']:.3f}", f"{r['mean_us']:.2f}", f"{r['pct']:.2f}"])
                print(f"[profile] time: {len(rows)} kernels -> {out}_kernel_stats.csv", flush=True)
        elif name in PMC_SETS:
            rc = _run(["rocprofv3", "--pmc", *PMC_SETS[name], "--output-format", "csv", "-d", str(d), "--", *prog],
                      args.timeout, log)
            csvs = glob.glob(str(d / "**" / "*counter_collection.csv"), recursive=True)
            if rc == 0 and csvs:
                write_pmc_summary(name, pmc_by_kernel(csvs), Path(f"{out}_pmc_{name}.csv"))
                print(f"[profile] {name}: -> {out}_pmc_{name}.csv", flush=True)
        else:
            print(f"[profile] unknown pass {name!r}", flush=True)
            rc = 2
        if rc != 0:
            print(f"[profile] pass {name} failed with {rc}; see {log}", flush=True)
            rc_all = rc
            break     # after a failed or timed-out GPU step nothing more is started
    return rc_all


if __name__ == "__main__":
    raise SystemExit(main())
