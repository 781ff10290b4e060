#!/usr/bin/env python3
"""Digest tools/ubench_pmc.sh output: per variant of the chain microbenchmark (the 512-hop launches), unit-busy fractions and instruction counts.
    python tools/ubench_pmc_summary.py r03  -> profiles/<tag>_ubench_chain_pmc.md"""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
transposed = len(sys.argv) > 2 and sys.argv[2] == "--brief-transposed"      # the round-4 pairs (chain_masked_kernel launches)
acc = collections.defaultdict(dict)          # (kernel, launch index among that kernel's big launches) -> counter -> value
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_ub*"))):
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        seen = collections.Counter()
        rows = list(csv.DictReader(open(f)))
        # one row per (dispatch, counter); dispatches come in launch order: warm-up (16 hops) then the timed one (512 hops), per variant
        by_disp = collections.OrderedDict()
        for r in rows:
            by_disp.setdefault(r["Dispatch_Id"], []).append(r)
        k = 0
        for disp, rs in by_disp.items():
            name = rs[0]["Kernel_Name"]
            if ("chain_masked_kernel" if transposed else "chain_kernel") not in name or "true>" in name:      # (the self-check launch of a transposed variant)
                continue
            k += 1
            if k % 2 == 1:      # the warm-up launch of each pair
                continue
            for r in rs:
                acc[(name, k // 2)][r["Counter_Name"]] = float(r["Counter_Value"])
labels = {1: "64 lanes, a record each", 2: "28 lanes, a record each", 3: "64 lanes, 16 lanes per chain (4 records per load)"}
if transposed:
    labels = {1: "per-lane fetch, 64 lanes", 2: "quad-transposed fetch, 64 lanes", 3: "per-lane fetch, 28 scattered lanes", 4: "quad-transposed fetch, 28 scattered lanes"}
out = [f"# PMC passes over the chain microbenchmark (`tools/ubench/chain {'--brief-transposed' if transposed else '--brief'}`, kernel mix, 8 waves/SIMD; `tools/ubench_pmc.sh {tag}`)\n",
       "Fractions are busy cycles / (CUs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs; a profiled launch runs at a lower clock than an unprofiled one.\n",
       "| variant | kernel cycles | TA busy | TD busy | L1 pending-stall | L1 line accesses per vector load | vector loads per CU per kcycle | cycles per vector load per CU | waves parked | VALU instructions per vector load | VALU instructions x 4 cycles / (4 SIMDs x kernel cycles): >= 1 = issue-bound |", "|---|---|---|---|---|---|---|---|---|---|---|"]
for (name, idx), c in sorted(acc.items(), key=lambda kv: kv[0][1]):
    if "GRBM_GUI_ACTIVE" not in c:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    f = lambda k: (c[k] / (256.0 * cyc)) if k in c else float("nan")
    vm = c.get("SQ_INSTS_VMEM_RD", float("nan"))
    out.append(f"| {labels.get(idx, name)} | {cyc:,.0f} | {f('TA_TA_BUSY_sum'):.2f} | {f('TD_TD_BUSY_sum'):.2f} | {f('TCP_PENDING_STALL_CYCLES_sum'):.2f} | "
               f"{c.get('TCP_TOTAL_CACHE_ACCESSES_sum', float('nan')) / vm:.1f} | {vm / 256.0 / cyc * 1e3:.1f} | {256.0 * cyc / vm:.1f} | {c.get('SQ_WAIT_ANY', float('nan')) / c.get('SQ_WAVE_CYCLES', float('nan')):.2f} | {c.get('SQ_INSTS_VALU', float('nan')) / vm:.1f} | {c.get('SQ_INSTS_VALU', float('nan')) * 4.0 / 4.0 / (256.0 * cyc):.2f} |")
import json
rows = []
for (name, idx), c in sorted(acc.items(), key=lambda kv: kv[0][1]):
    if "GRBM_GUI_ACTIVE" in c and "SQ_INSTS_VMEM_RD" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        rows.append({"variant": labels.get(idx, name), "kernel_cycles": cyc, "ta_busy": c.get("TA_TA_BUSY_sum", 0) / (256.0 * cyc), "td_busy": c.get("TD_TD_BUSY_sum", 0) / (256.0 * cyc),
                     "l1_lines_per_vector_load": c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / c["SQ_INSTS_VMEM_RD"], "cu_cycles_per_vector_load": 256.0 * cyc / c["SQ_INSTS_VMEM_RD"]})
suffix = "_transposed" if transposed else ""
if len(rows) >= 2 and not transposed:
    # least-squares line through (lines per load, cycles per load): what a wave-level vector load costs the CU's vector-memory path when it is saturated
    xs = [r["l1_lines_per_vector_load"] for r in rows]; ys = [r["cu_cycles_per_vector_load"] for r in rows]
    n = len(xs); mx, my = sum(xs) / n, sum(ys) / n
    b = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
    a = my - b * mx
    out.append(f"\nLeast-squares line through the three rows: a wave-level vector load costs a saturated vector-memory path **{a:.1f} cycles + {b:.2f} per distinct L1 line** "
               f"(TA and TD busy 86-98 % in all three: the ceiling of `roofline.chain` is this path saturating, not latency).")
    json.dump({"rows": rows, "fit": {"cycles_per_load": a, "cycles_per_line": b}}, open(os.path.join(ROOT, "profiles", f"{tag}_ubench_chain_pmc.json"), "w"), indent=1)
open(os.path.join(ROOT, "profiles", f"{tag}_ubench_chain{suffix}_pmc.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
