#!/usr/bin/env python3
"""Digest rocprofv3 CSV output (gpurun_out/prof_<tag>_*) into profiles/<tag>_summary.{md,json}.

    python tools/profile_summary.py r01 [kernel-substring]
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    kern = sys.argv[2] if len(sys.argv) > 2 else "crt_trace_kernel<false, false, false, false, false>"   # <COUNT, STAMP, SHADOW, TLAS, REFRACT>: the kernel of the timed region
    base = os.path.join(ROOT, "gpurun_out")
    out = {"tag": tag, "kernel": kern, "kernel_stats": [], "counters": {}, "bench_line": None}
    def newest(pattern):
        fs = glob.glob(pattern)
        return [max(fs, key=os.path.getmtime)] if fs else []
    for f in newest(os.path.join(base, f"prof_{tag}_stats", "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            out["kernel_stats"].append({"name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                        "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])})
    # launch overlap (frames in flight): per-launch durations vs the union of the launch intervals, from the trace itself
    for f in newest(os.path.join(base, f"prof_{tag}_stats", "*", "*_kernel_trace.csv")):
        iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"])
        if iv:
            union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
            for a, b in iv[1:]:
                if a > cur_e:
                    union += cur_e - cur_s; cur_s, cur_e = a, b
                else:
                    cur_e = max(cur_e, b)
            union += cur_e - cur_s
            total = sum(b - a for a, b in iv)
            out["launch_overlap"] = {"launches": len(iv), "mean_launch_us": total / len(iv) / 1e3, "busy_us_per_launch": union / len(iv) / 1e3,
                                     "mean_launches_in_flight": total / max(1, union)}
    log = os.path.join(base, f"prof_{tag}_stats.log")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith('{"metric"'):
                out["bench_line"] = json.loads(line)
    meta = {}
    for d in sorted(glob.glob(os.path.join(base, f"prof_{tag}_*"))):
        for f in newest(os.path.join(d, "*", "*_counter_collection.csv")):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
            for k, v in acc.items():
                out["counters"][k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    out["dispatch"] = meta
    c = {k: v["mean_per_launch"] for k, v in out["counters"].items()}
    d = {}
    if "FETCH_SIZE" in c:
        d["hbm_read_MB_raw_FETCH_SIZE"] = c["FETCH_SIZE"] / 1024.0
        d["hbm_read_MB_x2_gfx950_correction"] = 2 * c["FETCH_SIZE"] / 1024.0
    if "WRITE_SIZE" in c:
        d["hbm_write_MB"] = c["WRITE_SIZE"] / 1024.0
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
        d["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / max(1.0, c["SQ_ACTIVE_INST_VALU"] * 64.0)
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        # a wave64 VALU instruction holds its SIMD-32 for 2 cycles (MI355X_MICROARCH.md: constants table); GRBM_GUI_ACTIVE is
        # summed over the 8 XCDs. (r02 first divided the wave-centric SQ_ACTIVE_INST_VALU by the SIMD cycles and read 0.83: that
        # counter advances 4 cycles per instruction per WAVE, twice what the instruction costs the SIMD.)
        kcyc = c["GRBM_GUI_ACTIVE"] / 8.0
        d["kernel_cycles_profiled_launch"] = kcyc
        d["valu_issue_busy"] = c["SQ_INSTS_VALU"] * 2.0 / (1024.0 * kcyc)
        d["valu_insts_per_launch"] = c["SQ_INSTS_VALU"]
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
            d["l1_line_accesses_per_launch"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"]
            d["l1_line_accesses_per_cycle_per_cu"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (256.0 * kcyc)
        for k, name in (("TA_TA_BUSY_sum", "ta_busy"), ("TD_TD_BUSY_sum", "td_busy"), ("TCP_GATE_EN1_sum", "tcp_clocked"),
                        ("TCP_PENDING_STALL_CYCLES_sum", "tcp_pending_stall"), ("TA_ADDR_STALLED_BY_TC_CYCLES_sum", "ta_addr_stalled_by_tc")):
            if k in c:
                d[name] = c[k] / (256.0 * kcyc)
    if "SQ_INSTS_VMEM_RD" in c:
        d["vmem_rd_insts_per_launch"] = c["SQ_INSTS_VMEM_RD"]
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
            d["l1_lines_per_vmem_rd_inst"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / max(1.0, c["SQ_INSTS_VMEM_RD"])
        if "GRBM_GUI_ACTIVE" in c:
            # tools/ubench/chain.hip: a vector load costs the CU's address pipeline ~12-16 cycles + ~0.5 cycle per distinct line
            kcyc = c["GRBM_GUI_ACTIVE"] / 8.0
            d["vmem_rd_insts_per_cu_per_kcycle"] = c["SQ_INSTS_VMEM_RD"] / 256.0 / kcyc * 1e3
    if "SQ_INSTS_SMEM" in c:
        d["smem_insts_per_launch"] = c["SQ_INSTS_SMEM"]
    if "SQ_WAVE_CYCLES" in c and "SQ_WAVES" in c:
        d["wave_quad_cycles_per_wave"] = c["SQ_WAVE_CYCLES"] / max(1.0, c["SQ_WAVES"])
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        d["wave_parked_fraction"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
        d["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
    if "TCP_TCC_READ_REQ_sum" in c and "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
        d["l1_hit_rate"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / max(1.0, c["TCP_TOTAL_CACHE_ACCESSES_sum"])
    out["derived"] = d
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_summary.json"), "w"), indent=1)
    with open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w") as f:
        f.write(f"# rocprofv3 summary `{tag}` (MI355X, `python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras` + the PROF_ARGS of the run, see the bench line)\n\n")
        if out["bench_line"]:
            b = out["bench_line"]
            f.write(f"bench line under the profiler: {b['value']} {b['unit']}, {b['ms_per_step']} ms/frame, workload `{b['config']['workload']}`\n\n")
        f.write("## --kernel-trace --stats\n\n| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
        for k in sorted(out["kernel_stats"], key=lambda r: -r["pct"]):
            f.write(f"| `{k['name'][:90]}` | {k['calls']} | {k['avg_ns'] / 1e3:.1f} | {k['min_ns'] / 1e3:.1f} | {k['max_ns'] / 1e3:.1f} | {k['pct']:.2f} |\n")
        if out.get("launch_overlap") and out["bench_line"]:
            km = out["bench_line"].get("kernel_ms", {})
            f.write(f"\nThe same run's own HIP-event figures (bench.py, `kernel_ms`): mean launch duration {km.get('crt_trace_kernel_launch_mean', 0) * 1e3:.1f} us, "
                    f"device time per frame {km.get('device_time_per_frame', 0) * 1e3:.1f} us -- to be read against the trace's figures in the next paragraph "
                    f"(the trace also holds the 3 warm-up launches and the instrumented launch of another instantiation).\n")
        if out.get("launch_overlap"):
            o = out["launch_overlap"]
            f.write(f"\n`{kern}` launches in the trace: {o['launches']}, mean launch duration {o['mean_launch_us']:.1f} us, device time with at least one "
                    f"launch running {o['busy_us_per_launch']:.1f} us per launch, mean launches in flight {o['mean_launches_in_flight']:.2f} "
                    f"(bench.py: `roofline.launch_duration_ms` / `kernel_ms.device_time_per_frame` / `launches_in_flight`).\n")
        f.write(f"\n## PMC passes (mean per launch of `{kern}`; one counter group per run, 5-step runs)\n\ndispatch: {meta}\n\n| counter | mean per launch | launches |\n|---|---|---|\n")
        for k in sorted(out["counters"]):
            f.write(f"| {k} | {out['counters'][k]['mean_per_launch']:.4g} | {out['counters'][k]['launches']} |\n")
        f.write("\n## derived\n\n")
        for k, v in d.items():
            f.write(f"- {k}: {v:.4g}\n")
        f.write("\nFETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x "
                "(MI355X_MICROARCH.md, HBM section), so both the raw and the doubled figure are listed. SQ_* cycle counters are in quad-cycles.\n")
    print(json.dumps(d, indent=1))


if __name__ == "__main__":
    main()
