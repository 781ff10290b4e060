"""measure.py -- measurement aids of bench.py: SURVEY.md 8d's algorithmic bytes, the committed rocprofv3 summaries under profiles/
(HBM traffic, hit rates, issue-side counters), the live PMC passes, the microbenchmark ceilings, host CPU facts.
Nothing here renders or times a frame; bench.py (the contract's timed region) and the tools import it. (Round 5: moved out of bench.py.)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def algorithmic_bytes(c, pixels):
    """SURVEY.md 8d, layout-independent bytes from the work counters of one frame."""
    per_ray = 64 * c["innerVisits"] + 48 * c["triTests"] + 80 * c["traversals"]
    per_hit = (80 + 16 + 80 + 2 * 16 + 2 * 3) * c["hits"]
    per_miss = (16 + 3) * c["misses"]
    per_pixel = 16 * pixels  # float4 output write; RayGen is fused, so no 12 B ray write + read
    return per_ray + per_hit + per_miss + per_pixel


# tools/ubench/gather.hip on MI355X (profiles/r02_ubench_gather.txt), 64-B-per-lane record fetches as 4 x 16-B loads:
#  * every lane of a wave reads the SAME L1-resident record: 17 cycles per wave-level fetch per CU -> 3.76 records per
#    cycle per CU. Nothing a traversal does can beat that: the ceiling `gather.frac` is taken against.
#  * every lane reads a DIFFERENT record of a table resident in L2: 181 cycles -> 0.354 records per cycle per CU; a
#    reference point, not a bound -- coherent packets (many lanes on one line, L1 hits) legitimately run above it.
GATHER_CEILING_UNIFORM = 64.0 / 17.0
GATHER_DIVERGENT_L2 = 64.0 / 181.0


def chain_ceiling():
    """The dependent-gather ceiling from the committed run of tools/ubench/chain.hip (newest round first): records per cycle
    per CU at 8 waves/SIMD and the trace kernel's cache-hit mix, with 64 and with 28 chasing lanes per wave."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_ubench_chain.json")), reverse=True):
        try:
            runs = json.load(open(path))["runs"]
            pick = lambda lanes: [r for r in runs if r["mix"].startswith("kernel mix") and r.get("variant", "4 x dwordx4").startswith("4 x dwordx4 (")
                                  and r["waves_per_simd"] == 8 and r["active_lanes"] == lanes]
            full, part = pick(64), pick(28)
            if full:
                return {"full": full[0]["records_per_cycle_per_cu"], "lanes28": part[0]["records_per_cycle_per_cu"] if part else None,
                        "clock_ghz": full[0]["clock_ghz"], "source": os.path.relpath(path, ROOT), "mix": full[0]["mix"],
                        "hot": full[0].get("hot"), "warm": full[0].get("warm")}
        except Exception:
            continue
    return None


def usable_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box gives one GPU's share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


COUNTER_KEYS = ["rays", "primary", "secondary", "hits", "misses", "traversals", "pops", "innerVisits", "triTests", "shadowRays"]


def aggregate(dist, cnt, own_pixels, elapsed_s, kernel_ms_mean, device, group=None):
    """Whole-job totals: SUM of the per-rank work counters / pixels / algorithmic bytes, MAX of the per-rank times.
    `dist` is torch.distributed (initialised) or None for a single process. No pixel data is exchanged."""
    import torch
    vec = torch.tensor([float(cnt[k]) for k in COUNTER_KEYS] + [float(own_pixels), float(algorithmic_bytes(cnt, own_pixels))],
                       dtype=torch.float64, device=device)
    tmax = torch.tensor([float(elapsed_s), float(kernel_ms_mean)], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
    tot = dict(zip(COUNTER_KEYS + ["pixels", "alg_bytes"], vec.tolist()))
    return tot, tmax[0].item(), tmax[1].item()


def pmc_valu(kernel, workload_scene, width, height, dev_s, clock_ghz, num_cus):
    """Issue-side accounting of the dominant kernel from the committed PMC passes (profiles/*_summary.json): VALU
    instructions and L1 line accesses per launch are properties of the work, so they are put over THIS run's device time per
    launch; the fractions measured inside the (serialised, slower) profiled launch are passed through as they are."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")), reverse=True):
        try:
            d = json.load(open(path))
            dv = d.get("derived", {})
            if _summary_matches(d, kernel, workload_scene, width, height, False) and "valu_insts_per_launch" in dv:
                cyc = dev_s * clock_ghz * 1e9                     # cycles of device time per launch in this run
                out = {"issue_busy": round(dv["valu_insts_per_launch"] * 2.0 / (4.0 * num_cus * cyc), 3),
                       "issue_busy_profiled_launch": round(dv["valu_issue_busy"], 3),
                       "lane_utilisation": round(dv["valu_lane_utilisation"], 3),
                       "source": os.path.relpath(path, ROOT),
                       "note": "issue_busy = SQ_INSTS_VALU x 2 cycles (a wave64 instruction on a SIMD-32) / (SIMDs x this run's device cycles "
                               "per launch at the measured clock, roofline.chain.clock_ghz); lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)"}
                if "l1_line_accesses_per_launch" in dv:
                    out["l1_line_accesses_per_cycle_per_cu"] = round(dv["l1_line_accesses_per_launch"] / (num_cus * cyc), 3)
                    out["l1_divergent_ceiling"] = round(256.0 / 181.0, 3)     # tools/ubench/gather.hip: 64 lanes x 4 loads on 64 distinct L2-resident lines in 181 cycles
                for k in ("ta_busy", "td_busy", "tcp_pending_stall"):
                    if k in dv:
                        out[k + "_profiled_launch"] = round(dv[k], 3)
                return out
        except Exception:
            continue
    return None


def pmc_vmem(kernel, workload_scene, width, height, dev_s, clock_ghz, num_cus):
    """Vector-memory instruction budget of the dominant kernel: wave-level vector loads and L1 line accesses per launch from the
    committed PMC passes over THIS run's device cycles per launch, next to what one such instruction costs the CU's vector-memory
    path in tools/ubench/chain.hip (cycles per wave-hop / 4 loads / 32 waves per CU, by distinct lines per instruction)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")), reverse=True):
        try:
            d = json.load(open(path))
            dv = d.get("derived", {})
            if _summary_matches(d, kernel, workload_scene, width, height, False) and "vmem_rd_insts_per_launch" in dv:
                cyc = dev_s * clock_ghz * 1e9
                out = {"vector_loads_per_launch": int(dv["vmem_rd_insts_per_launch"]), "l1_lines_per_vector_load": round(dv.get("l1_lines_per_vmem_rd_inst", 0.0), 2),
                       "cu_cycles_per_vector_load": round(cyc * num_cus / dv["vmem_rd_insts_per_launch"], 2), "source": os.path.relpath(path, ROOT),
                       "note": "cu_cycles_per_vector_load = this run's device cycles per launch x CUs / vector loads per launch: the budget one wave-level "
                               "load gets on its CU's vector-memory path with frames in flight; ubench_cost: what one dwordx4 load of a dependent chain costs "
                               "that path in tools/ubench/chain.hip at 8 waves/SIMD, by distinct records per instruction"}
                for cpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_ubench_chain.json")), reverse=True):
                    runs = [r for r in json.load(open(cpath))["runs"] if r["mix"].startswith("kernel mix") and r["waves_per_simd"] == 8]
                    cost = {}
                    for r in runs:
                        v = r.get("variant", "")
                        key = {"4 x dwordx4 (the kernel's)": f"{r['active_lanes']} lanes, a record each", "4 x dwordx4, 4 lanes/chain": "64 lanes, 16 records",
                               "4 x dwordx4, 16 lanes/chain": "64 lanes, 4 records"}.get(v)
                        if key and (v != "4 x dwordx4, 16 lanes/chain" or r["active_lanes"] == 64):
                            # 8 waves/SIMD asked for, ~24.5 resident on average (the launch's extent / a wave's duration): use the rate, not the duration
                            cost[key] = round(r["active_lanes"] / r["records_per_cycle_per_cu"] / 4.0, 1)
                    out["ubench_cost_cycles_per_load"] = cost
                    break
                for ppath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_ubench_chain_pmc.json")), reverse=True):
                    fit = json.load(open(ppath))["fit"]
                    model = fit["cycles_per_load"] + fit["cycles_per_line"] * dv.get("l1_lines_per_vmem_rd_inst", 0.0)
                    out["busy_modelled"] = round(model / out["cu_cycles_per_vector_load"], 3)
                    out["busy_model"] = (f"({fit['cycles_per_load']:.1f} + {fit['cycles_per_line']:.2f} x L1 lines per load) cycles per vector load -- the line through the chain microbenchmark's "
                                         f"PMC passes, where TA/TD are 86-98 % busy ({os.path.relpath(ppath, ROOT)}) -- over cu_cycles_per_vector_load")
                    break
                return out
        except Exception:
            continue
    return None


def _summary_matches(d, kernel, workload_scene, width, height, shadows):
    b = d.get("bench_line") or {}
    cfg = b.get("config", {})
    return (d.get("kernel", "").startswith(kernel.split("<")[0]) and cfg.get("scene") == workload_scene and not cfg.get("diag_mix3")
            and cfg.get("width") == width and cfg.get("height") == height and (int(cfg.get("shadow", 0) or 0) > 0) == bool(shadows))


def pmc_summary(kernel, workload_scene, width, height, shadows=False):
    """The newest committed rocprofv3 summary (profiles/r*_summary.json, tools/profile_summary.py) of `kernel` on this workload
    (scene, frame size, with / without the shadow-ray extension)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")), reverse=True):
        try:
            d = json.load(open(path))
            if _summary_matches(d, kernel, workload_scene, width, height, shadows) and d.get("counters"):
                return d, os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


L2_PEAK_GBS = 34500.0   # MI355X_MICROARCH.md: aggregate L2 read bandwidth


def secondary_ceilings(kernel, workload_scene, width, height, dev_s):
    """SURVEY.md 8d's secondary ceilings from the committed PMC passes: L2 request bytes over THIS run's device time per launch
    against the guide's aggregate L2 bandwidth, the L1 / L2 hit rates, LDS instructions per launch (the traversal stack)."""
    d, src = pmc_summary(kernel, workload_scene, width, height)
    if d is None:
        return None
    c, dv = d["counters"], d.get("derived", {})
    out = {"source": src}
    if "TCP_TCC_READ_REQ_sum" in c:
        req = c["TCP_TCC_READ_REQ_sum"]["mean_per_launch"]
        gbs = req * 64.0 / dev_s / 1e9
        out["l2"] = {"achieved": round(gbs, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / L2_PEAK_GBS, 4),
                     "achieved_definition": "TCP_TCC_READ_REQ (L1 -> L2 read requests per launch) x 64 B / device time per launch of this run"}
    if "l1_hit_rate" in dv:
        out["l1_hit"] = round(dv["l1_hit_rate"], 4)
    if "l2_hit_rate" in dv:
        out["l2_hit"] = round(dv["l2_hit_rate"], 4)
    if "SQ_INSTS_LDS" in c:
        out["lds_instructions_per_launch"] = int(c["SQ_INSTS_LDS"]["mean_per_launch"])
        out["lds_note"] = "wave-level LDS instructions (traversal-stack pushes / pops, parked values); SQ_LDS_BANK_CONFLICT = %s" % (
            int(c["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"]) if "SQ_LDS_BANK_CONFLICT" in c else "n/a")
    return out


def live_pmc_traffic(scene, width, height, extra_args=(), budget_s=60.0, kern="crt_trace_kernel<false, false, false, false, false>"):
    """HBM-side bytes per launch measured NOW: two child runs of bench.py under `rocprofv3 --kernel-trace --pmc <one counter>` (FETCH_SIZE,
    then WRITE_SIZE: separate passes, nothing else traced, the program itself after `--`), mean per launch of the timed region's kernel.
    Called by bench.py AFTER every number of the line is final and its own session is closed (the children get the GPU to themselves);
    both passes together get `budget_s` seconds, after which -- or on any other failure: no rocprofv3, a pass that fails, no matching
    rows -- the committed profile's figure stays. `extra_args`: the parent's launch configuration (--frames-in-flight, --band-rows,
    --prewarm-ms) so that the children launch what the parent timed. Returns a dict: ok, bytes, note, child_rc (one per pass run)."""
    import csv, glob, shutil, subprocess, tempfile, time
    res = {"ok": False, "bytes": None, "note": None, "child_rc": [], "seconds": 0.0}
    rp = shutil.which("rocprofv3")
    if rp is None:
        res["note"] = "rocprofv3 not found on PATH"
        return res
    # this run is itself being profiled (a preloaded rocprofiler tool): a profiler inside a profiler is asking for trouble
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        res["note"] = "this run is already under a rocprofiler tool"
        return res
    t_start = time.time()
    total, launches = 0.0, []
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        left = budget_s - (time.time() - t_start)
        if left < 5.0:
            res["note"] = f"budget of {budget_s:.0f} s spent before the {counter} pass"
            res["seconds"] = round(time.time() - t_start, 1)
            return res
        d = tempfile.mkdtemp(prefix="crt_pmc_", dir="/tmp")
        cmd = [rp, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"),
               "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--scene", scene, "--width", str(width), "--height", str(height)] + [str(a) for a in extra_args]
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("CRT_KERNEL", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
            p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=left)
            res["child_rc"].append(p.returncode)
            vals = []
            for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
                for r in csv.DictReader(open(f)):
                    if kern in r["Kernel_Name"] and r["Counter_Name"] == counter:
                        vals.append(float(r["Counter_Value"]))
            if p.returncode != 0 or not vals:
                res["note"] = f"{counter} pass: rc {p.returncode}, {len(vals)} launches"
                res["seconds"] = round(time.time() - t_start, 1)
                return res
            total += sum(vals) / len(vals)
            launches.append(len(vals))
        except subprocess.TimeoutExpired:
            res["child_rc"].append("timeout")
            res["note"] = f"{counter} pass: over the {budget_s:.0f} s budget"
            res["seconds"] = round(time.time() - t_start, 1)
            return res
        except Exception as e:  # noqa: BLE001 - a measurement aid must never take the bench line down
            res["note"] = f"{counter} pass: {type(e).__name__}: {e}"
            res["seconds"] = round(time.time() - t_start, 1)
            return res
        finally:
            shutil.rmtree(d, ignore_errors=True)
    res.update(ok=True, bytes=int(total * 1024), seconds=round(time.time() - t_start, 1),
               note=(f"live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 5` spawned by this run after its own session was closed "
                     f"(mean over {launches[0]} / {launches[1]} launches of {kern}; KiB counters x 1024)"))
    return res


def pmc_traffic(kernel, workload_scene, width, height, shadows=False):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes (profiles/*_summary.json:
    rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of this same command, newest round first). FETCH_SIZE is exact
    for this kernel's 64-B record gathers (profiles/r01_fetch_calibration.md) and counts every byte leaving L2, so it
    is an upper bound on HBM reads. Returns (bytes, source) or (None, None) when no matching profile is committed."""
    d, src = pmc_summary(kernel, workload_scene, width, height, shadows)
    if d is None:
        return None, None
    c = d["counters"]
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None, None
    return int((c["FETCH_SIZE"]["mean_per_launch"] + c["WRITE_SIZE"]["mean_per_launch"]) * 1024), src


def hbm_object(workload_scene, width, height, shadows, ms_per_step):
    """The HBM-roofline sub-record of an extra view of the bench line (config 2 / 3 / 5, the shadow-ray lines, the dense and the
    reference-asset views): FETCH_SIZE + WRITE_SIZE per launch from the newest committed profile of exactly that workload over THIS
    run's device time per frame, with the L1 / L2 hit rates of the same passes. None when no such profile is committed."""
    d, src = pmc_summary("crt_trace_kernel", workload_scene, width, height, shadows)
    if d is None or "FETCH_SIZE" not in d["counters"] or "WRITE_SIZE" not in d["counters"] or not ms_per_step:
        return None
    c, dv = d["counters"], d.get("derived", {})
    tr = int((c["FETCH_SIZE"]["mean_per_launch"] + c["WRITE_SIZE"]["mean_per_launch"]) * 1024)
    gbs = tr / (ms_per_step * 1e-3) / 1e9
    out = {"traffic": tr, "traffic_source": src, "achieved_gbs": round(gbs, 1), "peak_gbs": HBM_PEAK_GBS, "frac": round(gbs / HBM_PEAK_GBS, 4),
           "kernel": d.get("kernel")}
    if "l1_hit_rate" in dv:
        out["l1_hit"] = round(dv["l1_hit_rate"], 4)
    if "l2_hit_rate" in dv:
        out["l2_hit"] = round(dv["l2_hit_rate"], 4)
    return out
