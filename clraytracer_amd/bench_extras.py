"""bench.py's sub-records OUTSIDE the contract's timed region (rank 0, N = 1): the other BASELINE configurations and views with their `hbm` objects, the
N = 1 point of the scaling curve, upstream's own assets, the device BuildBVH, the opt-in kernel forms against the default kernel measured the same way,
401 instances static and moving, animated instances. (The CPU baseline stays in bench.py: it is the one leg that may load the oracle, and nothing
under clraytracer_amd/ ever does.) Nothing here feeds
`value` / `ms_per_step`; every block is an extra that may fail without taking the line with it. Moved out of bench.py in round 6 (the
contract path stays readable on its own); bench.py hands over a namespace with what the blocks need and takes back the session that is open at the end."""
import ctypes as C
import os
import time

import numpy as np

from . import _lib, driver, scenes
from .measure import HBM_PEAK_GBS, hbm_object

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(ctx):
    """ctx: args, cnt, crt_render, device_index, flags, fp, hip, measure_view, n, out, rays_per_frame, sc, s, width, height. Returns the open session."""
    args, cnt, crt_render, device_index, flags, fp, hip = ctx.args, ctx.cnt, ctx.crt_render, ctx.device_index, ctx.flags, ctx.fp, ctx.hip
    measure_view, n, out, rays_per_frame, sc, s, width, height = ctx.measure_view, ctx.n, ctx.out, ctx.rays_per_frame, ctx.sc, ctx.s, ctx.width, ctx.height
    extras = n == 1 and not args.no_extras and not args.width and not args.height and args.scene == "multi-1M" and not args.shadows
    kx = max(20, min(100, args.steps))
    if extras:
        # BASELINE configs 3-4 as written ("primary + 1 shadow ray"): the shadow-ray extension on the bench scene (upstream has no
        # shadow ray, kernel_main.cl:256-258; semantics defined by the oracle, bit-exact in tests/test_gpu_shadows.py)
        out["with_shadow_rays"] = measure_view(s, flags | 32, kx, f"{sc.name} {width}x{height}, primary + reflection bounce + 1 any-hit shadow ray per lit first hit")
        out["with_shadow_rays"]["hbm"] = hbm_object(sc.name, width, height, True, out["with_shadow_rays"]["steady_state"]["ms_per_step"])
        # ... on config 3's own scene too (sponza-class-250k is loaded below with the other scenes)
        # the dense view of the same scene: 97 % of the primary rays hit, 46 inner visits per ray (the headline view is 69 % sky)
        dense = scenes.get("multi-1M-dense")
        s.set_camera(dense.camera_pos, dense.camera_front)
        out["dense_view"] = measure_view(s, flags, kx, f"multi-1M-dense: the same scene seen from among its instances, {width}x{height}")
        out["dense_view"]["synchronous_frames"] = measure_view(s, flags & ~4, kx, "same view, one frame at a time")["value"]
        out["dense_view"]["hbm"] = hbm_object("multi-1M-dense", width, height, False, out["dense_view"]["steady_state"]["ms_per_step"])
        s.set_camera(sc.camera_pos, sc.camera_front)
    if n == 1 and not args.width and not args.height and not args.no_config5:
        # the N = 1 point of BASELINE config 5 (the 3840x2160 frame the N > 1 lines tile over the ranks), so that a scaling
        # curve has a base on the same workload; outside the contract's timed region
        s.resize(3840, 2160)
        a5, iv5, ip5 = s.trace_args()
        q_args, q_iv, q_ip = C.byref(a5), iv5.ctypes.data_as(fp), ip5.ctypes.data_as(fp)
        s.render_raw(8 | (32 if args.shadows else 0))
        rays5 = s.counters()["rays"]
        for _ in range(5):
            _lib.check(crt_render(q_args, q_iv, q_ip, flags), "crt_render")
        _lib.check(hip.crt_sync(), "crt_sync")
        k5 = max(10, min(50, args.steps))
        t0 = time.perf_counter()
        for _ in range(k5):
            rc = crt_render(q_args, q_iv, q_ip, flags)
        _lib.check(hip.crt_sync(), "crt_sync")
        dt5 = (time.perf_counter() - t0) / k5
        _lib.check(rc, "crt_render")
        out["scale_base_n1"] = {"value": round(rays5 / dt5 / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(dt5 * 1e3, 4),
                                "rays_per_frame": rays5, "frames": k5, "workload": f"{sc.name} 3840x2160 (BASELINE config 5) on one GPU",
                                "hbm": hbm_object(sc.name, 3840, 2160, bool(args.shadows), dt5 * 1e3),
                                "note": "the N = 1 point of the scaling curve: lines with n_gpus > 1 render THIS frame tiled over the ranks (they carry the same "
                                        "measurement as single_gpu_same_workload); formerly `config5_one_gpu`"}
        s.resize(width, height)
    if extras:
        # upstream's own assets (Sponza + Sibenik .clm caches with their 27 JPEG texture imports) and BASELINE config 3
        # (sponza-class-250k, with its shadow ray): own sessions, one after the other (the library drives one at a time)
        s.close()
        for name, views in (("sponza-sibenik", (("reference_assets", flags),)),
                            ("sponza-class-250k", (("config3_with_shadow_rays", flags | 32), ("config3", flags))),
                            ("cornell-1k", (("config2", flags),))):
            sc2 = scenes.get(name)
            t0 = time.time()
            with driver.Session(width, height, device=device_index) as s2:
                s2.load_scene(sc2)
                load2 = time.time() - t0
                for key, vflags in views:
                    out[key] = measure_view(s2, vflags, kx, f"{name}: {sc2.num_tris} triangles, {len(sc2.instances)} instances, {width}x{height}"
                                            + (", + 1 shadow ray per lit first hit" if vflags & 32 else ""))
                    out[key]["synchronous_frames"] = measure_view(s2, vflags & ~4, kx, "same view, one frame at a time")["value"]
                    out[key]["scene_load_s"] = round(load2, 2)
                    out[key]["hbm"] = hbm_object(name, width, height, bool(vflags & 32), out[key]["steady_state"]["ms_per_step"])
        # the device BuildBVH (crt_build_bvh, SURVEY 8f rank 1: BVH.cpp:218-255 on the GPU, same bytes as the host builder) on the bench scene's
        # triangles: best of 3 builds incl. the re-layout for rendering; bytes = what the level passes must move at least
        try:
            with driver.Session(64, 48, device=device_index) as sb:
                sb.load_scene(sc)
                ab = sb.arenas()
                hb = _lib.host()
                counts = []
                for m in range(hb.crth_num_meshes()):
                    info = np.zeros(4, np.uint32); hb.crth_mesh_info(m, info.ctypes.data); counts.append(int(info[0]))
                cb = np.asarray(counts, np.uint32); trisb = np.ascontiguousarray(ab["tris"].copy())
                best_b, used_b = None, C.c_uint32(0)
                for _ in range(3):
                    _lib.check(hip.crt_upload_triangles(trisb.ctypes.data, 0, trisb.nbytes), "crt_upload_triangles")
                    t0 = time.perf_counter()
                    _lib.check(hip.crt_build_bvh(0, cb.ctypes.data, len(cb), 0, 0, C.byref(used_b)), "crt_build_bvh")
                    d = time.perf_counter() - t0
                    best_b = d if best_b is None else min(best_b, d)
                nb = np.zeros(used_b.value, _lib.NODE_DTYPE); rb = np.zeros(len(cb), np.uint32)
                _lib.check(hip.crt_download_bvh_nodes(nb.ctypes.data, 0, nb.nbytes), "crt_download_bvh_nodes")
                _lib.check(hip.crt_download_bvh_roots(rb.ctypes.data, 0, len(rb)), "crt_download_bvh_roots")
                # every level reads the triangles of its open nodes three times (bounds, bins + sweep, partition) and writes them once:
                # 4 x 80 B x (sum over nodes of their triangle count) = 320 B x sum over leaves of count x (depth + 1)
                frontier, depth, tri_levels = rb.astype(np.int64), 1, 0
                while len(frontier):
                    leaf = nb["triCount"][frontier] > 0
                    tri_levels += int(nb["triCount"][frontier][leaf].sum()) * depth
                    inner = frontier[~leaf]
                    left = nb["leftFirst"][inner].astype(np.int64)
                    frontier = np.concatenate([left, left + 1]); depth += 1
                moved = 320 * tri_levels
                lv_b, ln_b = C.c_uint32(0), C.c_uint32(0)
                hip.crt_debug_build_stats(C.byref(lv_b), C.byref(ln_b))
                out["bvh_build"] = {"ms": round(best_b * 1e3, 3), "triangles": int(len(trisb)), "nodes": int(used_b.value), "levels": depth - 1,
                                    "launches": int(ln_b.value), "level_handshakes": int(lv_b.value),
                                    "one_submission_floor": "3.86 ms against 4.10 (profiles/r06_bvh_replay.txt: every launch enqueued back to back with recorded grid sizes, no publish kernel, "
                                                            "no host wait per level): any device-driven level loop can gain at most 6 %",
                                    "bytes_moved": int(moved), "frac_of_hbm": round(moved / best_b / 1e9 / HBM_PEAK_GBS, 4),
                                    "triangles_per_s": round(len(trisb) / best_b, 0),
                                    "note": "crt_build_bvh incl. the re-layout for rendering, best of 3 (host wall clock around the call); bytes_moved = 320 B x the sum over all "
                                            "nodes of their triangle count (three reads and one write of an 80-B Tri per open node and level): a lower bound; the build is "
                                            "launch- and atomics-bound (`launches`, one 16-B control record published per level), not bandwidth-bound"}
        except Exception as e:  # pragma: no cover - an extra, never the reason for a missing line
            out["bvh_build"] = {"error": str(e)}
        # BASELINE config 4 as written ("LDS stack + wavefront compaction on"): the wavefront form of Trace -- bounce 0, ballot compaction of
        # the continuing paths into a queue, bounce 1 as dense 64-ray packets (CRT_KERNEL=wavefront, read by crt_init) -- on the bench frame, and the
        # two forms that compact INSIDE the wave (round 5; crt_refill.h): `refill` (a lane whose path has ended takes the next pixel of its 16x8
        # block) and `block` (a block's primary rays with a candidate instance, then its bounce rays, regrouped into dense packets between the
        # stages). All three are bit-identical to the default kernel and to the oracle at this size (tests/test_gpu_variants.py) and slower,
        # which is why they are opt-in; each is put against the default kernel in the same mode (frames in flight / one frame at a time).
        saved_kernel = os.environ.get("CRT_KERNEL")
        try:
            # like against like: the default kernel is measured by the same measure_view() calls in the same loop as the three forms,
            # not taken from the contract's timed region (whose 20 steps include the pipeline's fill and read 8 % below a 200-frame run).
            # The forms render no shadow rays (CRT_E_UNSUPPORTED): a --shadows run compares the plain frame.
            vflags = flags & ~(32 | 1024)
            ref_v = None
            for key, variant, label in ((None, "default", "crt_trace_kernel (the default megakernel)"),
                                        ("wavefront_compaction", "wavefront", "crt_primary_kernel -> compaction -> crt_bounce_kernel"),
                                        ("in_wave_refill", "refill", "crt_trace_refill_kernel: in-tile lane refill, 16x8 blocks"),
                                        ("in_wave_block_compaction", "block", "crt_trace_block_kernel: classify -> dense primary packets -> dense bounce packets, 16x8 blocks"),
                                        # north_star's "hot BVH tiles staged in LDS" (round 6; crt_ldstop.h): four-wave workgroups sharing a 15.75 KiB LDS copy of
                                        # every mesh's top tree levels, 15 stack slots per wave in LDS, 5 waves per SIMD instead of 8
                                        ("lds_tree_tops", "ldstop", "crt_trace_ldstop_kernel: 4 tiles per workgroup, the trees' top levels (252 pair records) staged in LDS")):
                os.environ["CRT_KERNEL"] = variant
                with driver.Session(width, height, device=device_index) as sw:
                    sw.load_scene(sc)
                    r = measure_view(sw, vflags, kx, f"{sc.name} {width}x{height}, primary + reflection bounce, {label}")
                    r["kernel"] = sw.last_kernel()                      # crt_debug_last_kernel: which kernel really rendered these frames
                    r["synchronous_frames"] = measure_view(sw, vflags & ~4, kx, "same frame, one at a time")["value"]
                    if key is None:
                        ref_v = r
                        continue
                    r["default_kernel_same_measurement"] = {"value": ref_v["value"], "synchronous_frames": ref_v["synchronous_frames"], "kernel": ref_v["kernel"]}
                    r["vs_default_kernel"] = round(r["value"] / ref_v["value"], 3)
                    r["vs_default_kernel_in_flight"] = r["vs_default_kernel"]
                    r["vs_default_synchronous"] = round(r["synchronous_frames"] / ref_v["synchronous_frames"], 3)
                    out[key] = r
        except Exception as e:  # pragma: no cover - an extra, never the reason for a missing line
            out["wavefront_compaction_error"] = str(e)
        finally:
            if saved_kernel is None:
                os.environ.pop("CRT_KERNEL", None)
            else:
                os.environ["CRT_KERNEL"] = saved_kernel
        # SURVEY 8f rank 1's other half, in the driver's record: (a) upstream's instance limit (Renderer.hpp:16: 401, every ray loops over all
        # of them, kernel_main.cl:198) with the linear sphere loop and with the instance tree; (b) every instance moved before every frame
        # (upstream's Engine_Tick -> SetMeshPosition -> dirty range -> clEnqueueWriteBuffer, Renderer.cpp:268-298,312-320 = crt_upload_instances)
        try:
            tiny = scenes.get("tiny")
            mi = {}
            saved_tlas = os.environ.get("CRT_TLAS")
            try:
                for tl in ("0", "1"):
                    os.environ["CRT_TLAS"] = tl          # read by crt_init: 0 = linear sphere loop, 1 = instance tree
                    with driver.Session(width, height, device=device_index) as sm:
                        sm.load_scene(tiny)
                        sm.h.crth_begin_instances()
                        for k in range(len(tiny.instances), 401):
                            m = scenes._trs(0.6 + 0.1 * (k % 5), (0.3, 1.0, 0.2), 0.37 * k, (float((k % 21) - 10) * 6.0, float((k // 21) - 9) * 6.0, -float(k % 7) * 2.0))
                            pm, keep = _lib.fptr(m)
                            sm.h.crth_register_instance(k % 2, 0xFFFF, pm)
                        sm.h.crth_end_instances()
                        sm.set_camera((0.0, 0.0, 23.0 * 6.0), scenes._normalize((0.0, 0.0, -1.0)))
                        mi[tl] = measure_view(sm, flags, 30, f"tiny's two meshes instanced 401 times on a grid, {width}x{height}, " + ("instance tree" if tl == "1" else "linear sphere loop"))
                        if tl == "1":
                            # ... and all 401 MOVING: upstream's Engine_Tick -> SetMeshPosition -> dirty range -> clEnqueueWriteBuffer (Renderer.cpp:268-298,312-320)
                            # = crt_upload_instances before every frame. An upload is host-only (memcpy + rebuild_instance_master: the uploaded records'
                            # bounding spheres and cull ranges, a refit of the instance tree); the frame's slot refreshes its tables in one launch on its own stream.
                            inst_m = sm.arenas()["instances"].copy()
                            a_m, iv_m, ip_m = sm.trace_args()
                            q_m = (C.byref(a_m), iv_m.ctypes.data_as(fp), ip_m.ctypes.data_as(fp))
                            def many_animated(move, frames):
                                for _ in range(6):
                                    crt_render(*q_m, flags)
                                _lib.check(hip.crt_sync(), "crt_sync")
                                t_up, t0 = 0.0, time.perf_counter()
                                for _ in range(frames):
                                    if move:
                                        inst_m["inv"][:, 3, 1] += np.float32(1e-4)
                                        tu = time.perf_counter()
                                        hip.crt_upload_instances(inst_m.ctypes.data, 0, len(inst_m))
                                        t_up += time.perf_counter() - tu
                                    r_ = crt_render(*q_m, flags)
                                _lib.check(hip.crt_sync(), "crt_sync"); _lib.check(r_, "crt_render")
                                return (time.perf_counter() - t0) / frames, t_up / frames
                            (dts, _), (dtm, tup) = many_animated(False, 30), many_animated(True, 30)
                            mi["animated"] = {"instances": len(inst_m), "value": round(mi["1"]["rays_per_frame"] / dtm / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(dtm * 1e3, 4),
                                              "vs_static": round(dts / dtm, 3), "static_ms_per_step": round(dts * 1e3, 4), "host_tlas_rebuild_us": round(tup * 1e6, 1), "frames": 30,
                                              "workload": f"the same 401 instances, every one re-uploaded (crt_upload_instances) before every frame, frames in flight; host_tlas_rebuild_us = "
                                                          "mean host time of that call (memcpy + bounding spheres + cull ranges + instance-tree refit, no device work); rays counted on the static scene"}
            finally:
                if saved_tlas is None:
                    os.environ.pop("CRT_TLAS", None)
                else:
                    os.environ["CRT_TLAS"] = saved_tlas
            out["many_instances"] = {"instances": 401, "value": mi["1"]["value"], "unit": "Mrays/s", "ms_per_step": mi["1"]["ms_per_step"],
                                     "rays_per_frame": mi["1"]["rays_per_frame"], "linear_loop": {"value": mi["0"]["value"], "ms_per_step": mi["0"]["ms_per_step"]},
                                     "tlas_vs_linear": round(mi["1"]["value"] / mi["0"]["value"], 3), "workload": mi["1"]["workload"]}
            if "animated" in mi:
                out["animated_many_instances"] = mi["animated"]
        except Exception as e:  # pragma: no cover - an extra, never the reason for a missing line
            out["many_instances"] = {"error": str(e)}
        s = driver.Session(width, height, device=device_index)
        s.load_scene(sc)
        try:
            inst = s.arenas()["instances"].copy()
            targs, iv, ip = s.trace_args()
            p_args, p_iv, p_ip = C.byref(targs), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp)
            def animated(move, frames):
                for _ in range(6):
                    crt_render(p_args, p_iv, p_ip, flags)
                _lib.check(hip.crt_sync(), "crt_sync")
                t0 = time.perf_counter()
                for _ in range(frames):
                    if move:
                        inst["inv"][:, 3, 1] += np.float32(1e-4)                 # every instance drifts a little
                        hip.crt_upload_instances(inst.ctypes.data, 0, len(inst))
                    rc = crt_render(p_args, p_iv, p_ip, flags)
                _lib.check(hip.crt_sync(), "crt_sync")
                _lib.check(rc, "crt_render")
                return (time.perf_counter() - t0) / frames
            dt_static, dt_moved = animated(False, kx), animated(True, kx)
            out["animated_instances"] = {"value": round(rays_per_frame / dt_moved / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(dt_moved * 1e3, 4),
                                         "vs_static": round(dt_static / dt_moved, 3), "static_ms_per_step": round(dt_static * 1e3, 4), "frames": kx,
                                         "workload": f"{sc.name} {width}x{height}, all {len(inst)} instances re-uploaded (crt_upload_instances) before every frame, frames in flight; "
                                                     "rays counted on the static scene"}
            _lib.check(hip.crt_upload_instances(s.arenas()["instances"].ctypes.data, 0, len(inst)), "crt_upload_instances")   # back to the scene's own table
        except Exception as e:  # pragma: no cover
            out["animated_instances"] = {"error": str(e)}
    return s
