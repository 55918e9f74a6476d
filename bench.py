#!/usr/bin/env python3
"""Benchmark of the ray-transport hot path on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME]

N = 1 (default)  BASELINE config 2, the headline metric: Msamples/s on the Cornell box (scenes/cbox.xml geometry),
                 512 x 512, 256 spp, path max_depth 6, tent filter.  One step = one full render, scene and all path
                 state resident in HBM.
N > 1            BASELINE config 5: the same scene at 4096 x 4096, 1024 spp (17.18 G samples per step), STRONG scaling:
                 the film is cut into interleaved 64-row bands dealt round-robin to the ranks (parallel.py); every rank
                 renders its bands with the global RNG keys, ONE RCCL gather of the finished bands to rank 0 inside the
                 timed region; value = total samples / max-over-ranks time.  `--gpus N` without a torchrun environment
                 launches the ranks itself (python -m torch.distributed.run ..., as fresh child processes; the parent
                 never touches the GPU) and relays rank 0's JSON line.
--config NAME    cbox (config 2) | cbox4k (config 5) | us_testring (the ring of config 4 as ultrasound phantom) | usmain_loop (one finite-difference
                 iteration of the reference's optimisation loop, USMain.py:262-289: 2 x us_render, device-resident) | us_sphere_box_emitter
                 (config 3 with every path's primary ray drawn from CustomEmitter.sample_ray) | us_sphere_box (config 3: MitsubaScenes/Sphere_Box.xml phantom,
                 5 x 64 rays x 838 912 paths = 268 M transducer paths, ultrasound mode; N > 1: path ranges + one
                 reduce(sum)) | testring (config 4: TestRing/TestRing.obj, 1024 x 1024, 512 spp, LDS-resident BVH;
                 N > 1: bands + gather).  Default: cbox at N = 1, cbox4k at N > 1.

With no --config at N = 1 the line also carries `also`: the other named scenes of BASELINE.json (us_sphere_box = config 3,
testring = config 4, cbox4k = config 5 on one GPU), 3 steps each after the headline steps, each with its own roofline and its L2
against the CPU port on a bounded sample; `seeds`: the headline render at seeds 1 and 2; and `scale_ref`: the
N = 1 value of the workload `--gpus N > 1` runs (cbox4k), so that a scaling curve has its own one-GPU point on the line.  At N > 1
the line carries `speedup_vs_scale_ref` when PBRT_SCALE_REF_MSAMPLES hands that value over.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the dominant kernels (k_bounce / k_us_bounce / k_trace_primary + k_trace + k_shade), bound = "hbm" as the contract
                  defines it: achieved = ALGORITHMIC bytes of the launches (DESIGN.md byte model, evaluated on the live-path
                  counters of this very run) / their HIP-event durations on the library's stream, frac = achieved / 8 TB/s --
                  measured by every run.  `traffic` = HBM bytes per launch from rocprofv3 PMC passes, copied from
                  profiles/pmc_traffic.json only if that file was recorded for the kernel sources this run uses (sha256 of
                  csrc/), else null; `traffic_source` says which, `traffic_over_algorithmic` is their ratio.  These kernels run
                  against VALU ISSUE, not HBM; that side rides along as valu_* keys (valu_frac =
                  valu_issue_busy x lane_active, the share of the chip's 78.6 T f32 lane-slots per second that did work; same
                  file, same hash guard, null when not recorded).
  per_rank_ms  -- N > 1: every rank's render / acquisition time per step and (collective_ms) its time in the gather / reduce.
  cpu_baseline -- the CPU oracle (C++ port of the same algorithm, same RNG: the reference's Python / Mitsuba path
                  cannot run on this box) on the host cores, on a bounded sample of the same workload (~15 s), rank 0,
                  N = 1 only; the same run gives the per-pixel L2 between the HIP result and the CPU result.
"""
from __future__ import annotations

import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "physics-based-ray-tracing_amd"
SCENES = os.path.join(ROOT, "tests", "scenes")
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12  # 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6 T f32 lane-operations per second

CONFIGS = {
    "cbox": dict(kind="radiance", scene="cbox.xml", res=512, spp=256, max_depth=6, band_rows=0, baseline_config=2,
                 metric="Msamples/s on cbox.xml 512x512 x 256 spp (radiance, path max_depth 6)",
                 what="tent filter, 6 analytic quads + 2 spheres"),
    "cbox4k": dict(kind="radiance", scene="cbox.xml", res=4096, spp=1024, max_depth=6, band_rows=64, baseline_config=5,
                   metric="Msamples/s on cbox.xml 4096x4096 x 1024 spp (radiance, path max_depth 6), band-sharded",
                   what="tent filter, 6 analytic quads + 2 spheres"),
    # pass_paths: the bench process owns its GPU, so the whole 512 Mi-path job is ONE pass (183 GB of workspace); the library's own
    # default stops at half of the device (two passes of 256 Mi) so that a second tenant finds room
    "testring": dict(kind="radiance", scene="testring.xml", res=1024, spp=512, max_depth=6, band_rows=0, baseline_config=4, pass_paths=512 << 20,
                     metric="Msamples/s on TestRing/TestRing.obj 1024x1024 x 512 spp (radiance, path max_depth 6, LDS-resident BVH)",
                     what="tent filter, TestRing.obj 1152 triangles + ground + area light"),
    "us_testring": dict(kind="ultrasound", scene="us_testring.xml", ppr=838912, baseline_config=4,
                        metric="Msamples/s on the TestRing.obj phantom (ultrasound twin of config 4), 5 x 64 rays x 838912 paths (UltraBSDF, max_depth 10, LDS-resident BVH)",
                        what="TestRing.obj 1152 triangles + 5 walls, UltraBSDF, 5 angles x 64 elements, channel buffer 5 x 64 x 10000"),
    # SURVEY 8(f-1 / f-2), the caller of the hot path: one iteration of the reference's finite-difference loop (USMain.py:262-289) =
    # 2 x (params.update() + us_render: acquisition -> DAS -> envelope -> log compression) on the USMain.py:26-90 scene
    "usmain_loop": dict(kind="usmain_loop", scene="us_plate.xml", ppr=(1, 64, 4096), baseline_config=None,
                        metric="ms per finite-difference iteration of USMain.py:262-289 (2 x us_render + 2 x params.update) at paths_per_ray 1",
                        what="USMain.py:26-90 plate + wall, 5 angles x 64 elements, channel buffer 5 x 64 x 10000, lambda / 4 scan grid"),
    # BASELINE config 3 read literally ("Sphere_Box.xml with CustomBSDF + CustomEmmitter"): every path draws its primary ray from
    # CustomEmitter.sample_ray (PBRT_US_PRIMARY_EMITTER, DESIGN D15) -- no first-bounce tables, bounce 0 traced by all 268 M paths
    "us_sphere_box_emitter": dict(kind="ultrasound", scene="us_sphere_box.xml", ppr=838912, baseline_config=3,
                                  load=dict(primary_rays="emitter"),
                                  metric="Msamples/s on MitsubaScenes/Sphere_Box.xml phantom with CustomEmitter primary rays, 5 x 64 rays x 838912 paths (ultrasound, UltraBSDF, max_depth 10)",
                                  what="sphere + 5 walls, UltraBSDF, primary rays from CustomEmitter.sample_ray (64 elements, +-15 degrees in 5 strata), channel buffer 5 x 64 x 10000"),
    "us_sphere_box": dict(kind="ultrasound", scene="us_sphere_box.xml", ppr=838912, baseline_config=3,
                          metric="Msamples/s on MitsubaScenes/Sphere_Box.xml phantom, 5 x 64 rays x 838912 paths (ultrasound, UltraBSDF, max_depth 10)",
                          what="sphere + 5 walls, UltraBSDF, 5 angles x 64 elements, channel buffer 5 x 64 x 10000"),
}


def kernel_source_hash():
    """sha256 over the kernel sources and the build flags: ties profiles/pmc_traffic.json to the code it was measured on (the
    GPU box has no .git to ask for HEAD).  Comments and blank lines are left out, so a reworded comment does not orphan a
    measurement.  Of the Makefile only the lines that set the compiler, the target and the flags count."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, PKG, "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")) or name == "Makefile":
            h.update(name.encode())
            for line in open(os.path.join(d, name), "r", errors="replace"):
                code = line.split("//", 1)[0].strip() if name != "Makefile" else line.split("#", 1)[0].strip()
                if name == "Makefile" and not code.startswith(("HIPCC", "ARCH", "HIPFLAGS")):
                    continue
                if code:
                    h.update(code.encode())
                    h.update(b"\n")
    return h.hexdigest()[:16]


def visible_gpu_count():
    """GPUs this process would see, counted WITHOUT loading a GPU runtime: the KFD topology in sysfs (a node with
    simd_count > 0 is a GPU; CPUs are nodes too), narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when they are
    set.  The launcher parent must stay a process that has never touched the GPU (its children are forked from it)."""
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(base):
            try:
                with open(os.path.join(base, node, "properties")) as f:
                    for line in f:
                        k, _, v = line.partition(" ")
                        if k == "simd_count" and int(v) > 0:
                            n += 1
                            break
            except (OSError, ValueError):
                continue
    except OSError:
        n = 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            listed = len([t for t in v.split(",") if t.strip() != ""])
            n = min(n, listed) if n else 0
    return n


def launch_ranks(args, argv):
    """--gpus N > 1 outside torchrun: start the ranks as fresh children of a parent that has not initialised the GPU
    (no torch import, no HIP call: the GPUs are counted in sysfs)."""
    # under rocprofv3 the profiler's preloaded library has initialised the GPU in THIS process already and the children would
    # inherit the preload: refuse (profile multi-rank runs rank by rank, `rocprofv3 ... -- python3 bench.py` under torchrun's env)
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROFILER_", "ROCPROF_")) for k in os.environ):
        print("bench.py: --gpus N > 1 would launch ranks from a profiled process; run one rank per rocprofv3 instead", file=sys.stderr)
        return 2
    if not args.rehearse_on_one_gpu:
        have = visible_gpu_count()
        if args.gpus > have:
            print(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s); a multi-rank rehearsal on one GPU is "
                  f"--rehearse-on-one-gpu (gloo, not a benchmark)", file=sys.stderr)
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    assert "torch" not in sys.modules, "the launcher parent must not load torch (or any GPU runtime)"
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="0: 10 for cbox / us_sphere_box, 3 for the larger workloads")
    ap.add_argument("--warmup", type=int, default=-1, help="-1: 2 for cbox / us_sphere_box, 1 for the larger workloads")
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="N = 1 default run: only the headline, not the other named scenes / seeds")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU work of the cpu_baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0: min(visible cores, 64)")
    ap.add_argument("--res", type=int, default=0, help="override the film size (rehearsals only: not the named workload)")
    ap.add_argument("--spp", type=int, default=0, help="override spp / paths per ray (rehearsals only)")
    ap.add_argument("--force-collective", action="store_true",
                    help="under torchrun with ONE rank: initialise the nccl (RCCL) process group anyway and issue the gather / reduce at "
                         "world size 1 -- runs torch's bundled RCCL beside libpbrt_hip.so in one process on a 1-GPU box")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a 1-GPU box: every rank uses device 0 and the collective runs over gloo "
                         "(host staging); exercises sharding + stitching, NOT RCCL -- numbers are not benchmark numbers")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    os.environ["PBRT_DEVICE"] = str(local_rank)
    name = args.config or ("cbox" if world == 1 else "cbox4k")
    cfg = dict(CONFIGS[name])
    overridden = bool(args.res or args.spp)
    if args.res:
        cfg["res"] = args.res
    if args.spp:
        cfg["spp" if cfg["kind"] == "radiance" else "ppr"] = args.spp
    big = name in ("cbox4k", "testring")
    steps = args.steps or (3 if big else 10)
    warmup = args.warmup if args.warmup >= 0 else (1 if big else 2)

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the ray-transport hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force = bool(args.force_collective and "WORLD_SIZE" in os.environ)
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    mi = importlib.import_module(PKG)
    par = importlib.import_module(PKG + ".parallel")
    env = dict(mi=mi, par=par, np=np, torch=torch, dist=dist, device=device, rank=rank, world=world, args=args, force=force)
    if cfg["kind"] == "usmain_loop":
        if world > 1:
            sys.exit("bench.py --config usmain_loop: the reference's optimisation loop is one serial chain of renders (replicas only)")
        out = measure_usmain_loop(env, cfg, args.steps or 20, args.warmup if args.warmup >= 0 else 3, with_cpu=not args.no_cpu_baseline)
        print(json.dumps(out), flush=True)
        if force:
            dist.barrier()
            dist.destroy_process_group()
        return out
    out = measure(env, name, cfg, steps, warmup, seed=0, overridden=overridden, with_cpu=(world == 1 and not args.no_cpu_baseline))
    if rank == 0 and world == 1 and args.config is None and not overridden and not args.no_also:
        # the driver runs this command once: the other named scenes of BASELINE.json ride on the same line (three timed steps after
        # TWO warm-up steps: after the CPU legs of the entry before, the first acquisition or two run ~10 % slow -- clocks,
        # tools/idle_gap_probe.py)
        out["also"] = []
        for other in ("us_sphere_box", "us_sphere_box_emitter", "testring", "us_testring", "cbox4k"):
            o = measure(env, other, dict(CONFIGS[other]), 3, 2, seed=0, overridden=False, with_cpu=not args.no_cpu_baseline,
                        cpu_seconds=min(args.cpu_seconds, 6.0))
            keep = {k: o[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "roofline", "l2_vs_cpu_ref", "cpu_baseline", "config")
                    if k in o}
            out["also"].append(keep)
            if other == "cbox4k":
                # the N = 1 point of the workload `--gpus N > 1` runs (BASELINE config 5): a 1 -> 8 curve must be read against
                # THIS value, not against the headline (config 2, another workload)
                out["scale_ref"] = {"config": "cbox4k", "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"],
                                    "samples_per_step": o["config"]["samples_per_step"], "n_gpus": 1}
        # the caller of the hot path: one finite-difference iteration of the reference's loop, device-resident
        o = measure_usmain_loop(env, dict(CONFIGS["usmain_loop"]), 10, 2, with_cpu=False)
        out["also"].append({k: o[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "higher_is_better", "roofline", "sizes", "config")})
        out["seeds"] = {"0": {"value": out["value"], "ms_per_step": out["ms_per_step"]}}
        for sd in (1, 2):  # seeds 0, 1, 2
            o = measure(env, name, dict(cfg), 5, 1, seed=sd, overridden=False, with_cpu=False)
            out["seeds"][str(sd)] = {"value": o["value"], "ms_per_step": o["ms_per_step"]}
        if not args.no_cpu_baseline:
            out["seeds"]["parity"] = seed_parity(env, cfg)
    if rank == 0 and name == "cbox4k":
        if world == 1 and not overridden:
            out["scale_ref"] = {"config": "cbox4k", "value": out["value"], "unit": out["unit"], "ms_per_step": out["ms_per_step"],
                                "samples_per_step": out["config"]["samples_per_step"], "n_gpus": 1}
        elif world > 1:
            # the one-GPU value of the same workload, handed over by whoever ran it (the N = 1 line's scale_ref.value)
            ref = os.environ.get("PBRT_SCALE_REF_MSAMPLES")
            try:
                ref = float(ref) if ref else None
            except ValueError:
                ref = None
            out["scale_ref"] = {"config": "cbox4k", "value": ref, "unit": "Msamples/s", "n_gpus": 1,
                                "source": "PBRT_SCALE_REF_MSAMPLES" if ref else "not given (set PBRT_SCALE_REF_MSAMPLES to the N = 1 line's scale_ref.value)"}
            out["speedup_vs_scale_ref"] = round(out["value"] / ref, 4) if ref else None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or force:
        dist.barrier()
        dist.destroy_process_group()
    return out


def measure(env, name, cfg, steps, warmup, seed, overridden, with_cpu, cpu_seconds=None):
    """W untimed + K timed steps of one named workload; rank 0 gets the result record (others None)."""
    mi, par, np, torch, dist = env["mi"], env["par"], env["np"], env["torch"], env["dist"]
    device, rank, world, args = env["device"], env["rank"], env["world"], env["args"]
    radiance = cfg["kind"] == "radiance"
    if radiance:
        RES, SPP = cfg["res"], cfg["spp"]
        scene = mi.load_file(os.path.join(SCENES, cfg["scene"]), res=RES, spp=SPP, max_depth=cfg["max_depth"])
        # band layout: the whole film in one call on one GPU unless the config is band-sharded by definition; with
        # several ranks interleaved 64-row bands (8 per rank at 4096 rows and N = 8: max / mean band cost <= 1.05)
        band_rows = cfg["band_rows"] or (RES if world == 1 else 64)
        total_units = RES * RES * SPP
    else:
        PPR = cfg["ppr"]
        scene = mi.load_file(os.path.join(SCENES, cfg["scene"]), **cfg.get("load", {}))
        ui = scene.integrator()
        band_rows = 0
        total_units = ui.n_angles * ui.n_elements * PPR
    scene.device()  # upload once, outside the timed region
    ctx = mi.default_context()
    acc = dict(bounce_ms=0.0, bounce_bytes=0.0, trace_bytes=0.0, kernel_ms=0.0, launches=0, segments=0, samples=0)

    def account():
        st = ctx.stats()
        acc["bounce_ms"] += st["bounce_ms"]
        acc["bounce_bytes"] += st["bounce_model_bytes"]
        acc["trace_bytes"] += st.get("trace_model_bytes", 0)
        acc["launches"] += st["bounce_launches"]
        acc["kernel_ms"] += st["kernel_ms"]
        acc["segments"] += st["segments"]
        acc["samples"] += st["samples"]

    def barrier():
        if world > 1 or env["force"]:
            dist.barrier()
        torch.cuda.synchronize()

    keep = {}
    # this rank's own clock, split at the collective: the render / acquisition (library calls, synchronous on return) and the
    # gather / reduce (which includes the wait for the slowest rank) -- rank 0's line lists every rank's pair (per_rank_ms,
    # collective_ms), so a band imbalance or a slow link shows in the record of the run itself
    mine = dict(work=0.0, coll=0.0)

    def step():
        t_a = time.perf_counter()
        if radiance:
            try:
                tile, layout = par.render_tiles(scene, SPP, seed, rank, world, band_rows, device=device, on_call=account,
                                                tile=keep.get("tile"), pass_paths=cfg.get("pass_paths", 0) if world == 1 else 0)
            except RuntimeError as e:
                # the one-pass size of config 4 needs 183 GB: on a device that cannot give them, go on with the library's default
                # (and say so in the record)
                if not (world == 1 and cfg.get("pass_paths")):
                    raise
                cfg["pass_paths"] = 0
                cfg["what"] += f" [pass_paths request refused: {str(e)[:80]}; library default used]"
                ctx.trim()
                tile, layout = par.render_tiles(scene, SPP, seed, rank, world, band_rows, device=device, on_call=account,
                                                tile=keep.get("tile"))
            keep["tile"] = tile
            if args.rehearse_on_one_gpu and world > 1:
                tile = tile.cpu()  # gloo gathers host tensors
            torch.cuda.synchronize()
            t_b = time.perf_counter()
            if "parts" not in keep:  # the gather list of rank 0, allocated once, outside the timed steps
                keep["parts"] = par.gather_buffers(tile, rank, world, force_collective=env["force"])
            film = par.gather_film(tile, layout, RES, RES, rank, world, force_collective=env["force"], parts=keep["parts"])
            if tile.is_cuda:
                torch.cuda.synchronize()
            mine["work"] += t_b - t_a
            mine["coll"] += time.perf_counter() - t_b
            return film
        tm = {}
        buf = par.distributed_acquire(scene, PPR, seed=seed, device=device, on_call=account,
                                      host_collective=bool(args.rehearse_on_one_gpu and world > 1), force_collective=env["force"],
                                      timing=tm)
        mine["work"] += tm.get("work_s", time.perf_counter() - t_a)
        mine["coll"] += tm.get("collective_s", 0.0)
        return buf

    # Python's cyclic collector stays out of the timed steps: after the CPU legs of the entry before -- hundreds of MB of NumPy
    # results -- a generation-2 pass landed inside one timed step of every other run, 40 - 60 ms of host time in a 18 ms step.
    # Collected BEFORE the warm-up steps, not between them and the timed ones: the collection is a pause of the host, and the
    # first steps after a pause run on low clocks (tools/idle_gap_probe.py; a collection right in front of the timed steps made
    # the 6 ms headline step 2.6 % and the 10 ms acquisition 9 % slower)
    import gc
    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    for _ in range(warmup):
        step()
    if warmup == 0 and radiance and (world > 1 or env["force"]):  # no warm-up step to allocate the gather list in
        tile0, _ = par.render_tiles(scene, SPP, seed, rank, world, band_rows, device=device, render_band=lambda crop, view: None)
        keep["tile"] = tile0
        keep["parts"] = par.gather_buffers(tile0.cpu() if (args.rehearse_on_one_gpu and world > 1) else tile0, rank, world,
                                           force_collective=env["force"])
    for k in acc:
        acc[k] = 0
    mine["work"] = mine["coll"] = 0.0
    barrier()
    t0 = time.perf_counter()
    step_s = []
    for _ in range(steps):
        t_s = time.perf_counter()
        result = step()
        step_s.append(time.perf_counter() - t_s)
    barrier()
    dt = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    if os.environ.get("BENCH_DEBUG_STEPS"):  # where does a step's wall-clock go?  (stderr; not part of the line)
        print(f"[bench] {name}: wall per step {[round(x * 1e3, 2) for x in step_s]} ms, kernel ms summed {acc['kernel_ms']:.2f}, "
              f"rank work {mine['work'] * 1e3:.2f} ms, collective {mine['coll'] * 1e3:.2f} ms", file=sys.stderr, flush=True)
    per_rank = None
    if world > 1:
        cdev = "cpu" if args.rehearse_on_one_gpu else device
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        pr = torch.tensor([mine["work"], mine["coll"]], dtype=torch.float64, device=cdev)
        allpr = [torch.zeros_like(pr) for _ in range(world)]
        dist.all_gather(allpr, pr)
        per_rank = [[float(x[0]) / steps * 1e3, float(x[1]) / steps * 1e3] for x in allpr]
    elif env["force"]:
        per_rank = [[mine["work"] / steps * 1e3, mine["coll"] / steps * 1e3]]

    out = None
    if rank == 0:
        ms = dt / steps * 1e3
        value = total_units / (dt / steps) / 1e6
        achieved = (acc["bounce_bytes"] / 1e9) / (acc["bounce_ms"] / 1e3) if acc["bounce_ms"] > 0 else 0.0
        traffic, traffic_source = None, "not measured by this run (HBM bytes need separate rocprofv3 --pmc passes)"
        traffic_by_kernel = None
        valu_busy = lane_active = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        src_hash = kernel_source_hash()
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc)).get(name)
                if rec and rec.get("kernel_source_sha16") == src_hash:
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_by_kernel = rec.get("traffic_over_algorithmic_by_kernel")
                    valu_busy = rec.get("valu_issue_busy")
                    lane_active = rec.get("lane_active")
                    traffic_source = f"profiles/pmc_traffic.json[{name}] recorded on these kernel sources ({src_hash}): {rec.get('how', '')}"
                    cal = rec.get("calibration_file")
                    if cal and not os.path.exists(os.path.join(ROOT, cal)):
                        traffic_source += f"  [calibration file {cal} is missing]"
                elif rec:
                    traffic_source += f"; profiles/pmc_traffic.json[{name}] is for sources {rec.get('kernel_source_sha16')}, this run uses {src_hash}"
            except Exception:
                pass
        if radiance:
            sharding = ("whole film on one GPU" if world == 1 and band_rows >= RES else
                        f"interleaved {band_rows}-row bands dealt round-robin to {world} rank(s), one gather of the finished bands to rank 0")
            workload = (f"{cfg['scene']} {RES}x{RES}, {SPP} spp, path max_depth {cfg['max_depth']}, {cfg['what']}; {sharding}")
            if world == 1 and cfg.get("pass_paths"):
                workload += f"; pass_paths {cfg['pass_paths'] >> 20} Mi asked for by the caller (pbrt_film_desc.pass_paths)"
        else:
            sharding = "all paths on one GPU" if world == 1 else f"contiguous path ranges per rank, one reduce(sum) of the channel buffer to rank 0"
            workload = f"{cfg['scene']} ({cfg['what']}), {PPR} paths per ray; {sharding}"
        if overridden:
            workload += "  [SIZE OVERRIDDEN ON THE COMMAND LINE: not the named BASELINE workload]"
        kernel = (("k_trace + k_us_shade" if name == "us_testring" else "k_us_bounce") if not radiance
                  else ("k_trace_primary + k_trace + k_shade" if name == "testring" else "k_bounce"))
        valu_frac = round(valu_busy * lane_active, 4) if (valu_busy is not None and lane_active is not None) else None
        alg_per_launch = acc["bounce_bytes"] / max(acc["launches"], 1)
        out = {
            "metric": cfg["metric"], "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak" if (world == 1 and name == "cbox") else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "baseline_config": cfg["baseline_config"], "samples_per_step": total_units, "seed": seed,
                       "mean_segments_per_sample": round(acc["segments"] / max(acc["samples"], 1), 4)},
            # The contract's roofline: ALGORITHMIC bytes of the bounce launches (DESIGN.md byte model, evaluated on the live-path
            # counters of this very run) / their HIP-event durations on the library's stream, against the 8 TB/s of HBM3E --
            # measured by every run, always numeric.  What these kernels actually run against is VALU issue (DESIGN.md section
            # 7); those figures come from PMC passes and ride along as valu_* keys when they were recorded for these sources.
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_over_algorithmic": round(traffic / alg_per_launch, 3) if (traffic and alg_per_launch) else None,
                         "traffic_over_algorithmic_by_kernel": traffic_by_kernel,
                         "algorithmic_bytes_per_launch": round(alg_per_launch),
                         "avg_launch_ms": round(acc["bounce_ms"] / max(acc["launches"], 1), 5), "launches": acc["launches"],
                         "kernel_ms_per_step": round(acc["kernel_ms"] / steps, 3), "scope": "rank 0's launches of the timed steps",
                         "valu_frac": valu_frac, "valu_issue_busy": valu_busy, "lane_active": lane_active,
                         "valu_peak": round(VALU_PEAK_TLANEOPS, 2), "valu_unit": "T f32 lane-op/s",
                         "valu_achieved": round(valu_frac * VALU_PEAK_TLANEOPS, 2) if valu_frac is not None else None,
                         "valu_source": ("SQ_INSTS_VALU / (16 x SQ_BUSY_CYCLES) and SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) of the "
                                         "dominant kernels, 4th pass of tools/profile_bench.sh (a wave64 VALU instruction occupies its SIMD-32 "
                                         "for 2 cycles; 1024 SIMDs, the counters are summed over 32 shader engines)"
                                         if valu_busy is not None else "not recorded for these kernel sources")},
        }
        if per_rank is not None:
            out["per_rank_ms"] = [round(x[0], 3) for x in per_rank]     # render / acquisition per step, rank by rank
            out["collective_ms"] = [round(x[1], 3) for x in per_rank]   # gather / reduce per step incl. the wait for the slowest rank
            out["gather_ms"] = round(per_rank[0][1], 3)                  # rank 0's: it ends when the last band has arrived
        if radiance and name in ("cbox", "cbox4k"):
            out["roofline"]["note"] = ("the brute-force bounce kernels walk up to six bounces of a path in registers (the library picks the chain "
                                       "lengths from the path survival of the scene's last render; Cornell box: one launch per pass), so the "
                                       "only HBM traffic left is one 16-byte radiance record per path (hbm_* keys)")
        if name in ("testring", "us_testring"):
            # per step, so that the PMC bytes of the two kernel families (profiles/pmc_traffic.json) have their own denominators
            walk, shade = ("k_trace_primary + k_trace", "k_shade") if radiance else ("k_trace", "k_us_shade")
            out["roofline"]["algorithmic_bytes_per_step"] = {walk: round(acc["trace_bytes"] / steps),
                                                             shade: round((acc["bounce_bytes"] - acc["trace_bytes"]) / steps)}
        if radiance and name == "testring":
            out["roofline"]["note"] = ("a bounce is two launches: k_trace (stream of closest-hit and shadow queries against the LDS-resident BVH4, "
                                       "8 waves per SIMD; the camera rays: k_trace_primary, one tree walk per 64-path tile) and k_shade (full "
                                       "waves, HBM-bound: 96-byte path state, 32-byte shadow rays, hit records)")
        if not radiance and name == "us_testring":
            out["roofline"]["note"] = ("ultrasound on a BVH scene: k_trace (closest hits + the unbounded occlusion rays towards the receive elements) "
                                       "and k_us_shade per bounce; depth 0 comes from the first-bounce tables; SURVEY 8(d)'s 'natural extra' of config 4")
        elif not radiance:
            out["roofline"]["note"] = ("k_us_bounce (GGX / impedance sample, expf, sinf, acosf; one launch walks every bounce of a pass)")
        if args.rehearse_on_one_gpu and world > 1:
            out["rehearsal"] = rehearsal_check(mi, np, scene, cfg, result, seed, world)
        if env["force"]:
            out["collective"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                 "what": "gather / reduce issued through torch.distributed at world size 1 (RCCL in this process)"}
        if with_cpu:
            out.update(cpu_baseline(mi, np, scene, cfg, name, seed, args, cpu_seconds))
    return out


def measure_usmain_loop(env, cfg, steps, warmup, with_cpu):
    """One step = one iteration of the reference's finite-difference loop (USMain.py:279-283): forward(rough) and
    forward(rough + eps), each params.update() + us_render (acquisition -> DAS -> envelope -> log compression -> the display image
    on the host).  Timed at paths_per_ray 1 (the reference's own: one ray per (angle, element)), 64 and 4096; the value of the line is
    the first.  Per size: wall-clock per iteration split into update / acquire / image formation (queueing) / wait + copy, the device
    time of every image-formation kernel (HIP events on the library's stream, a separate profiled loop), and the same iteration
    through round 4's host-pointer chain (device_resident=False) as the before / after."""
    mi, np, torch = env["mi"], env["np"], env["torch"]
    scene = mi.load_file(os.path.join(SCENES, cfg["scene"]))
    scene.device()
    ctx = mi.default_context()
    integ = scene.integrator()
    params = mi.traverse(scene)
    key = [k for k in params.keys() if k.endswith("flat_plate.bsdf.roughness")][0]
    sizes = []

    def iteration(ppr, rough, acc, kernel_stats=False, **kw):
        for r in (rough, rough + 1e-3):                                         # USMain.py:280-282
            t0 = time.perf_counter()
            params[key] = r
            params.update()                                                     # :264-265
            t1 = time.perf_counter()
            tm = {}
            img = mi.us_render(scene, seed=0, paths_per_ray=ppr, return_bmode=False, timing=tm, **kw)[0]
            t2 = time.perf_counter()
            acc["update"] += t1 - t0
            acc["render"] += t2 - t1
            for k, v in tm.items():
                acc[k] = acc.get(k, 0.0) + v
            if kernel_stats:  # (asks the library for the statistics of the queued acquisition: that waits for the stream)
                acc["acquire_kernel_ms"] += ctx.stats()["kernel_ms"]
        return img

    for ppr in cfg["ppr"]:
        acc = dict(update=0.0, render=0.0, acquire_kernel_ms=0.0)
        for _ in range(warmup):
            iteration(ppr, 0.1, acc)
        acc = dict(update=0.0, render=0.0, acquire_kernel_ms=0.0)
        torch.cuda.synchronize()
        ctx.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            img = iteration(ppr, 0.1 + 0.01 * i, acc)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        # the same loop with every call of the chain queued one by one (no recording): the before / after of the hipGraph replay
        t0 = time.perf_counter()
        for i in range(steps):
            iteration(ppr, 0.1 + 0.01 * i, dict(update=0.0, render=0.0, acquire_kernel_ms=0.0), graph=False)
        ctx.synchronize()
        dt_plain = time.perf_counter() - t0
        # the same iteration with an event pair around every image-formation kernel (not part of the timed loop above)
        ctx.set_profiling(True)
        dev = dict(das_ms=0.0, envelope_ms=0.0, log_ms=0.0)
        nprof = min(steps, 5)
        das_bytes = 0
        pacc = dict(update=0.0, render=0.0, acquire_kernel_ms=0.0)
        for i in range(nprof):
            iteration(ppr, 0.1 + 0.01 * i, pacc, kernel_stats=True)
            st = ctx.image_stats()
            for k in dev:
                dev[k] += st[k]
            das_bytes = st["das_model_bytes"]
        ctx.set_profiling(False)
        # before: round 4's chain through the host-pointer entry points (H2D + kernel + D2H + sync per step)
        old = dict(update=0.0, render=0.0, acquire_kernel_ms=0.0)
        nold = max(2, min(steps, 5))
        iteration(ppr, 0.1, dict(update=0.0, render=0.0, acquire_kernel_ms=0.0), device_resident=False)
        tb = time.perf_counter()
        for i in range(nold):
            iteration(ppr, 0.1 + 0.01 * i, old, device_resident=False)
        t_old = (time.perf_counter() - tb) / nold
        n_r = 2 * steps
        rec = {"paths_per_ray": ppr, "ms_per_iteration": round(dt / steps * 1e3, 4),
               "renders_replayed_from_the_recording": int(acc.get("replayed", 0)), "renders": 2 * steps,
               "ms_per_iteration_calls_queued_one_by_one": round(dt_plain / steps * 1e3, 4),
               "per_render_ms": {"params_update": round(acc["update"] / n_r * 1e3, 4), "acquire_queueing": round(acc["acquire"] / n_r * 1e3, 4),
                                 "image_formation_queueing": round(acc["queue"] / n_r * 1e3, 4),
                                 "wait_and_copy_of_the_image": round(acc["wait_copy"] / n_r * 1e3, 4)},
               "device_ms_per_render": {"acquisition_kernels": round(pacc["acquire_kernel_ms"] / (2 * nprof), 4), "das": round(dev["das_ms"] / nprof, 4),
                                        "envelope": round(dev["envelope_ms"] / nprof, 4), "log_compression": round(dev["log_ms"] / nprof, 4)},
               "host_pointer_chain_ms_per_iteration": round(t_old * 1e3, 4),
               "speedup_vs_host_pointer_chain": round(t_old / (dt / steps), 3),
               "image": list(img.shape), "das_model_bytes": das_bytes,
               "das_gbs": round(das_bytes / 1e9 / (dev["das_ms"] / nprof / 1e3), 2) if dev["das_ms"] > 0 else None}
        sizes.append(rec)
    first = sizes[0]
    das_ms = first["device_ms_per_render"]["das"]
    achieved = first["das_gbs"] or 0.0
    # HBM bytes of one k_das_beamform launch from the PMC passes over THIS command (tools/profile_bench.sh r05_usmain --config usmain_loop),
    # if they were recorded on these kernel sources
    das_traffic, das_traffic_source = None, "not measured by this run (HBM bytes need separate rocprofv3 --pmc passes)"
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("usmain_loop")
        if rec and rec.get("kernel_source_sha16") == kernel_source_hash():
            das_traffic = rec.get("hbm_bytes_per_launch")
            das_traffic_source = f"profiles/pmc_traffic.json[usmain_loop] recorded on these kernel sources: {rec.get('how', '')}"
        elif rec:
            das_traffic_source += f"; profiles/pmc_traffic.json[usmain_loop] is for sources {rec.get('kernel_source_sha16')}"
    except Exception:
        pass
    out = {"metric": cfg["metric"], "value": first["ms_per_iteration"], "unit": "ms", "n_gpus": 1, "steps": steps, "warmup": warmup,
           "ms_per_step": first["ms_per_iteration"], "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (sample positions f64)",
           "data": "synthetic",
           "config": {"workload": f"{cfg['scene']} ({cfg['what']}); one step = forward(rough) + forward(rough + 1e-3), image {first['image'][1]} x {first['image'][0]} pixels; "
                                  "replicas only (a serial optimisation loop)", "paths_per_ray": list(cfg["ppr"])},
           "roofline": {"bound": "hbm", "kernel": "k_das_beamform", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": das_traffic, "traffic_source": das_traffic_source,
                        "algorithmic_bytes_per_launch": first["das_model_bytes"], "avg_launch_ms": das_ms,
                        "note": "algorithmic bytes = the channel buffer once + the image once; the kernel gathers 2 x n_angles x |aperture| samples per pixel "
                                "from L2 at f64 sample positions: it runs against the texture / L2 gather rate and f64 issue, not HBM"},
           "sizes": sizes}
    if with_cpu:
        from oracle import beamform as obf
        from oracle import binding as ob
        osc = ob.OracleScene.from_scene(scene)
        A, E, T = integ.n_angles, integ.n_elements, integ.time_samples
        lam = integ.sound_speed / integ.frequency
        xs = np.arange(-0.04, 0.04 + lam / 4, lam / 4)
        zs = np.arange(0.001, 0.05 + lam / 4, lam / 4)
        tc = time.perf_counter()
        ref, tx = osc.us_acquire(integ.us_params(scene), 0, 64)
        t_acq = time.perf_counter() - tc
        ex = integ.pitch * (np.arange(E, dtype=np.float32) - (E - 1) / 2)
        tc = time.perf_counter()
        rf = obf.das_beamform(ref.reshape(A, E, T), np.asarray(tx).reshape(A, E), ex, xs, zs, integ.fs, integ.sound_speed)
        envl = obf.envelope(rf)
        img_ref = obf.log_compress(envl, 60.0).T
        t_img = time.perf_counter() - tc
        display = mi.us_render(scene, seed=0, paths_per_ray=64, return_bmode=False)[0]
        out["cpu_baseline"] = {"value": round(2 * (t_acq + t_img) * 1e3, 1), "unit": "ms", "cores": 1, "kind": "port",
                               "sample": f"ONE us_render at paths_per_ray 64 (C++ oracle acquisition {t_acq:.2f} s + numpy f64 image formation {t_img:.1f} s "
                                         f"on the {len(xs)} x {len(zs)} grid), doubled for the two renders of an iteration"}
        out["l2_vs_cpu_ref"] = {"display_max_abs": float(np.abs(display - img_ref).max()),
                                "display_rmse": float(np.sqrt(np.mean((display - img_ref) ** 2))), "compared_on": "the display image at paths_per_ray 64, seed 0"}
    return out


def seed_parity(env, cfg):
    """seeds 1 and 2 of the headline scene against the CPU port at 16 spp (bit for bit, like seed 0)"""
    mi, np = env["mi"], env["np"]
    from oracle import binding as ob
    scene = mi.load_file(os.path.join(SCENES, cfg["scene"]), res=cfg["res"], spp=16, max_depth=cfg["max_depth"])
    integ, sens = scene.integrator(), scene.sensors()[0]
    osc = ob.OracleScene.from_scene(scene)
    cores = env["args"].cpu_threads or min(len(os.sched_getaffinity(0)), 64)
    rec = {}
    for sd in (1, 2):
        ref = osc.render(sens.camera(), integ._film_desc(scene, sens, sd, 16), n_threads=cores)
        img = integ.render(scene, seed=sd, spp=16)
        d = img.astype(np.float64) - ref.astype(np.float64)
        rec[str(sd)] = {"rmse": float(np.sqrt(np.mean(d * d))), "bit_exact_fraction": float(np.mean(img == ref)),
                        "compared_on": f"{cfg['res']}x{cfg['res']} x 16 spp"}
    return rec


def rehearsal_check(mi, np, scene, cfg, result, seed, world):
    """the stitched / reduced result of the sharded job against the unsharded library call on the same device"""
    got = result.cpu().numpy()
    if cfg["kind"] == "radiance":
        y0 = (cfg["res"] // 2 // 64) * 64
        crop = (0, y0, cfg["res"], min(128, cfg["res"] - y0))  # two bands that belong to different ranks
        whole = scene.integrator().render(scene, seed=seed, spp=cfg["spp"], crop=crop)
        return {"backend": "gloo", "ranks_on_device_0": world, "compared": f"rows {crop[1]}..{crop[1] + crop[3]} of the stitched film vs the unsharded render of that crop",
                "stitched_equals_unsharded": bool(np.array_equal(got[crop[1]:crop[1] + crop[3]], whole))}
    ui = scene.integrator()
    whole = ui._acquire(scene, ui.quirks, paths_per_ray=cfg["ppr"], seed=seed)
    return {"backend": "gloo", "ranks_on_device_0": world,
            "rel_l2_vs_unsharded": float(np.linalg.norm(got - whole) / (np.linalg.norm(whole) + 1e-30))}


def cpu_baseline(mi, np, scene, cfg, name, seed, args, cpu_seconds=None):
    """The CPU oracle on a bounded sample of the same workload, and the L2 between the two results on that sample."""
    from oracle import binding as ob
    cores = args.cpu_threads or min(len(os.sched_getaffinity(0)), 64)
    budget = cpu_seconds if cpu_seconds is not None else args.cpu_seconds
    osc = ob.OracleScene.from_scene(scene)
    if cfg["kind"] == "radiance":
        integ, sens = scene.integrator(), scene.sensors()[0]
        RES, SPP = cfg["res"], cfg["spp"]
        # bounded sample: the full film at reduced spp (cbox), or a centred crop at reduced spp for the big films
        crop = None if RES <= 512 else (RES // 2 - 128, RES // 2 - 128, 256, 256)
        px = RES * RES if crop is None else crop[2] * crop[3]
        tc = time.perf_counter()
        osc.render(sens.camera(), integ._film_desc(scene, sens, seed, 2, crop=crop), n_threads=cores)
        per_spp = (time.perf_counter() - tc) / 2
        spp = 1
        while spp < SPP and per_spp * spp * 2 <= budget:
            spp *= 2
        fd = integ._film_desc(scene, sens, seed, spp, crop=crop)
        tc = time.perf_counter()
        ref = osc.render(sens.camera(), fd, n_threads=cores)
        tcpu = time.perf_counter() - tc
        img = integ.render(scene, seed=seed, spp=spp, crop=crop)
        d = img.astype(np.float64) - ref.astype(np.float64)
        where = f"{RES}x{RES}" if crop is None else f"the centred {crop[2]}x{crop[3]} crop of the {RES}x{RES} film"
        # single-thread figure: the same film on one thread, about 3 s of it (1 spp first, to size the sample)
        ts = time.perf_counter()
        osc.render(sens.camera(), integ._film_desc(scene, sens, seed, 1, crop=crop), n_threads=1)
        t1 = time.perf_counter() - ts
        spp1 = int(max(1, min(spp, 3.0 / max(t1, 1e-3))))
        if spp1 > 1:
            ts = time.perf_counter()
            osc.render(sens.camera(), integ._film_desc(scene, sens, seed, spp1, crop=crop), n_threads=1)
            t1 = time.perf_counter() - ts
        return {"cpu_baseline": {"value": round(px * spp / tcpu / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
                                 "sample": f"{cfg['scene']} {where} x {spp} spp (of {SPP}), {tcpu:.1f} s, C++ oracle, std::thread over rows",
                                 "single_thread": {"value": round(px * spp1 / t1 / 1e6, 4), "unit": "Msamples/s", "cores": 1,
                                                   "sample": f"{where} x {spp1} spp, {t1:.1f} s"}},
                "l2_vs_cpu_ref": {"rmse": float(np.sqrt(np.mean(d * d))), "max_abs": float(np.abs(d).max()),
                                  "bit_exact_fraction": float(np.mean(img == ref)), "tolerance": 1e-3,
                                  "compared_on": f"{where} x {spp} spp, seed {seed}"}}
    ui = scene.integrator()
    n_rays = ui.n_angles * ui.n_elements
    tc = time.perf_counter()
    osc.us_acquire(ui.us_params(scene), seed, 64)
    per_path = (time.perf_counter() - tc) / 64
    ppr = 64
    while ppr < cfg["ppr"] and per_path * ppr * 2 <= budget:
        ppr *= 2
    tc = time.perf_counter()
    ref, _ = osc.us_acquire(ui.us_params(scene), seed, ppr)
    tcpu = time.perf_counter() - tc
    buf = ui._acquire(scene, ui.quirks, paths_per_ray=ppr, seed=seed)
    d = buf.astype(np.float64) - ref.astype(np.float64)
    return {"cpu_baseline": {"value": round(n_rays * ppr / tcpu / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
                             "sample": f"{cfg['scene']} {n_rays} rays x {ppr} paths (of {cfg['ppr']}), {tcpu:.1f} s, C++ oracle, one thread "
                                       f"(path order is the summation order of its f64 accumulators)"},
            "l2_vs_cpu_ref": {"rel_l2": float(np.linalg.norm(d) / (np.linalg.norm(ref.astype(np.float64)) + 1e-300)),
                              "max_abs_over_max_ref": float(np.abs(d).max() / (np.abs(ref).max() + 1e-300)),
                              "same_nonzero_bins": bool(np.array_equal(buf != 0, ref != 0)), "tolerance": 1e-3,
                              "compared_on": f"{n_rays} rays x {ppr} paths, seed {seed}"}}


if __name__ == "__main__":
    main()
