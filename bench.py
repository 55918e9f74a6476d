#!/usr/bin/env python3
"""Headline benchmark: Msamples/s on the Cornell box of BASELINE config 2 (scenes/cbox.xml geometry,
512 x 512, 256 spp per GPU, max_depth 6, tent filter), radiance mode, on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

One step = one full render.  Weak scaling: the film stays 512 x 512 and the job renders spp = 256 * N; rank r
traces samples [256 r, 256 (r + 1)) of EVERY pixel (the same RNG keys as a single-GPU render of all 256 * N
samples) into un-normalised accumulators, so per-GPU work is exactly the single-GPU workload (512*512*256
samples) and the job total is 512*512*256*N samples.  Scene and all path state are resident in HBM; the
accumulators (4 MB per rank) are added on rank 0 with ONE RCCL reduce(sum) inside the timed region, and rank 0
divides.  (The band-sharded split of parallel.py -- bit-identical to the single-GPU film, one gather -- is the
one for large films; for a small film at high spp its per-rank crops are too small to fill a GPU's film kernel.)
Rank 0 prints ONE JSON line.

The line also carries
  roofline     -- the dominant kernel (k_bounce, one launch per bounce per pass): algorithmic HBM bytes of
                  its launches (DESIGN.md byte model, counted from the live-path counters of the run)
                  divided by their HIP-event durations on the library's stream, against 8 TB/s.
  cpu_baseline -- the CPU oracle (port of the same algorithm, same RNG) on this box's host cores, timed on
                  a bounded sample (512 x 512 x 32 spp of the 256), rank 0, N = 1 only; the same render
                  gives the per-pixel L2 between the HIP film and the CPU film.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RES, SPP_PER_GPU, MAX_DEPTH = 512, 256, 6
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=0, help="0: pick the largest power of two <= 256 that takes about 15 s")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0: min(visible cores, 64)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a 1-GPU box: every rank uses device 0 and the gather runs over gloo "
                         "(host staging); exercises sharding + stitching, NOT RCCL -- numbers are not benchmark numbers")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched through torch.distributed.run (one rank per GPU)")
        args.gpus = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    os.environ["PBRT_DEVICE"] = str(local_rank)

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the ray-transport hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    mi = importlib.import_module("physics-based-ray-tracing_amd")
    par = importlib.import_module("physics-based-ray-tracing_amd.parallel")
    spp = SPP_PER_GPU * world
    scene = mi.load_file(os.path.join(ROOT, "tests", "scenes", "cbox.xml"), res=RES, spp=spp, max_depth=MAX_DEPTH)
    scene.device()  # upload once, outside the timed region
    ctx = mi.default_context()
    band_rows = RES
    seed = 0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        if world == 1:
            tile, layout = par.render_tiles(scene, spp, seed, rank, world, band_rows, device=device)
            return par.gather_film(tile, layout, RES, RES, rank, world), tile
        raw = par.render_sample_shard(scene, spp, seed, rank, world, device=device)
        if args.rehearse_on_one_gpu:
            raw = raw.cpu()  # gloo reduces host tensors
        return par.reduce_film(raw, rank, world), raw

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    bounce_ms = bounce_bytes = kernel_ms = 0.0
    launches = 0
    segments = samples = 0
    for _ in range(args.steps):
        film, _ = step()
        # per-call statistics of this rank (its last band); accumulated for the roofline of rank 0
        st = ctx.stats()
        bounce_ms += st["bounce_ms"]
        bounce_bytes += st["bounce_model_bytes"]
        launches += st["bounce_launches"]
        kernel_ms += st["kernel_ms"]
        segments += st["segments"]
        samples += st["samples"]
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    total_samples = RES * RES * spp
    out = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = total_samples / (dt / args.steps) / 1e6
        achieved = (bounce_bytes / 1e9) / (bounce_ms / 1e3) if bounce_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("k_bounce_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Msamples/s on cbox.xml 512x512 x 256 spp (radiance, path max_depth 6)",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cbox.xml {RES}x{RES}, {SPP_PER_GPU} spp per GPU (spp={spp}), path max_depth {MAX_DEPTH}, "
                                   f"tent filter, 6 analytic quads + 2 spheres; "
                                   + ("whole film on one GPU" if world == 1 else
                                      f"rank r traces samples [{SPP_PER_GPU} r, {SPP_PER_GPU} (r + 1)) of every pixel, one reduce(sum) of the accumulators"),
                       "samples_per_step": total_samples, "seed": seed,
                       "mean_segments_per_sample": round(segments / max(samples, 1), 4)},
            "roofline": {"bound": "hbm", "kernel": "k_bounce", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": round(bounce_bytes / max(launches, 1)),
                         "avg_launch_ms": round(bounce_ms / max(launches, 1), 5), "launches": launches,
                         "kernel_ms_per_step": round(kernel_ms / args.steps, 3)},
        }
        if args.rehearse_on_one_gpu and world > 1:
            # the reduced film of the sharded job against the un-sharded render of all the samples (same samples,
            # the partial sums are added in another order)
            whole = scene.integrator().render(scene, seed=seed, spp=spp)
            got = film.cpu().numpy()
            out["rehearsal"] = {"backend": "gloo", "ranks_on_device_0": world,
                                "max_rel_diff_vs_unsharded": float(np.max(np.abs(got - whole) / np.maximum(np.abs(whole), 1e-3))),
                                "reduced_matches_unsharded": bool(np.allclose(got, whole, rtol=1e-5, atol=1e-6))}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import binding as ob
            cores = args.cpu_threads or min(len(os.sched_getaffinity(0)), 64)
            integ, sens = scene.integrator(), scene.sensors()[0]
            osc = ob.OracleScene.from_scene(scene)
            if not args.cpu_spp:  # calibrate on 2 spp, then take ~15 s worth of samples (bounded by the full 256)
                tc = time.perf_counter()
                osc.render(sens.camera(), integ._film_desc(scene, sens, seed, 2), n_threads=cores)
                per_spp = (time.perf_counter() - tc) / 2
                args.cpu_spp = 1
                while args.cpu_spp < SPP_PER_GPU and per_spp * args.cpu_spp * 2 <= 15.0:
                    args.cpu_spp *= 2
            fd = integ._film_desc(scene, sens, seed, args.cpu_spp)
            tc = time.perf_counter()
            ref = osc.render(sens.camera(), fd, n_threads=cores)
            tcpu = time.perf_counter() - tc
            img = integ.render(scene, seed=seed, spp=args.cpu_spp)
            d = img.astype(np.float64) - ref.astype(np.float64)
            out["cpu_baseline"] = {"value": round(RES * RES * args.cpu_spp / tcpu / 1e6, 4), "unit": "Msamples/s",
                                   "cores": cores, "kind": "port",
                                   "sample": f"cbox.xml {RES}x{RES} x {args.cpu_spp} spp (of {SPP_PER_GPU}), {tcpu:.1f} s, "
                                             f"C++ oracle, std::thread over rows"}
            out["l2_vs_cpu_ref"] = {"rmse": float(np.sqrt(np.mean(d * d))), "max_abs": float(np.abs(d).max()),
                                    "bit_exact_fraction": float(np.mean(img == ref)), "tolerance": 1e-3,
                                    "compared_on": f"{RES}x{RES} x {args.cpu_spp} spp, seed {seed}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
