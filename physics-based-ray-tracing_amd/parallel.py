"""Multi-GPU sharding of the hot path: one process per GPU (torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" on CPU for tests).

Radiance mode  -- the film is cut into interleaved bands of `band_rows` rows; rank r owns bands
    r, r + G, r + 2G, ...  Each rank renders its bands with the SAME global RNG keys (pixel, sample) as
    a single-GPU render; the library renders the filter halo rows around a band redundantly, so the
    stitched image is bit-identical to the single-GPU image.  No collective on the data path; ONE
    gather of finished bands to rank 0 at the end.
Ultrasound mode -- the P paths of every (angle, element) ray are split into contiguous ranges; every
    rank accumulates its own channel buffer (already divided by the total P) and ONE reduce(sum) adds
    them.
"""
from __future__ import annotations

import numpy as np


def band_layout(height: int, world_size: int, band_rows: int = 64):
    """-> list over ranks of [(y0, rows), ...]; bands dealt round-robin."""
    bands = [(y, min(band_rows, height - y)) for y in range(0, height, band_rows)]
    return [[b for i, b in enumerate(bands) if i % world_size == r] for r in range(world_size)]


def rows_of(layout_rank):
    return sum(r for _, r in layout_rank)


def path_ranges(paths_per_ray: int, world_size: int):
    """-> [(offset, count)] contiguous split of the per-ray path index"""
    base, rem = divmod(paths_per_ray, world_size)
    out, off = [], 0
    for r in range(world_size):
        c = base + (1 if r < rem else 0)
        out.append((off, c))
        off += c
    return out


def _dist():
    import torch.distributed as dist
    return dist


def render_tiles(scene, spp, seed, rank, world_size, band_rows=64, render_band=None, device=None, sample_offset=0,
                 on_call=None, tile=None, pass_paths=0):
    """Render this rank's bands into one [rows_max, W, 3] float32 torch tensor (padded to the largest
    rank).  render_band(crop, out_rows_tensor) fills a [rows, W, 3] view; the default calls the HIP
    library and writes straight into the tensor's device memory (no PCIe traffic).  on_call() runs after
    every library call (one per band; bench.py adds up the per-call statistics there).  tile: a tensor of a previous call
    to render into again (every row of a rank's bands is overwritten).  pass_paths: paths in flight per pass (0: the library's
    default, which leaves half of the device to other tenants; a process that owns its GPU may ask for more)."""
    import torch

    sens = scene.sensors()[0]
    W, H = sens.film().size()
    layout = band_layout(H, world_size, band_rows)
    rows_max = max(rows_of(l) for l in layout)
    dev = device if device is not None else torch.device("cpu")
    if tile is None or tuple(tile.shape) != (rows_max, W, 3) or tile.device != dev:
        tile = torch.zeros((rows_max, W, 3), dtype=torch.float32, device=dev)
    integ = scene.integrator()
    off = 0
    for (y0, rows) in layout[rank]:
        view = tile[off:off + rows]
        crop = (0, y0, W, rows)
        if render_band is not None:
            render_band(crop, view)
        else:
            if dev.type != "cuda":
                raise RuntimeError("the HIP render path needs a device tensor (torch 'cuda' == HIP on ROCm)")
            integ.render(scene, sensor=sens, seed=seed, spp=spp, crop=crop, sample_offset=sample_offset,
                         out_dev=view.data_ptr(), pass_paths=pass_paths)
            if on_call is not None:
                on_call()
        off += rows
    return tile, layout


def gather_buffers(tile, rank, world_size, force_collective=False):
    """The gather list of rank 0 (one tensor per rank, shaped like `tile`), to be allocated ONCE and handed to every
    gather_film call of a loop; None on the other ranks and when no collective runs."""
    import torch

    if (world_size == 1 and not force_collective) or rank != 0:
        return None
    return [torch.empty_like(tile) for _ in range(world_size)]


def gather_film(tile, layout, width, height, rank, world_size, group=None, force_collective=False, parts=None):
    """One gather of every rank's tile to rank 0, then de-interleave.  -> [H, W, 3] tensor on rank 0, None elsewhere.
    force_collective: issue the gather even at world_size 1 (a process group must be initialised): runs the collective
    library on a box with one GPU.  parts: rank 0's gather list from gather_buffers (allocated per call when None)."""
    import torch

    dist = _dist()
    if world_size == 1 and not force_collective:
        parts = [tile]
    else:
        if rank == 0 and (parts is None or len(parts) != world_size or parts[0].shape != tile.shape or parts[0].device != tile.device):
            parts = gather_buffers(tile, rank, world_size, force_collective=True)
        dist.gather(tile, gather_list=parts if rank == 0 else None, dst=0, group=group)
        if rank != 0:
            return None
    film = torch.empty((height, width, 3), dtype=tile.dtype, device=tile.device)
    for r in range(world_size):
        off = 0
        for (y0, rows) in layout[r]:
            film[y0:y0 + rows] = parts[r][off:off + rows]
            off += rows
    return film


def distributed_render(scene, spp, seed=0, band_rows=64, render_band=None, device=None, group=None):
    """Whole job: render this rank's bands, gather on rank 0.  Works for world_size 1 without an
    initialised process group."""
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    tile, layout = render_tiles(scene, spp, seed, rank, world, band_rows, render_band, device)
    W, H = scene.sensors()[0].film().size()
    return gather_film(tile, layout, W, H, rank, world, group)


def distributed_acquire(scene, paths_per_ray, seed=0, acquire=None, device=None, group=None, apply_pulse=None,
                        on_call=None, host_collective=False, force_collective=False, timing=None):
    """Ultrasound: every rank traces its path range into its own (already normalised) channel buffer;
    one reduce(sum) to rank 0.  acquire(offset, count, norm, out_tensor) fills the tensor; the default
    calls the HIP library on the tensor's device memory.  With pulse_model 'gaussian' rank 0 convolves the reduced
    buffer with the pulse (apply_pulse: None = follow the integrator, False = never, True / callable = do it).
    timing: a dict that receives this rank's work_s (acquisition) and collective_s (the reduce, incl. the wait for the others).
    -> [n_angles, n_elements, T] tensor on rank 0."""
    import time

    import torch

    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    t_a = time.perf_counter()
    ui = scene.integrator()
    off, cnt = path_ranges(paths_per_ray, world)[rank]
    dev = device if device is not None else torch.device("cpu")
    buf = torch.zeros((ui.n_angles, ui.n_elements, ui.time_samples), dtype=torch.float32, device=dev)
    if cnt > 0:
        if acquire is not None:
            acquire(off, cnt, paths_per_ray, buf)
        else:
            if dev.type != "cuda":
                raise RuntimeError("the HIP acquisition path needs a device tensor")
            ui._acquire(scene, ui.quirks, paths_per_ray=cnt, path_offset=off, norm_paths=paths_per_ray, seed=seed,
                        out_dev=buf.data_ptr(), pulse=False)
            if on_call is not None:
                on_call()
    if buf.is_cuda:
        torch.cuda.synchronize()
    t_b = time.perf_counter()
    if world > 1 or (force_collective and dist.is_initialized()):
        if host_collective:  # gloo rehearsal on one GPU: the collective runs on host tensors
            buf = buf.cpu()
        dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM, group=group)
        if buf.is_cuda:
            torch.cuda.synchronize()
    if timing is not None:
        timing["work_s"] = t_b - t_a
        timing["collective_s"] = time.perf_counter() - t_b
    if rank != 0:
        return None
    if apply_pulse is None:
        # what the kernel was actually told: echoes deposited without the carrier (PBRT_USQ_NO_CARRIER) still want their pulse.
        # (pulse_model sets that bit; a caller who edits `quirks` afterwards must not get a doubly / never convolved buffer.)
        from . import _capi
        apply_pulse = bool(int(ui.us_params(scene).quirks) & _capi.USQ_NO_CARRIER)
    if apply_pulse:
        # pulse_model 'gaussian': the shards hold bare echo amplitudes; the pulse is linear, so ONE convolution of the
        # reduced buffer equals the single-GPU result (UltraIntegrator._acquire applies it to its host buffer).
        # apply_pulse may be a callable (traces, fs, frequency, sigma) -> traces (CPU tests pass the restatement).
        if callable(apply_pulse):
            _pulse = apply_pulse
        else:
            from .beamform import apply_pulse as _pulse
        host = _pulse(buf.detach().cpu().numpy(), ui.fs, ui.frequency, ui.pulse_sigma)
        buf = torch.from_numpy(np.ascontiguousarray(host, dtype=np.float32)).to(buf.device)
    return buf


# ---- radiance mode, sharded by SAMPLES -------------------------------------------------------------------------------
# The other natural split of a film of fixed size: every rank renders the WHOLE film with its own contiguous range of
# the global sample indices (same RNG keys as the single-GPU render of all the samples) into un-normalised accumulators
# (sum w*rgb, sum w; PBRT_FILM_RAW_ACCUM), ONE reduce(sum) of H*W*16 bytes adds them on rank 0, which divides.  Per-rank
# work is exactly the single-GPU workload whatever the world size (no halo rows, no small crops, perfect balance), which
# is what a weak-scaling run over spp wants; the sum of the partial accumulators differs from the sequential
# accumulation only in the order of a few float additions (<= 1e-6 relative), so the image is deterministic for a given
# world size but not bit-identical across world sizes -- the band split above is.
def sample_ranges(spp: int, world_size: int):
    """-> [(first_sample, count)] contiguous split of the sample index"""
    return path_ranges(spp, world_size)


def render_sample_shard(scene, spp, seed, rank, world_size, render_raw=None, device=None):
    """This rank's samples of the whole film -> raw accumulators [H, W, 4] (float32 torch tensor).
    render_raw(first_sample, count, out_tensor) fills the tensor; the default calls the HIP library on the tensor's
    device memory."""
    import torch

    sens = scene.sensors()[0]
    W, H = sens.film().size()
    dev = device if device is not None else torch.device("cpu")
    raw = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    first, count = sample_ranges(spp, world_size)[rank]
    if count > 0:
        if render_raw is not None:
            render_raw(first, count, raw)
        else:
            if dev.type != "cuda":
                raise RuntimeError("the HIP render path needs a device tensor (torch 'cuda' == HIP on ROCm)")
            scene.integrator().render(scene, sensor=sens, seed=seed, spp=count, sample_offset=first, raw=True,
                                      out_dev=raw.data_ptr())
    return raw


def reduce_film(raw, rank, world_size, group=None):
    """ONE reduce(sum) of the raw accumulators to rank 0, then rgb / w.  -> [H, W, 3] tensor on rank 0, None elsewhere."""
    import torch

    if world_size > 1:
        _dist().reduce(raw, dst=0, op=_dist().ReduceOp.SUM, group=group)
        if rank != 0:
            return None
    w = raw[..., 3:4]
    inv = torch.where(w > 0, 1.0 / w, torch.zeros_like(w))   # k_film_resolve: rgb * (w > 0 ? 1 / w : 0)
    return raw[..., :3] * inv


def distributed_render_samples(scene, spp, seed=0, render_raw=None, device=None, group=None):
    """Whole job, sample-sharded: render this rank's samples, reduce on rank 0."""
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    raw = render_sample_shard(scene, spp, seed, rank, world, render_raw, device)
    return reduce_film(raw, rank, world, group)
