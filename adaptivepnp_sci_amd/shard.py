"""Multi-GPU sharding of independent reconstruction units (measurements / colour cubes / spatial tiles).

The reference is single-GPU and loops over measurements sequentially
(two_stage_ADMM_Online_FFD_Warm.py:241 `for iframe in range(nmea)`); units are independent SCI problems as
long as every unit starts from the same denoiser weights (`reuse_model=False`, :272-275), so they shard
with NO communication inside the solve: unit i runs on rank i % world (one process per GPU,
torch.distributed over RCCL/xGMI), and the (H,W,B) mosaics -- optionally the (H,W,3,B) colour cubes --
are collected with ONE gather at the end.  The same code runs on the `gloo` backend with CPU tensors for the
collective plumbing (tests/test_shard.py); the solve itself has no CPU path.
"""
import copy

import torch
import torch.distributed as dist


def partition(n_units, world, rank):
    """Round-robin unit -> rank map (unit i on rank i % world); returns this rank's unit indices."""
    if not (0 <= rank < world):
        raise ValueError(f'rank {rank} outside world {world}')
    return list(range(rank, n_units, world))


def slots_per_rank(n_units, world):
    return (n_units + world - 1) // world


def gather_units(local, n_units, unit_shape, device, dtype=torch.float32, dst=0, group=None):
    """ONE collective: every rank contributes a (slots, *unit_shape) buffer holding its units in
    partition order (zero padding when n_units is not a multiple of world); rank `dst` returns the
    list of all n_units tensors in unit order, the others return None.

    local: dict {unit index: tensor of unit_shape} for exactly partition(n_units, world, rank)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = partition(n_units, world, rank)
    if sorted(local) != mine:
        raise ValueError(f'rank {rank} must provide units {mine}, got {sorted(local)}')
    slots = slots_per_rank(n_units, world)
    buf = torch.zeros((slots,) + tuple(unit_shape), dtype=dtype, device=device)
    for s, u in enumerate(mine):
        buf[s].copy_(local[u])
    if world == 1 and not dist.is_initialized():
        return [buf[s] for s in range(len(mine))]
    recv = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return [recv[u % world][u // world] for u in range(n_units)]


def reconstruct_sharded(units, solve, unit_shape, device, model=None, dst=0, group=None, streams=1):
    """units: list of per-unit argument tuples (same on every rank); solve(unit_args, model_copy) -> tensor of
    unit_shape on `device`.  Every unit gets its own deep copy of `model` so that the online finetune of one
    unit cannot leak into another (parity-exact sharding).

    streams > 1: this rank's units are solved by that many host threads, each on its own HIP stream.  Results do not
    depend on it (units are independent, kernels deterministic) -- except when `solve` draws from a process-global RNG,
    as the FastDVDnet finetune does with NumPy's (keep streams=1 there so the draws stay in unit order).  It is an option,
    not the default: with the host side free of stalls (see _lib.cap_host_threads) one stream already keeps the GPU busy
    -- four 256 x 256 x 16 tiles with the online finetune take 191-237 ms on one stream and 208-271 ms on two."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = partition(len(units), world, rank)

    def one(u):
        return solve(units[u], copy.deepcopy(model) if model is not None else None)

    local = {}
    if streams <= 1 or len(mine) <= 1 or not torch.cuda.is_available():
        for u in mine:
            local[u] = one(u)
    else:
        import concurrent.futures as cf
        cur = torch.cuda.current_stream()
        pool_streams = [torch.cuda.Stream() for _ in range(min(streams, len(mine)))]
        for st in pool_streams:
            st.wait_stream(cur)

        def worker(j):
            out = {}
            with torch.cuda.stream(pool_streams[j]):
                for u in mine[j::len(pool_streams)]:
                    out[u] = one(u)
            return out

        with cf.ThreadPoolExecutor(len(pool_streams)) as ex:
            for part in ex.map(worker, range(len(pool_streams))):
                local.update(part)
        for st in pool_streams:
            cur.wait_stream(st)
    return gather_units(local, len(units), unit_shape, device, dst=dst, group=group)


# ---------------------------------------------------------------------------------------------- spatial tiles
def tile_grid(H, W, tile):
    """Origins (r, c) of the Bayer-aligned `tile` x `tile` patches covering an (H, W) frame, row-major
    (BASELINE configs[4]: a 1024 x 1024 x 16 cube as 16 patches of 256 x 256)."""
    if tile % 2 or H % tile or W % tile:
        raise ValueError(f'tile {tile} must be even and divide the frame {H}x{W} (patches keep the RGGB phase)')
    return [(r, c) for r in range(0, H, tile) for c in range(0, W, tile)]


def tile_cube(y, Phi, tile, x0=None, orig=None):
    """Cut measurement (H,W), masks (H,W,B) and optional warm start / ground truth (H,W,B) into per-tile argument
    tuples (y_t, Phi_t, x0_t, orig_t); the forward operator is per-pixel, so every patch is an independent SCI problem
    (SURVEY 8e).  Works on NumPy arrays and torch tensors alike."""
    H, W = y.shape[:2]
    units = []
    for r, c in tile_grid(H, W, tile):
        sl = (slice(r, r + tile), slice(c, c + tile))
        units.append((y[sl], Phi[sl], None if x0 is None else x0[sl], None if orig is None else orig[sl]))
    return units


def stitch_tiles(tiles, H, W, tile):
    """Inverse of tile_cube for a list of (tile, tile, ...) tensors in tile_grid order -> (H, W, ...)."""
    first = tiles[0]
    out = first.new_empty((H, W) + tuple(first.shape[2:]))
    for t, (r, c) in zip(tiles, tile_grid(H, W, tile)):
        out[r:r + tile, c:c + tile] = t
    return out


def reconstruct_tiled(y, Phi, tile, solve, device, x0=None, orig=None, model=None, dst=0, group=None, streams=1):
    """Tile a large cube, reconstruct the patches independently on the ranks of `group` (tile j on rank j % world, each
    with its own deep copy of `model`), gather them with ONE collective and stitch on rank `dst`.
    solve((y_t, Phi_t, x0_t, orig_t), model_copy) -> (tile, tile, B) tensor on `device`.
    `streams`: as reconstruct_sharded (one stream keeps the GPU busy: ~48 ms per 256x256x16 tile with the online finetune).
    Returns the (H, W, B) mosaic on rank dst, None elsewhere."""
    H, W, B = Phi.shape
    units = tile_cube(y, Phi, tile, x0, orig)
    got = reconstruct_sharded(units, solve, (tile, tile, B), device, model=model, dst=dst, group=group, streams=streams)
    return None if got is None else stitch_tiles(got, H, W, tile)


# ---------------------------------------------------------------------------------------------- fixed-total timed jobs
def timed_job(n_units, prepare, iterate, finish, unit_shape, device, steps, dst=0, group=None, sync=None, batched=False):
    """A job of a FIXED total of `n_units` independent units (BASELINE configs[3]: 8 cubes; configs[4]: the 16 tiles of a
    1024 x 1024 x 16 cube) over the ranks of `group`, timed the way bench.py's contract asks:

        state_u = prepare(u)                for this rank's units (unit u on rank u % world)       -- untimed
        barrier; t0
        iterate(state_u, k), k < steps      for every unit of the rank, unit after unit              -- "solve"
        finish(state_u) -> unit_shape       ONE gather of all units to rank `dst`                    -- "gather"
        barrier; t1

    `sync()` (torch.cuda.synchronize on a GPU rank, None on CPU) is called where a time is taken.  Returns
    (units, timing): the list of all n_units tensors in unit order on rank dst (None elsewhere) and, on every rank,
    timing = {'total_s': max over ranks of t1 - t0, 'solve_s': [per rank], 'gather_s': [per rank], 'units': [per rank]}
    -- the per-rank figures travel in one small all_gather AFTER the timed region (metrics, not the data path).

    batched=True (round 4, unit batches): the rank's units are stepped TOGETHER -- prepare(list of the rank's units) -> ONE
    state, iterate(state, k) advances every unit of the rank by one iteration in one launch sequence
    (solver.AdmmRun(units=...)), finish(state) -> {unit: tensor}.  Same timed region, same single gather."""
    import time
    inited = dist.is_initialized()
    world = dist.get_world_size(group) if inited else 1
    rank = dist.get_rank(group) if inited else 0
    sync = sync or (lambda: None)
    mine = partition(n_units, world, rank)
    states = (prepare(mine) if mine else None) if batched else {u: prepare(u) for u in mine}

    def barrier():
        sync()
        if inited:
            dist.barrier(group)
        sync()

    import gc
    gc.collect()                    # (no generation-2 collector pass inside the timed region: a host pause of tens of ms idles the GPU)
    gc_was = gc.isenabled()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    if batched:
        for k in range(steps if mine else 0):
            iterate(states, k)
        local = finish(states) if mine else {}
    else:
        for u in mine:
            for k in range(steps):
                iterate(states[u], k)
        local = {u: finish(states[u]) for u in mine}
    sync()
    t_solve = time.perf_counter() - t0
    got = gather_units(local, n_units, unit_shape, device, dst=dst, group=group)
    sync()
    t_gather = time.perf_counter() - t0 - t_solve
    barrier()
    total = time.perf_counter() - t0
    if gc_was:
        gc.enable()
    mine_t = torch.tensor([total, t_solve, t_gather, float(len(mine))], dtype=torch.float64, device=device)
    if inited:
        allt = [torch.empty_like(mine_t) for _ in range(world)]
        dist.all_gather(allt, mine_t, group=group)
        allt = torch.stack(allt).cpu()
    else:
        allt = mine_t.cpu().unsqueeze(0)
    timing = {'total_s': float(allt[:, 0].max()), 'solve_s': [float(v) for v in allt[:, 1]],
              'gather_s': [float(v) for v in allt[:, 2]], 'units': [int(v) for v in allt[:, 3]]}
    return got, timing
