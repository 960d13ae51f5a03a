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


def reconstruct_sharded(units, solve, unit_shape, device, model=None, dst=0, group=None):
    """units: list of per-unit argument tuples (same on every rank); solve(unit_args, model_copy) -> tensor of
    unit_shape on `device`.  Every unit gets its own deep copy of `model` so that the online finetune of one
    unit cannot leak into another (parity-exact sharding)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    local = {}
    for u in partition(len(units), world, rank):
        local[u] = solve(units[u], copy.deepcopy(model) if model is not None else None)
    return gather_units(local, len(units), unit_shape, device, dst=dst, group=group)
