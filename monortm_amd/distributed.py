"""Profile sharding across the GPUs of one node.

The only parallel axis of the reference is its independent-profile loop (reference
src/monortm.f90:357): profiles are dealt in contiguous blocks of ceil(P/G) to G ranks (one process per
GPU), every rank holds the whole line table, and the per-profile spectral outputs are collected with ONE
gather to rank 0 (RCCL over xGMI when the backend is "nccl"; the same code runs on "gloo" for the CPU
tests).  A single profile is never split across GPUs.
"""
from __future__ import annotations

import math

import torch
import torch.distributed as dist


def shard_bounds(nprof: int, world: int) -> list[tuple[int, int]]:
    """[start, stop) of every rank's contiguous block; the last ranks may be short or empty."""
    per = math.ceil(nprof / world) if world > 0 else nprof
    return [(min(r * per, nprof), min((r + 1) * per, nprof)) for r in range(world)]


def gather_to_root(local: torch.Tensor, nprof: int, group=None) -> torch.Tensor | None:
    """local: [n_local, ...] rows of this rank's block -> [nprof, ...] on rank 0 (None elsewhere).
    One collective: blocks are padded to the common size ceil(P/G) so that a plain gather suffices."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = math.ceil(nprof / world)
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0], *local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
    dist.gather(local, bufs, dst=0, group=group)
    if rank != 0:
        return None
    return torch.cat(bufs, dim=0)[:nprof]


class GatherPlan:
    """The per-step gather of a sharded job with everything allocated once: rank 0 owns one [world * per, ...] buffer whose
    contiguous blocks are the receive slots, so no concatenation follows the collective, and the collective is issued
    asynchronously - the next step's kernels run while the previous step's outputs travel.  The tensor handed to start() is
    kept alive until the collective has completed; result() is valid on rank 0 after wait()."""

    def __init__(self, nprof: int, like: torch.Tensor, group=None, field_major: bool = False):
        """field_major: the rank's rows arrive as ONE block [F, n_local, ...] (api.DeviceBatch.spectral_block(): the kernels'
        own output block, handed over without a copy) instead of [n_local, F, ...]; rank 0 then holds [world, F, per, ...] and
        result() returns the [nprof, F, ...] view of it.  The caller must not overwrite a block while its gather is in flight
        (DeviceBatch.pingpong: consecutive steps write alternate blocks; start() waits for the gather before the last one)."""
        self.group, self.nprof, self.field_major = group, nprof, field_major
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.per = math.ceil(nprof / max(self.world, 1))
        self.trail = tuple(like.shape[1:]) if not field_major else (like.shape[0], *like.shape[2:])
        self.out = None
        if self.world > 1 and self.rank == 0:
            shape = (self.world * self.per, *self.trail) if not field_major else (self.world, like.shape[0], self.per, *like.shape[2:])
            self.out = torch.empty(shape, dtype=like.dtype, device=like.device)
        self.slots = list(self.out.chunk(self.world, dim=0)) if self.out is not None else None
        if field_major and self.slots is not None:
            self.slots = [x[0] for x in self.slots]   # [F, per, ...] each
        self.pad = None
        self.work = None
        self.inflight = None
        self.local = None

    def start(self, local: torch.Tensor):
        self.wait()
        if self.world == 1:
            self.local = local
            return
        if self.field_major:
            if local.shape[1] < self.per:   # a short last block: padded copy (the equal blocks of the benchmark never come here)
                if self.pad is None:
                    self.pad = torch.zeros((local.shape[0], self.per, *local.shape[2:]), dtype=local.dtype, device=local.device)
                self.pad[:, : local.shape[1]] = local
                local = self.pad
        elif local.shape[0] < self.per:
            if self.pad is None or self.pad.shape[0] != self.per:
                self.pad = torch.zeros((self.per, *self.trail), dtype=local.dtype, device=local.device)
            self.pad[: local.shape[0]] = local
            local = self.pad
        self.inflight = local.contiguous()   # (no copy for a contiguous block)
        self.work = dist.gather(self.inflight, self.slots, dst=0, group=self.group, async_op=True)

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
            self.inflight = None

    def result(self):
        self.wait()
        if self.world == 1:
            return self.local.permute(1, 0, *range(2, self.local.dim())) if self.field_major else self.local
        if self.rank != 0:
            return None
        if self.field_major:   # [world, F, per, ...] -> [world * per, F, ...]
            o = self.out.permute(0, 2, 1, *range(3, self.out.dim()))
            return o.reshape(self.world * self.per, *o.shape[2:])[: self.nprof]
        return self.out[: self.nprof]


def run_sharded(profiles, compute, group=None):
    """Run `compute(list_of_profiles) -> tensor [n, ...]` on this rank's block of `profiles` and gather.
    `compute` is the HIP path in production (DeviceBatch.step + spectral_outputs): its tensors live on the rank's GPU in the
    context's real kind, and the collective runs on exactly those (RCCL takes device tensors only; gloo takes either)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(len(profiles), world)[rank]
    mine = profiles[lo:hi]
    local = compute(mine) if mine else None
    if dist.is_initialized() and world > 1:
        # an empty block still has to take part in the collective with the trailing shape, dtype and device kind of the
        # others: exchange the descriptions as objects (no tensor of a fixed device / dtype is involved)
        desc = None if local is None else (tuple(local.shape[1:]), str(local.dtype).replace("torch.", ""), local.device.type)
        descs = [None] * world
        dist.all_gather_object(descs, desc, group=group)
        ref = next((d for d in descs if d is not None), None)
        if ref is None:
            return None
        if local is None:
            dev = torch.device("cuda", torch.cuda.current_device()) if ref[2] == "cuda" else torch.device("cpu")
            local = torch.zeros((0, *ref[0]), dtype=getattr(torch, ref[1]), device=dev)
    return gather_to_root(local, len(profiles), group)
