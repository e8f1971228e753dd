"""Timing / sharding harness shared by bench.py and the CPU (gloo) tests.

The data path has no collective: independent capture segments are dealt to ranks, every rank works
through its own list, and the only communication is the barrier on both sides of the timed region plus
one MAX all-reduce of the elapsed time (bench contract)."""
import time


def shard_segments(nsegments, world, rank):
    """Segment g goes to rank g mod world.  Returns this rank's segment ids."""
    return [g for g in range(nsegments) if g % world == rank]


def timed_steps(step, steps, warmup, fence):
    """warmup untimed calls, then exactly `steps` timed calls bracketed by fence() on both sides."""
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    return time.perf_counter() - t0


def make_fence(dist, device_sync):
    """barrier (if distributed) + device synchronise."""
    def fence():
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            dist.barrier()
        device_sync()
    return fence


def max_over_ranks(dist, torch, value, device):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def ranks_seen(dist, torch, device):
    """SUM over ranks of 1: how many ranks really took part (goes into the bench line)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return 1
    t = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def gather_per_rank(dist, value):
    """every rank's `value` (a small picklable object), on every rank, in rank order."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [value]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, value)
    return out


def run_segments(plan, mine, run_one, concurrency=2):
    """Decode this rank's segments of a plan (segment.plan_segments), `concurrency` at a time (threads: the chains of
    one process overlap on the device).  run_one(first_block, end_block) -> decoded bits.  Returns {segment: bits}."""
    import threading
    todo, parts, lock, errors = list(mine), {}, threading.Lock(), []

    def worker():
        while True:
            with lock:
                if not todo or errors:
                    return
                g = todo.pop(0)
            b0, _, b1 = plan[g]
            try:
                parts[g] = run_one(b0, b1)
            except Exception as e:                      # noqa: BLE001
                errors.append(e)
                return

    ts = [threading.Thread(target=worker) for _ in range(max(1, concurrency))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errors:
        raise errors[0]
    return parts


def gather_and_stitch(dist, plan, parts, bits_per_block, stitch):
    """all_gather the per-rank {segment: bits} dictionaries (the ONLY exchange of the segmented mode: decoded bits, a few
    kB) and stitch them on rank 0.  Returns (bits, {"matched", "total"}) on rank 0, (b"", None) elsewhere."""
    allparts = gather_per_rank(dist, parts)
    rank = dist.get_rank() if (dist is not None and dist.is_initialized()) else 0
    if rank != 0:
        return b"", None
    merged = {}
    for d in allparts:
        merged.update(d)
    missing = [g for g in range(len(plan)) if g not in merged]
    if missing:
        raise RuntimeError("segments %s were decoded by no rank" % missing)
    ovl = [int((plan[i + 1][1] - plan[i + 1][0]) * bits_per_block) for i in range(len(plan) - 1)]
    bits, ok, tot = stitch([merged[g] for g in range(len(plan))], ovl)
    return bits, {"matched": ok, "total": tot}
