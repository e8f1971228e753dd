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
