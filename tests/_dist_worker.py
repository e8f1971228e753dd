"""world_size-2 gloo worker for tests/test_dist_harness.py: exercises the exact sharding / fence /
MAX-over-ranks code bench.py uses, with a CPU stand-in for the per-segment work (no GPU here)."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg  # noqa: E402

load_pkg()
from isee3_decoder_amd import harness  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
nseg = int(sys.argv[1])
mine = harness.shard_segments(nseg, world, rank)
done = []


def step():
    for g in mine:
        time.sleep(0.01 * (1 + rank))        # rank 1 is slower: MAX must pick it up
        done.append(g)


fence = harness.make_fence(dist, lambda: None)
dt = harness.timed_steps(step, steps=2, warmup=1, fence=fence)
dt_max = harness.max_over_ranks(dist, torch, dt, "cpu")
allseg = [None] * world
dist.all_gather_object(allseg, mine)
if rank == 0:
    print(json.dumps({"dt_max": dt_max, "dt_rank0": dt, "segments": allseg, "world": world,
                      "units": nseg * 2}), flush=True)
dist.destroy_process_group()
