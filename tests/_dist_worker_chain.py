"""world_size-2 gloo worker for tests/test_dist_harness.py: bench.py's segmented-chain path (BASELINE configs[4]) --
plan_segments -> this rank's segments (two at a time) -> all_gather_object of the decoded parts -> stitch on rank 0 --
with a CPU stand-in for run_chain (no GPU here): every segment "decodes" its slice of one known bit stream, with an
unsettled start and a ragged end like a restarted vdecode."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg  # noqa: E402

load_pkg()
from isee3_decoder_amd import harness, segment  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
nseg, nblocks, warm = int(sys.argv[1]), int(sys.argv[2]), 7
BITS_PER_BLOCK = 512.27
truth = bytes(np.random.default_rng(11).integers(48, 50, int(nblocks * BITS_PER_BLOCK) + 8, dtype=np.uint8))
calls = []


def fake_run_chain(b0, b1):
    calls.append((b0, b1))
    lo, hi = int(round(b0 * BITS_PER_BLOCK)), int(round(b1 * BITS_PER_BLOCK)) - 200 - 37 * ((b0 + b1) % 9)
    bits = bytearray(truth[lo:hi])
    if b0 > 0:                                       # a restarted decode: ~2100 unreliable bits
        junk = np.random.default_rng(b0).integers(48, 50, 2100, dtype=np.uint8)
        bits[:2100] = bytes(junk)
    return bytes(bits)


plan = segment.plan_segments(nblocks, nseg, warm)
mine = harness.shard_segments(len(plan), world, rank)
out = {}


def step():
    out["parts"] = harness.run_segments(plan, mine, fake_run_chain, concurrency=2)


fence = harness.make_fence(dist, lambda: None)
dt = harness.timed_steps(step, steps=1, warmup=1, fence=fence)
dt_max = harness.max_over_ranks(dist, torch, dt, "cpu")
seen = harness.ranks_seen(dist, torch, "cpu")
per_rank = harness.gather_per_rank(dist, round(dt * 1e3, 3))
bits, seams = harness.gather_and_stitch(dist, plan, out["parts"], BITS_PER_BLOCK, segment.stitch)
if rank == 0:
    n = len(bits)
    print(json.dumps({"world": world, "ranks_seen": seen, "per_rank_ms": per_rank, "segments": len(plan), "seams": seams,
                      "equal_to_truth": bits == truth[:n], "nbits": n, "mine": mine, "calls_rank0": len(calls) // 2}), flush=True)
dist.destroy_process_group()
