"""CPU, world_size 2, gloo: the multi-rank path of bench.py (segments dealt to ranks, barrier on both
sides, MAX over ranks, rank 0 prints one JSON line)."""
import json
import os
import subprocess
import sys

from conftest import ROOT, load_pkg


def test_shard_segments_partition():
    load_pkg()
    from isee3_decoder_amd import harness
    for world in (1, 2, 4, 8):
        for nseg in (1, 8, 64, 13):
            parts = [harness.shard_segments(nseg, world, r) for r in range(world)]
            flat = sorted(g for p in parts for g in p)
            assert flat == list(range(nseg))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_two_rank_gloo_harness():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29531",
                        os.path.join(ROOT, "tests", "_dist_worker.py"), "6"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(line) == 1, "exactly one JSON line, from rank 0"
    r = json.loads(line[0])
    assert r["world"] == 2 and r["segments"] == [[0, 2, 4], [1, 3, 5]]
    # rank 1 sleeps twice as long per segment: the reported time is the slow rank's
    assert r["dt_max"] >= 0.11 and r["dt_max"] >= r["dt_rank0"]
