"""CPU, world_size 2, gloo: the multi-rank path of bench.py (segments dealt to ranks, barrier on both
sides, MAX over ranks, rank 0 prints one JSON line)."""
import json
import os
import subprocess
import sys

import numpy as np

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


def test_stitch_locates_the_seam_by_position_and_verifies_it():
    """segment.stitch: the probe is only looked for where the overlap can be, must be unique there, and the two parts
    must agree over the rest of the overlap; repetitive telemetry one frame off, or a corrupted overlap, is NOT a match."""
    load_pkg()
    from isee3_decoder_amd import segment
    rng = np.random.default_rng(5)
    full = bytes(rng.integers(48, 50, 24000, dtype=np.uint8))
    ovl, cut = 3500, 9000
    junk = bytes(rng.integers(48, 50, 2100, dtype=np.uint8))          # unsettled start of a restarted decode
    for tail in (200, 450, 720):
        p0 = full[:cut - tail]
        p1 = junk + full[cut - ovl + 2100:18000]
        out, ok, tot = segment.stitch([p0, p1], [ovl])
        assert (ok, tot) == (1, 1) and out == full[:18000]
    # three parts
    p0, p1, p2 = full[:8700], junk + full[9000 - ovl + 2100:15600], junk + full[16000 - ovl + 2100:]
    out, ok, tot = segment.stitch([p0, p1, p2], [ovl, ovl])
    assert (ok, tot) == (2, 2) and out == full
    # the same bits elsewhere in the previous part (the old whole-string rfind would have taken them): ignored
    decoy = full[9000 - ovl + 3000:9000 - ovl + 3400]
    p0d = full[:2000] + decoy + full[2400:8700]
    out, ok, tot = segment.stitch([p0d, p1], [ovl])
    assert ok == 1 and out[8700:] == full[8700:15600]
    # a corrupted overlap: the probe may match somewhere, the verification does not -> reported as unmatched
    bad = bytearray(p1)
    for i in range(2300, 3400, 7):
        bad[i] ^= 1
    out, ok, tot = segment.stitch([p0, bytes(bad)], [ovl])
    assert ok == 0 and len(out) == len(p0) + len(bad)
    # periodic frames: alignment is decided by position, never a frame off
    frame = bytes(rng.integers(48, 50, 1024, dtype=np.uint8))
    per = frame * 24
    out, ok, tot = segment.stitch([per[:9000 - 300], per[9000 - ovl:20000]], [ovl])
    assert ok == 1 and out == per[:20000]
    # ... at EVERY tail the stages can leave (vdecode's delay .. delay + one symdemod window): the window of possible
    # positions is narrower than a frame, so a frame-periodic stream never matches twice
    for tail in range(200, 713, 32):
        out, ok, tot = segment.stitch([per[:9000 - tail], per[9000 - ovl:20000]], [ovl])
        assert ok == 1 and out == per[:20000], tail
    # a probe that occurs twice inside the window but whose overlap verifies only once: placed, not rejected as ambiguous
    twice = bytearray(full)
    twice[9000 - ovl + 3200 + 300:9000 - ovl + 3200 + 300 + 160] = full[9000 - ovl + 3200:9000 - ovl + 3360]
    twice = bytes(twice)
    out, ok, tot = segment.stitch([twice[:9000 - 400], junk + twice[9000 - ovl + 2100:18000]], [ovl])
    assert ok == 1 and out == twice[:18000]


def test_two_rank_gloo_segmented_chain_path():
    """bench.py --workload chain --chain-segments S on two ranks, without a GPU: the same plan / shard / run_segments /
    all_gather_object / stitch code with a stand-in for run_chain.  Every seam must verify and the stitched stream must
    be the one stream the segments were cut from; the line carries ranks_seen and the per-rank times."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533",
                        os.path.join(ROOT, "tests", "_dist_worker_chain.py"), "16", "64"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(line) == 1
    r = json.loads(line[0])
    assert r["world"] == 2 and r["ranks_seen"] == 2 and len(r["per_rank_ms"]) == 2
    # 16 segments of 4 blocks, warm-up 7 blocks: the first two start at block 0 and are merged
    assert r["segments"] == 15 and r["seams"] == {"matched": 14, "total": 14}
    assert r["equal_to_truth"] and r["nbits"] > 30000
    assert r["mine"] == list(range(0, 15, 2)) and r["calls_rank0"] == 8


def _bench_lines(cmd, env=None, timeout=300):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout,
                       env=dict(os.environ, OMP_NUM_THREADS="1", **(env or {})))
    return p, [l for l in p.stdout.decode().splitlines() if l.startswith("{")]


def test_bench_gpus_n_starts_n_ranks_by_itself():
    """`python bench.py --gpus 2` (the form the driver types for N = 1) must run TWO ranks: bench.py starts
    torch.distributed.run as a child before anything touches a device.  --dry-ranks swaps the decode for a sleep and
    RCCL for gloo; launcher, sharding, fences, MAX and the ranks_seen SUM are the real run's."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-ranks", "--steps", "2",
                        "--warmup", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(line) == 1, "exactly one JSON line, from rank 0"
    r = json.loads(line[0])
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and len(r["ms_per_step_per_rank"]) == 2
    assert r["config"]["segments_by_rank"] == [[0], [1]]
    # rank 1's stand-in step is twice as long: the reported time is the slowest rank's
    assert r["ms_per_step"] >= max(r["ms_per_step_per_rank"]) - 1e-3 and r["ms_per_step_per_rank"][1] >= 19.0


def test_bench_under_torchrun_is_not_started_twice_and_checks_world_size():
    """The driver's N > 1 form (torch.distributed.run ... bench.py --gpus N): RANK is set, so bench.py does not spawn
    again; a --gpus that disagrees with WORLD_SIZE is an error, not a silently smaller run."""
    p, line = _bench_lines([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", "29537", os.path.join(ROOT, "bench.py"),
                            "--gpus", "2", "--dry-ranks", "--steps", "1", "--warmup", "0"])
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    assert len(line) == 1 and json.loads(line[0])["ranks_seen"] == 2
    p, line = _bench_lines([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-ranks"],
                           env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and not line and b"WORLD_SIZE is 2" in p.stderr
    p, line = _bench_lines([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-ranks", "--steps", "1"],
                           env={})
    assert p.returncode == 0 and len(line) == 1 and json.loads(line[0])["n_gpus"] == 1


def test_bench_record_helpers_without_a_device():
    """bench.py's pure helpers: the chain roofline object built from the committed PMC record (bytes per IQ sample and
    stage x samples / step time against 8 TB/s, the SURVEY 8(d) algorithmic figure beside it) and the host description of
    the CPU baselines.  No device, no timing: the schema and the arithmetic."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_chain.json")))
    for cfg in pmc["configs"]:
        fs, nsamp, dt = cfg["samprate"], 15_000_000, 0.03
        r = bench.chain_roofline(fs, cfg["binsize"], nsamp, dt, [6.0, 8.0, 26.0])
        total = sum(cfg["hbm_bytes_per_sample"].values())
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
        assert abs(r["traffic"] - total * nsamp) <= 1 and abs(r["achieved"] - total * nsamp / dt / 1e9) < 0.1
        assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3
        assert abs(r["algorithmic_bytes_per_sample"] - (8 + 1024.545058 / fs * 17301504)) < 0.1
        assert set(r["stages"]) == {"pmdemod", "symdemod", "viterbi", "what"}
        assert r["stages"]["viterbi"]["hbm_bytes"] == int(cfg["hbm_bytes_per_sample"]["viterbi"] * nsamp)
    assert bench.chain_roofline(123456.0, 1.0, 1000, 1.0) is None            # no PMC record for that rate: no invented number
    h = bench.host_info()
    assert h["nproc"] >= 1 and h["nproc_allowed"] >= 1 and isinstance(h["cpu_model"], str) and isinstance(h["avx2"], bool)
    with bench.pinned_to_one_core() as pin:
        assert not pin.ok or os.sched_getaffinity(0) == {pin.core}
    assert len(os.sched_getaffinity(0)) == h["nproc_allowed"]


def test_bench_pmc_constants_match_the_committed_kernel_record():
    """The VALU-issue roofline of the headline is formed with constants from profiles/r03_pmc_lds15.json: the file must name the
    kernel the bench times, carry both numbers, and the bytes must be what its own FETCH / WRITE entries add up to
    (FETCH_SIZE x 2 on gfx950, KiB -> bytes)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pmc, src = bench.pmc_constants("k_acs_lds15")
    assert src == "profiles/r03_pmc_lds15.json"
    assert 1000 < pmc["valu_insts_per_wave"] < 1500 and 3.5 < pmc["valu_cycles_per_inst"] < 5.0
    want = int((2 * pmc["fetch_size_kb_raw"] + pmc["write_size_kb"]) * 1024)
    assert abs(pmc["hbm_bytes_per_launch"] - want) <= 1024
    # one read + one write of the 2^23 u16 metrics + 15 decision rows of 1 MiB is the least a 15-step launch can move
    assert pmc["hbm_bytes_per_launch"] >= 2 * (1 << 24) + 15 * (1 << 20)
    assert bench.pmc_constants("no_such_kernel") == ({}, None)
    # the segmented form's concurrency is an argument with the measured default
    import subprocess, sys
    h = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120).stdout
    assert "--segment-concurrency" in h
