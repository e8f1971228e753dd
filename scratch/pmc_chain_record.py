"""profiles/r03_pmc_chain.json from the per-kernel PMC sums of scratch/r03_pmc_chain.sh (one entry per sample rate):
HBM bytes per IQ sample and stage = sum over the stage's kernels of (FETCH_SIZE x 2 + WRITE_SIZE) KiB / samples processed
in the profiled runs (= dispatches of the first FFT pass x N).
usage: pmc_chain_record.py out.json rate:bin:logN:per_kernel.json [...]"""
import json, sys
STAGES = {"viterbi": ("k_acs", "k_decodebit", "k_l15", "k_init", "k_snapshot", "k_count_diff", "k_argmin", "k_export", "k_chainback", "k_max"),
          "pmdemod": ("k_fft_pass", "k_mix", "k_rotate", "k_sum2", "k_peak", "k_dft", "k_carrier_steps", "k_twiddles", "k_iq"),
          "symdemod": ("k_scan", "k_timesearch", "k_ts_argmax", "k_window_demod", "k_seq_energy", "k_par_energy", "k_demod", "k_slide"),
          "copies_and_fills": ("__amd_rocclr",)}
out = {"what": "physical HBM traffic of the in-process chain by stage, bytes per IQ sample: rocprofv3 --kernel-trace --pmc FETCH_SIZE and "
               "--pmc WRITE_SIZE in two separate runs of `bench.py --workload chain --chain-rate R --chain-seconds S --steps 1 --warmup 0 "
               "--no-cpu` (scratch/r03_pmc_chain.sh), summed by kernel (profiles/r03_pmc_chain_<rate>_by_kernel.json); FETCH_SIZE x 2 "
               "(gfx950 reports half the bytes of wide streaming reads, MI355X_MICROARCH.md HBM section), KiB -> bytes",
       "configs": []}
for arg in sys.argv[2:]:
    rate, binsize, logn, path = arg.split(":")
    per = json.load(open(path))
    first = [k for k in per if k.startswith("k_fft_pass") and ", true, false" in k and ", 1, " in k]
    if len(first) > 1:                      # the search transform's first pass (a fall-back adds the double one's dispatches)
        first = [k for k in first if "float" in k]
    assert len(first) == 1, first
    blocks = per[first[0]]["dispatches"]
    nsamp = blocks * (1 << int(logn))
    bps, unk = {}, []
    for k, v in per.items():
        b = (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0
        for st, pre in STAGES.items():
            if k.startswith(pre):
                bps[st] = bps.get(st, 0.0) + b / nsamp
                break
        else:
            unk.append(k)
    assert not unk, unk
    out["configs"].append({"samprate": float(rate), "binsize": float(binsize), "fft_log2": int(logn), "blocks_profiled": blocks,
                           "samples_profiled": nsamp, "hbm_bytes_per_sample": {k: round(v, 3) for k, v in sorted(bps.items())},
                           "source": "profiles/r03_pmc_chain.json, %g S/s" % float(rate)})
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out["configs"], indent=1))
