"""bench — the parts of bench.py (repo root) that are not the timed schedule itself: constants, command line, runtime helpers, committed-profile lookup,
the two oracle legs (cpu_baseline, parity_sample), the multi-rank exchange probe, the live-stream sweep and the roofline objects.  bench.py stays the
entry point the driver runs and prints the one JSON line; split in round 5 without behaviour change (tests/test_gpu_bench_contract.py)."""
