"""Importable pieces of bench.py (the driver's entry point stays `bench.py` at the repo root: argument parsing, the timed region and
the JSON line).  inputs: the synthetic workload; kernels: live per-kernel timing + the `roofline` object; cpu_baseline: the fp32 oracle
on the host cores (the ONLY module here that imports `oracle/`); vae_clip: VAE timing and the end-to-end clip; launch: starting the N
ranks and the N-rank self-check; probe: the layout probe of an N >= 4 run; emulate: one rank's share of an N-rank step on one GPU."""
