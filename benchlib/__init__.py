"""Pieces of bench.py (the repository-root bench keeps the CLI and the JSON line): cpu (cpu_baseline), timers (SpMM roofline,
PMC table, whole-step HBM fraction), workloads (the measured steps), model (the multi-GPU exchange model)."""
