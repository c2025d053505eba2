"""The halves of bench.py (the driver's entry point stays bench.py at the repo root): common -- inputs and constants;
single -- N = 1; pipeline -- N >= 2; cpu -- the CPU baselines."""
