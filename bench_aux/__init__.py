"""Auxiliary legs of bench.py (everything except the timed N = 1 path): counter passes, the clock / power sampler, the ciphertext
chains, PRINCE, the sharded multiply, CPU baselines, the doc/Perf_NTT.txt table.  bench.py imports them; nothing here is on the
product path."""
