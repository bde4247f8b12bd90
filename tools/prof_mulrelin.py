"""Runs bench.py's mul+relin loop alone (for rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import numpy as np, torch
import bench
from cuhe_amd import capi
args = argparse.Namespace(relin_batch=8, relin_threads=4)
r = bench.bench_mulrelin(capi.lib, capi.check, torch, np, torch.device("cuda", 0), args)
print(r)
