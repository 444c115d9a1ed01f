"""Longer run of the chain soak (tests/test_gpu_chain_soak.py): 3000 updates per size, alone and under load, every repeat of a
noise level bit-identical to its first occurrence (logpdf; the whole factor near the end)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_chain_soak as T
from boss_jl_amd import api
api.load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for N in (1408, 4096):
    for load in (False, True):
        t = time.perf_counter(); calls = T.soak(api, N, n, with_load=load); dt = time.perf_counter() - t
        print(f"N={N} {n} updates {'beside ' + str(calls) + ' acquisition calls of a second thread' if load else 'alone'}: bit-identical ({dt:.1f} s)", flush=True)
