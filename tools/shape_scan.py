"""Efficiency scan over row counts / codebook sizes / dims (dev tool): fraction of the fp32-MFMA peak per shape, to find
launch-heuristic cliffs.  python tools/shape_scan.py"""
import sys, os
sys.path[:0] = [os.path.dirname(os.path.abspath(__file__))]
import quick_bench as qb
for D in (64, 256, 512):
    for K in (256, 1024, 8192):
        for M in (1000, 4096, 16384, 50000, 90000, 131073, 200000):
            qb.bench(M, K, D, iters=10)
