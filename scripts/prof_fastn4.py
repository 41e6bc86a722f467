"""Bounded experiments with the run-time-radix kernels' knobs (csrc/fastn.h):  (a) C5's column pass with 64-byte row segments (four float64 sequences per workgroup,
verdict r4 item 6), (b) the chirp-convolution columns of the ERA5 grid at other widths / thread counts, (c) lengths with factors 7 / 11 / 13 on the one-pass kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_fastn import one
import xrft_amd as xrft
print("== (a) (64, 1440, 720) float64: table kernels, then run-time radices with 2 / 4 sequences per column workgroup")
one(64, 1440, 720, "float64", profile=True)
for gc, tc in ((2, 192), (2, 256), (4, 192), (4, 256), (4, 384), (4, 512)):
    one(64, 1440, 720, "float64", env={"XRFTHIP_FASTN_TABLES": 0, "XRFTHIP_FASTN_GC": gc, "XRFTHIP_FASTN_TC": tc}, profile=True)
print("== (b) (64, 721, 1440) float32")
for gc, tc in ((2, 256), (4, 256), (4, 512), (8, 512), (8, 1024)):
    one(64, 721, 1440, "float32", env={"XRFTHIP_FASTN_GC": gc, "XRFTHIP_FASTN_TC": tc}, profile=True)
