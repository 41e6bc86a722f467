"""Which entries of fastm.h's table does the run-time-radix pipeline match within 10 %?  (square float32 / float64 slabs of the table lengths)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_fastn import one
F32 = [1536, 1600, 1800, 1920, 2000, 2160, 2400, 2560, 2880, 3000, 3072, 3600, 3840, 4320]
F64 = [1080, 1200, 1280, 1440, 1500, 1800, 1920, 2000, 2160]
for n in F32:
    nt = max(4, min(64, (64 * 2000 * 2000) // (n * n)))
    one(nt, n, n, "float32")
    one(nt, n, n, "float32", env={"XRFTHIP_FASTN_TABLES": 0})
for n in F64:
    nt = max(4, min(64, (32 * 2000 * 2000) // (n * n)))
    one(nt, n, n, "float64")
    one(nt, n, n, "float64", env={"XRFTHIP_FASTN_TABLES": 0})
