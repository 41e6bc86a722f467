"""Run-time radices against the table kernels on table shapes (XRFTHIP_FASTN_TABLES=0).  python scripts/prof_fastn3.py on the GPU box"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_fastn import one
for t in [(64, 1440, 720, "float64"), (64, 1440, 720, "float32"), (32, 1000, 1000, "float32"), (64, 2000, 2000, "float32"), (32, 2000, 2000, "float64")]:
    one(*t, profile=True)
    one(*t, env={"XRFTHIP_FASTN_TABLES": 0}, profile=True)
    for tc, tr in ((256, 256), (512, 256), (512, 512)) if t[3] == "float32" else ((192, 192), (256, 192), (384, 192), (384, 384)):
        one(*t, env={"XRFTHIP_FASTN_TABLES": 0, "XRFTHIP_FASTN_TC": tc, "XRFTHIP_FASTN_TR": tr}, profile=True)
    for dbg in (1, 2, 4, 7):
        one(*t, env={"XRFTHIP_FASTN_TABLES": 0, "XRFTHIP_FASTN_DBG": dbg}, profile=True)
