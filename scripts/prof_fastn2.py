"""Thread counts / sequences per workgroup / ablations of csrc/fastn.h on a few shapes.  python scripts/prof_fastn2.py on the GPU box"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_fastn import one
S = [(16, 1215, 1215, "float32"), (16, 2200, 2200, "float32"), (16, 3000, 3000, "float64"), (32, 750, 1500, "float64")]
T = {"float32": (256, 320, 384, 512, 768, 1024), "float64": (192, 256, 320, 384, 512)}
print("== column pass: sequences per workgroup x threads (rows fixed at the default)")
for t in S:
    for gc in ((2, 4, 8) if t[3] == "float32" else (1, 2, 4)):
        for tc in T[t[3]]:
            one(*t, env={"XRFTHIP_FASTN_GC": gc, "XRFTHIP_FASTN_TC": tc}, profile=True)
print("== row pass: rows per workgroup x threads")
for t in S[:3]:
    for rpu in (1, 2, 4):
        for tr in T[t[3]]:
            one(*t, env={"XRFTHIP_FASTN_RPU": rpu, "XRFTHIP_FASTN_TR": tr}, profile=True)
print("== ablations (1 no LDS passes, 2 no stores, 4 no first pass)")
for t in S:
    for dbg in (0, 1, 2, 3, 4, 5, 6, 7):
        one(*t, env={"XRFTHIP_FASTN_DBG": dbg}, profile=True)
