import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_fastn import one
for t, knobs in (((32, 750, 1500, "float64"), [dict(), dict(XRFTHIP_FASTN_GC=4), dict(XRFTHIP_FASTN_GC=4, XRFTHIP_FASTN_TC=256), dict(XRFTHIP_FASTN_GC=2, XRFTHIP_FASTN_TC=256), dict(XRFTHIP_FASTN_GC=1)]),
                 ((16, 1215, 1215, "float32"), [dict(), dict(XRFTHIP_FASTN_GC=8), dict(XRFTHIP_FASTN_GC=2, XRFTHIP_FASTN_TC=256), dict(XRFTHIP_FASTN_TC=256), dict(XRFTHIP_FASTN_RPU=1), dict(XRFTHIP_FASTN_RPU=4), dict(XRFTHIP_FASTN_TR=512)]),
                 ((16, 3000, 3000, "float64"), [dict(), dict(XRFTHIP_FASTN_GC=1), dict(XRFTHIP_FASTN_GC=1, XRFTHIP_FASTN_RPU=1), dict(XRFTHIP_FASTN_RPU=1), dict(XRFTHIP_FASTN_TC=384, XRFTHIP_FASTN_TR=384)])):
    for k in knobs:
        one(*t, env=k, profile=True)
