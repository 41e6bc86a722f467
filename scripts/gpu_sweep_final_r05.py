"""Round-5 closing sweep on the GPU, every random generator of tests/test_random_differential.py on the final tree (the geometry rules of the last commits touch every
lengths-as-data kernel): one-pass kernels, small slabs, table pipelines, one transform axis on table lengths and on any length (Rader / Bluestein lengths included), generic."""
import sys, os, warnings, collections, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
warnings.simplefilter("ignore")
import test_random_differential as t
bad = 0
def sweep(name, fn, seeds, **kw):
    global bad
    t0 = time.time(); n = 0
    for seed in seeds:
        n += 1
        try:
            fn(seed, **kw)
        except Exception as e:
            bad += 1
            print("FAIL", name, seed, kw, repr(e)[:400], flush=True)
    print(f"{name}: {n} cases, {time.time() - t0:.0f} s", flush=True)
sweep("one-pass", lambda s: t.run_random_one_pass(s, big=(s % 4 == 0)), range(7000, 7120))
sweep("small slab", t.run_random_small_slab, range(7200, 7400))
for dt in ("float64", "float32"):
    sweep("fastm " + dt, t.run_random_fastm, range(7400, 7440), dtype=dt)
sweep("fast", t.run_random_fast, range(7500, 7540))
sweep("one axis (table lengths)", t.run_random_one_axis, range(7600, 7720))
sweep("one axis (any length)", t.run_random_any_axis, range(7800, 8200))
sweep("generic", t.run_random, range(8300, 8360))
print("done, failures:", bad)
