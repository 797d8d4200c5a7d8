"""Dev: where does the host time of a Vis / CVP training step go?  cProfile over a few steps, top cumulative entries."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ["bench_train.py", "--arch", sys.argv[1] if len(sys.argv) > 1 else "vis", "--steps", "3", "--warmup", "2"]
import runpy
pr = cProfile.Profile()
pr.enable()
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_train.py"), run_name="__main__")
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(400)
lines = s.getvalue().splitlines()
print("\n".join(l[:170] for l in lines[:12]))
print("\n".join(l[:170] for l in lines if any(k in l for k in ("ops.py", "training.py", "_lib.py", "run_backward", "frontend.py", "model_cas.py", "adam", "bench_train.py"))))
