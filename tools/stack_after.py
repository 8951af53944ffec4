"""Run a script and dump every thread's Python stack after N seconds (then keep going): python tools/stack_after.py 60 main_merging.py args..."""
import faulthandler, runpy, sys, time
import os
secs, script = int(sys.argv[1]), sys.argv[2]
sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
sys.argv = sys.argv[2:]
faulthandler.dump_traceback_later(secs, repeat=True)
t0 = time.time()
runpy.run_path(script, run_name="__main__")
print(f"[stack_after] total {time.time() - t0:.1f} s")
