"""Where the planner loop's host time goes (examples/franka_planner_loop.py through the reference-shaped classes): cProfile of 100
iterations after a warm-up run, by own time and by cumulative time.  python tools/studies/facade_profile.py [--moving]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "examples"))
import franka_planner_loop as f  # noqa: E402

moving = "--moving" in sys.argv
f.main(iters=10, n_traj=1024, horizon=32, quiet=True, moving=moving)
pr = cProfile.Profile()
pr.enable()
f.main(iters=100, n_traj=1024, horizon=32, quiet=True, moving=moving)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)
