import sys, os, cProfile, pstats
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "examples"))
import franka_planner_loop as f
pr = cProfile.Profile()
pr.enable()
m = f.main(iters=30, n_traj=1024, horizon=32, quiet=True)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
