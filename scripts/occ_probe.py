"""Experiment driver: kernel time of the wave engine vs replicas for the lib given by SF_AMD_LIB."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solverforge_amd as sfa
from solverforge_amd import datasets
p = datasets.make_cvrp(1000, 100, 55, seed=0)
for R in [int(x) for x in sys.argv[1:]]:
    d = sfa.build_cvrp(p, n_replicas=R)
    d.set_engine(2)
    d.configure(sfa.SolverConfig(random_seed=0))
    d.calculate_score(); d.phase_start()
    d.solve_steps(100); d.profile_solve()
    b = d.total_stats()
    for _ in range(3): d.solve_steps(100, sync=False)
    ms, n = d.profile_solve()
    a = d.total_stats()
    print(os.environ.get("SF_AMD_LIB", "default")[-16:], "R", R, "ms/launch %.2f" % (ms / n), "Gmoves/s %.2f" % ((a["moves_evaluated"] - b["moves_evaluated"]) / ms / 1e6), flush=True)
    d.close()
