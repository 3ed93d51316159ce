"""cProfile of the benchmark script's Python side (ring-fed and resident updates, rollout legs): which host functions the steps spend
their time in (cumulative, srl_amd's own files).  python scripts/prof_step.py"""
import cProfile, pstats, sys, os, io
sys.argv = ["bench.py", "--steps", "6", "--warmup", "2", "--seeds", "0", "--no-cpu-baseline", "--no-profile", "--no-plain-copy", "--no-closed-loop"]
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "bench.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s).sort_stats("cumulative")
st.print_stats(r"(mappo|ingest|obs_ring|actor_critic|hipnet|h2path|trainer)\.py", 45)
print(s.getvalue()[-9000:])
