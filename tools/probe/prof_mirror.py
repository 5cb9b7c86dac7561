import cProfile, pstats, sys, runpy, io
sys.argv=["tools/mirror_bench.py","300"]
pr=cProfile.Profile(); pr.enable()
try:
    runpy.run_path("tools/mirror_bench.py", run_name="__main__")
finally:
    pr.disable()
    s=io.StringIO(); st=pstats.Stats(pr, stream=s); st.sort_stats("cumulative").print_callees("_verify_step_native")
    out=s.getvalue().splitlines()
    print("\n".join(out[:60]))
    s=io.StringIO(); st=pstats.Stats(pr, stream=s); st.sort_stats("tottime").print_stats("lantern_amd|ctypes|torch._C|built-in", 40)
    print("\n".join(s.getvalue().splitlines()[:70]))
