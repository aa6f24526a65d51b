"""cProfile of tools/selfplay_rate.py for one actor kind (where the host spends a move batch): python tools/selfplay_host_profile.py device-batch"""
import cProfile, pstats, sys, os
sys.argv = ["selfplay_rate.py", "--game", "tictactoe", "--envs", "65536", "--moves", "120", "--kinds", sys.argv[1]]
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import selfplay_rate
pr = cProfile.Profile()
pr.enable()
selfplay_rate.main()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
