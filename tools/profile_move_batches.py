"""Where does a board-game move batch spend its time?  TicTacToe with an FC-16 network, 32768 device envs,
DeviceSelfPlay.play_moves(18): cProfile of the host side of five batches; under `rocprofv3 --kernel-trace --stats` the same
run gives the kernel shares (profiles/r02_tictactoe_fc_batch_kernel_stats.csv).

    python tools/profile_move_batches.py"""
import cProfile, pstats, importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sp = importlib.import_module("muzero-hypermodel_amd.self_play"); models = importlib.import_module("muzero-hypermodel_amd.models")
mod = importlib.import_module("muzero-hypermodel_amd.games.tictactoe"); config = mod.MuZeroConfig()
config.network, config.encoding_size = "fullyconnected", 16
config.fc_representation_layers, config.fc_dynamics_layers = [], [16]
config.fc_reward_layers = config.fc_value_layers = config.fc_policy_layers = [16]
config.temperature_threshold = None
torch.manual_seed(0); weights = models.MuZeroNetwork(config).get_weights()
actor = sp.DeviceSelfPlay({"weights": weights}, "tictactoe", config, 0, 32768)
actor.engine.set_fused_options("auto", publish_tree=False)
n = [0]
def on_games(b): n[0] += len(b)
for _ in range(3): actor.play_moves(18, 1.0, on_games=on_games)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): actor.play_moves(18, 1.0, on_games=on_games)
actor.flush(on_games=on_games); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
