import importlib, sys, numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
from parity_helpers import cartpole_model_and_weights
eng = importlib.import_module("muzero-hypermodel_amd.engine")
models = importlib.import_module("muzero-hypermodel_amd.models")
config = importlib.import_module("muzero-hypermodel_amd.games.cartpole").MuZeroConfig()
model, _ = cartpole_model_and_weights(models, config, "cuda")
E = 83
obs = torch.from_numpy(np.random.RandomState(4).uniform(-0.05, 0.05, (E, 4)).astype(np.float32)).cuda()
legal = [[0, 1] if e % 11 else [] for e in range(E)]
engine = eng.BatchedMCTS(config, E, seeds=[1000 + e for e in range(E)], group_width=16)
engine.configure_fused_fc(model)
engine.set_fused_options("narrow", publish_tree=False)
for r in range(4):
    out = engine.run_moves([obs] * 6, legal, [0] * E, np.ones(E), True)
    print(r, np.bincount(out["moves_done"], minlength=7), out["actions"][:, :12].tolist())
