import importlib, os, sys, torch
sys.path[:0]=[os.environ.get("GRAFT_REPO_ROOT","."), os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"tests")]
from parity_helpers import synthetic_model
models = importlib.import_module("muzero-hypermodel_amd.models")
config = importlib.import_module("muzero-hypermodel_amd.games.connect4").MuZeroConfig()
model,_ = synthetic_model(models, config, "cuda")
b=1024
state=torch.rand(b,64,6,7,device="cuda"); action=torch.randint(0,7,(b,1),device="cuda")
planes=models.state_action_planes(state,action,7)
with torch.no_grad():
    for _ in range(3): model._recurrent_tower(planes,None)
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): model._recurrent_tower(planes,None)
    g.replay(); torch.cuda.synchronize()
    a,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record(); 
    for _ in range(5): g.replay()
    e.record(); torch.cuda.synchronize()
    print("MZ_TOWER_DEBUG", os.environ.get("MZ_TOWER_DEBUG"), "tower us", 1e3*a.elapsed_time(e)/50)
