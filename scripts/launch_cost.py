import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from torchdriveenv_amd import _abi, _lib, ops
_lib.kernel_override(step=os.environ.get('TDE_STEP'))   # (the scripts' own switch; the library reads no environment)
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world
dev = torch.device("cuda:0")
world = synthetic_world(n_scn=64, A=16, seed=0, n_maps=4)
dw = world.to_device(dev)
cfg = _abi.default_config(seed=1, distance_cutoff=0.25)
for B in (512, 1024, 2048, 4096, 8192, 16384):
    st = EnvState(B, 16, device=dev, with_info=False)
    ops.env_reset(cfg, dw, st)
    act = torch.zeros(B, 2, device=dev)
    out = torch.empty(B, 8, device=dev)
    res = {}
    for name, fn in (("state_obs", lambda: ops.state_obs(dw, st, out)), ("env_step", lambda: ops.env_step(cfg, dw, st, action=act))):
        for _ in range(100): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(1000): fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1), (time.perf_counter() - t0) * 1e3)
    print(f"B={B:6d}  state_obs dev {res['state_obs'][0]:.2f} us wall {res['state_obs'][1]:.2f} us | env_step dev {res['env_step'][0]:.2f} us wall {res['env_step'][1]:.2f} us  (kernel={os.environ.get('TDE_STEP','trio')})")
