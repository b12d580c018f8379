import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd.config import EnvConfig
from torchdriveenv_amd.env import BatchedWaypointEnv
from torchdriveenv_amd.synth import synthetic_world
venv = BatchedWaypointEnv(EnvConfig(seed=0), synthetic_world(n_scn=64, A=16), num_envs=8192, frame_stack=3)
obs = venv.reset()
print(obs.shape, obs.dtype, obs.device)
obs, reward, terminated, truncated, info = venv.step(torch.zeros(8192, 2, device="cuda"))
reward_kb, done_kb = venv.rollout(torch.zeros(250, 8192, 2, device="cuda"))
print(reward_kb.shape, done_kb.shape)
from torchdriveenv_amd import ops
streams = [torch.cuda.Stream() for _ in range(3)]
actions, img = torch.zeros(250, 8192, 2, device="cuda"), None
ops.fork_streams(streams)
for t in range(250):
    img = ops.env_step_render(venv.tde_cfg, venv.dworld, venv.state, streams, action=actions[t], out=img)
ops.join_streams(streams)
torch.cuda.synchronize()
print(img.shape, "README snippet ok")
