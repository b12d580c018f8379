"""Minimal closed-loop use of the batched env (the shape of examples/rl_training.py in the reference, without the SB3
trainer): a random policy over 4096 envs, birdview observations with a frame stack of 3, episode statistics from the
info tensors the reference's trainer logs (ref examples/rl_training.py:46-63).

    python examples/random_policy.py [num_envs] [steps]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from torchdriveenv_amd.config import EnvConfig
from torchdriveenv_amd.env import BatchedWaypointEnv
from torchdriveenv_amd.synth import synthetic_world


def main():
    num_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    cfg = EnvConfig(seed=0, distance_cutoff=0.25, frame_stack=3)           # the shipped training config's reward terms
    world = synthetic_world(n_scn=64, A=16, seed=0, n_maps=4)               # or a WaypointSuite from the loaders
    env = BatchedWaypointEnv(cfg, world, num_envs=num_envs, frame_stack=3)
    obs = env.reset()                                                       # uint8 [B, 9, 64, 64] on the GPU
    low = torch.tensor(env.action_space.low, device=obs.device)
    high = torch.tensor(env.action_space.high, device=obs.device)
    ret = torch.zeros(num_envs, device=obs.device)
    stats = torch.zeros(4, device=obs.device)            # episodes, summed return, offroad endings, collision endings
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        action = low + (high - low) * torch.rand(num_envs, 2, device=obs.device)     # policy(obs) goes here
        obs, reward, terminated, truncated, info = env.step(action)
        ret += reward
        done = (terminated | truncated).float()          # statistics stay on the device: no host sync in the loop
        # (info["offroad"] / info["collision"] are MAGNITUDES, as in the reference, gym_env.py:427-428: > 0 counts the endings)
        stats += torch.stack([done.sum(), (ret * done).sum(), ((info["offroad"] > 0).float() * done).sum(),
                              ((info["collision"] > 0).float() * done).sum()])
        ret *= 1.0 - done
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{num_envs} envs x {steps} steps in {dt:.2f} s = {num_envs * steps / dt:.3g} env-steps/s (closed loop, birdview x3)")
    episodes, ep_return, offroad, collision = stats.tolist()
    if episodes:
        print(f"{int(episodes)} episodes: mean return {ep_return / episodes:.1f}, offroad {offroad / episodes:.0%}, "
              f"collision {collision / episodes:.0%}")


if __name__ == "__main__":
    main()
