"""Stress of the one-step kernel forms against the oracle: every (agents per env, lights, magnitudes) variant of the three-role
kernel and the one-role kernel, several batch sizes (1 .. 2+ workgroups per CU), full state compared bit for bit after EVERY step.
Written after a build of the 32-slot variant re-spawned wrongly on one env-finish in a few hundred (profiles/r05_a32_respawn_anomaly.md):
    python scripts/stress_step_forms.py [steps]"""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
import torch

from oracle import oracle
from torchdriveenv_amd import _abi, _lib, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world

T = int(sys.argv[1]) if len(sys.argv) > 1 else 40
SKIP = ("action", "env_cache", "slot_cache", "act_cache", "obs")
bad = 0
for A in (8, 16, 32, 64, 128):
    for lights in (False, True):
        world = synthetic_world(n_scn=16, A=A, seed=2 + A, n_maps=2)
        flags = _abi.F_ALL | (_abi.F_TRAFFIC_LIGHTS if lights else 0)
        cfg = _abi.default_config(seed=21, flags=flags, max_steps=40, distance_cutoff=0.25)
        dw = world.to_device("cuda:0")
        for B in ((48, 600) if A == 128 else (512, 1024, 1536, 4096)):      # (128 slots: eight wavefronts per env / four / one role)
            for mag in (True, False):
                forms = ("trio", "solo") if A in (8, 16, 32) else ("auto", "duo", "solo") if A == 128 else ("solo",)
                hs = EnvState(B, A, with_magnitudes=mag)
                oracle.env_reset(cfg, world, hs)
                ds = [EnvState(B, A, device="cuda:0", with_obs=True, with_magnitudes=mag) for _ in forms]
                for d in ds:
                    d.load(hs.host())
                rng = np.random.default_rng(B + A)
                fails = {}
                for t in range(T):
                    act = np.stack([rng.uniform(-0.2, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
                    hs["action"][...] = act
                    oracle.env_step(cfg, world, hs)
                    a = torch.from_numpy(act).cuda()
                    for form, d in zip(forms, ds):
                        if form in fails:
                            continue
                        _lib.kernel_override(step=form)
                        ops.env_step(cfg, dw, d, action=a)
                        for k, h in hs.arrays.items():
                            if h is None or k in SKIP or d.arrays.get(k) is None:
                                continue
                            g = d[k].cpu().numpy()
                            if g.shape != np.asarray(h).shape:
                                continue
                            h = np.asarray(h)
                            same = np.array_equal(g, h, equal_nan=True) if h.dtype.kind == "f" else np.array_equal(g, h)
                            if k == "info":      # psi_reward: float64 cosine, libm on the host and the kernel's own on the device (one ulp)
                                g2, h2 = g.reshape(-1, 4), h.reshape(-1, 4)
                                same = np.array_equal(g2[:, [0, 1, 3]], h2[:, [0, 1, 3]]) and np.abs(g2[:, 2] - h2[:, 2]).max(initial=0.0) <= 8e-15
                            if k == "magnitudes":
                                same = np.array_equal(g.view(np.uint32), h.view(np.uint32))
                            if not same:
                                fails[form] = (t, k)
                                break
                _lib.kernel_override()
                tag = "FAIL %s" % fails if fails else "ok"
                bad += len(fails)
                print(f"A={A:3d} lights={int(lights)} B={B:5d} mag={int(mag)} forms={forms}: {tag}", flush=True)
print("stress:", "FAILED" if bad else "all equal to the oracle")
sys.exit(1 if bad else 0)
