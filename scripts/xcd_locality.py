"""Does the rasteriser gain from L2 locality?  Renders the same mid-episode batch with the envs (a) in their natural order
(scenarios drawn at random per env), (b) permuted so that the four views of workgroup w all come from scenarios with
scn % 8 == w % 8 (workgroups are dealt to the 8 XCDs round-robin: every XCD's L2 then serves 1/8 of the maps), (c) sorted by
scenario, (d) every env of ONE scenario (upper bound of locality).  Pixels are compared through the permutation."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torchdriveenv_amd import _abi, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
A = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
dw = world.to_device(dev)
cfg = _abi.default_config(seed=1, distance_cutoff=0.25)
st = EnvState(B, A, device=dev, with_info=False)
ops.env_reset(cfg, dw, st)
g = torch.Generator().manual_seed(0)
acts = torch.stack([torch.rand(50, B, generator=g) * 2 - 1, torch.rand(50, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
ops.env_rollout(cfg, dw, st, acts)
torch.cuda.synchronize()
base = {k: v.clone() for k, v in ((k, st[k]) for k in st.host().keys()) if v is not None}
scn = base["scn"].cpu().numpy()


def apply(perm):
    p = torch.as_tensor(perm, device=dev, dtype=torch.long)
    for k, v in base.items():
        t = st[k]
        if v.shape[0] == B:
            t.copy_(v[p])
        elif v.shape[0] == B * A:
            t.copy_(v.view(B, A, *v.shape[1:])[p].reshape(v.shape))


def timed(tag, perm, ref=None):
    apply(perm)
    out = ops.render_ego(cfg, dw, st)
    for _ in range(5):
        ops.render_ego(cfg, dw, st, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.render_ego(cfg, dw, st, out=out); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    same = None if ref is None else bool(torch.equal(out, ref[torch.as_tensor(perm, device=dev, dtype=torch.long)]))
    print(f"{tag:46s} median {statistics.median(ts):6.2f}  min {min(ts):6.2f} us   pixels equal (through the permutation): {same}", flush=True)
    return out


ident = np.arange(B)
ref = timed("natural order", ident)
# (b) workgroup w (views 4w .. 4w+3) <- envs of scenarios with scn % 8 == w % 8
buckets = [list(np.nonzero(scn % 8 == x)[0]) for x in range(8)]
perm = []
for w in range(B // 4):
    x = w % 8
    for _ in range(4):
        src = buckets[x] if buckets[x] else max(buckets, key=len)
        perm.append(src.pop())
timed("scn % 8 == workgroup % 8 (XCD-local maps)", np.array(perm), ref)
timed("sorted by scenario", np.argsort(scn, kind="stable"), ref)
one = np.nonzero(scn == np.bincount(scn).argmax())[0]
timed("one scenario only (locality upper bound)", np.resize(one, B), ref)
# (e) finer: scn -> XCD by map id (4 maps): scenarios of one map on two XCDs
smap = world.arrays["scn"]["map"][scn]
key = (smap * 2 + (scn // 4) % 2) % 8
buckets = [list(np.nonzero(key == x)[0]) for x in range(8)]
perm = []
for w in range(B // 4):
    x = w % 8
    for _ in range(4):
        src = buckets[x] if buckets[x] else max(buckets, key=len)
        perm.append(src.pop())
timed("map-major: one map on two XCDs", np.array(perm), ref)
