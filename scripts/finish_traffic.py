"""gpurun_out/<tag>/traffic.json and traffic_config5.json in the form bench.py reads from profiles/ (roofline.traffic, roofline.valu_issue),
assembled from the PMC passes of scripts/measure_r06.sh in that directory; the note names the pass they come from."""
import json
import os
import re
import sys

O, tag = sys.argv[1], sys.argv[2]
ro = json.load(open(os.path.join(O, "traffic_rollout.json")))
sq = open(os.path.join(O, "pmc_sq_per_group_step.txt")).read()


def counter(name):
    m = re.search(name + r"\s+n=\s*\d+\s+mean/div=\s*([0-9.]+)", sq)
    return float(m.group(1)) if m else None


ro.update(valu_insts_per_group_step=counter("SQ_INSTS_VALU"), salu_insts_per_group_step=counter("SQ_INSTS_SALU"), clock_ghz=2.4,
          note=f"FETCH_SIZE / WRITE_SIZE / SQ_INSTS_* from separate rocprofv3 --pmc passes of scripts/run_rollout.py 159 (TDE_F_ALL incl. "
               f"TDE_F_NPC_FIRST_STEP) (scripts/measure_r06.sh, pass {tag}); FETCH_SIZE doubled (gfx950 counts 64 B per 128-B line moved: "
               f"profiles/r04_z_fetch_calibration.txt)")
json.dump(ro, open(os.path.join(O, "traffic.json"), "w"), indent=1)
rd, s32 = json.load(open(os.path.join(O, "traffic_render.json"))), json.load(open(os.path.join(O, "traffic_step32.json")))
c5 = dict(workload="configs[4]: 8192 envs x 32 agents, step + 64x64x3 birdview per timestep", envs=8192, agents=32,
          kernels={"render_views_kernel<64>": rd["hbm_bytes_per_launch"], "env_step_kernel<32>": s32["hbm_bytes_per_launch"]},
          hbm_bytes_per_timestep=rd["hbm_bytes_per_launch"] + s32["hbm_bytes_per_launch"],
          note=f"sum of the two kernels' FETCH_SIZE (doubled) + WRITE_SIZE per launch at 8192 views / envs, separate rocprofv3 --pmc passes of "
               f"scripts/run_render.py and scripts/run_step.py 200 solo 32 8192 (scripts/measure_r06.sh, pass {tag}); the sub-batch launches of "
               f"tde_env_step_render move the same bytes")
json.dump(c5, open(os.path.join(O, "traffic_config5.json"), "w"), indent=1)
print(json.dumps(dict(rollout_MB_per_launch=ro["hbm_bytes_per_launch"] / 1e6, valu=ro["valu_insts_per_group_step"], config5_MB_per_timestep=c5["hbm_bytes_per_timestep"] / 1e6)))
