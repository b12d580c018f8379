"""Turn the golden cases (tests/golden/reward_golden.json, produced by oracle/gen_golden.py from the reference's
own gym_env.py) into flat per-step operator inputs and expected outputs."""
import os

import numpy as np

from torchdriveenv_amd import _abi


def case_config(case):
    c = case["config"]
    return _abi.default_config(waypoint_bonus=c["waypoint_bonus"], heading_penalty=c["heading_penalty"],
                               distance_bonus=c["distance_bonus"], distance_cutoff=c["distance_cutoff"],
                               max_steps=c["max_environment_steps"],
                               terminated_at_infraction=int(c["terminated_at_infraction"]))


def case_inputs(case):
    """Every step of a case is an independent operator call given the carried counters, so a whole case can be
    evaluated as one batch of n = T rows with the counters taken from the golden record itself."""
    st = np.asarray(case["states"], np.float32)
    T = len(case["steps"])
    pre, post = st[:T], st[1:T + 1]
    off = (np.asarray(case["offroad"][1:T + 1]) > 0).astype(np.uint8)   # gym_env.py:415 tests `> 0`
    col = (np.asarray(case["collision"][1:T + 1]) > 0).astype(np.uint8)
    tl = (np.asarray(case["traffic_light_violation"][1:T + 1]) > 0).astype(np.uint8)
    wp = np.asarray(case["waypoints"], np.float64)[None]
    wp_n = np.asarray([wp.shape[1]], np.int32)
    steps = np.arange(T, dtype=np.int32)                                  # environment_steps before the step
    target = np.asarray([s["target_idx_before"] for s in case["steps"]], np.int32)
    reached_after = np.asarray([s["reached_waypoint_num"] for s in case["steps"]], np.int32)
    reached_before = np.concatenate([[0], reached_after[:-1]]).astype(np.int32)
    return dict(pre=pre, post=post, off=off, col=col, tl=tl, wp=wp, wp_n=wp_n, scn=np.zeros(T, np.int32),
                steps=steps, target=target, reached=reached_before)


def case_expected(case):
    s = case["steps"]
    return dict(reward=np.asarray([x["reward"] for x in s], np.float64),
                terminated=np.asarray([x["terminated"] for x in s], np.uint8),
                truncated=np.asarray([x["truncated"] for x in s], np.uint8),
                target_after=np.asarray([x["target_idx_after"] for x in s], np.int32),
                reached=np.asarray([x["reached_waypoint_num"] for x in s], np.int32),
                info=np.asarray([[x["psi_smoothness"], x["speed_smoothness"], x["psi_reward"], x["dist_reward"]]
                                 for x in s], np.float64),
                is_success=np.asarray([x["is_success"] for x in s], np.uint8))


def write_validation_suite_yaml(path):
    """tests/golden/validation_suite.json (the data of the reference's five validation cases, oracle/gen_validation_fixture.py)
    written back as a YAML file in the reference's WaypointSuite schema (ref env_utils.py:20-28), so that tests go through
    `load_waypoint_suite_data` exactly as a user's file would"""
    import json
    import os

    import yaml

    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "validation_suite.json")) as f:
        d = json.load(f)

    def unpack(v):
        return [list(v["row"]) for _ in range(v["repeat"])] if isinstance(v, dict) else v

    d["car_sequence_suite"] = [None if c is None else {int(k): unpack(v) for k, v in c.items()} for c in d["car_sequence_suite"]]
    for s in d["scenarios"]:
        if s is not None:
            s["recurrent_states"] = None
    with open(path, "w") as f:
        yaml.safe_dump(d, f)
    return path


# a background-traffic file in the reference's schema (made by oracle/gen_validation_fixture.py, not a reference file)
BACKGROUND_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "background_traffic")
