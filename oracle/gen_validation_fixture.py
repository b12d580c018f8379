"""Writes tests/golden/validation_suite.json (the DATA of the reference's five validation cases) and a self-made
background-traffic file in the reference's schema.  TEST INFRASTRUCTURE; runs only in the build container, where
/root/reference is mounted:

    python oracle/gen_validation_fixture.py

  * validation_suite.json: values of /root/reference/torchdriveenv/data/validation_cases.yml as the WaypointSuite fields the
    step path consumes (ref gym_env.py:63-68, env_utils.py:20-28): locations, waypoint_suite, scenarios (agent_states,
    agent_attributes; the 132-float recurrent states feed the remote model only, ref gym_env.py:216,286-287, and are
    dropped), car_sequence_suite.  A run of identical replay rows (a parked car, ref env_utils.py:86-91) is stored as
    {"repeat": n, "row": [...]}.
  * background_traffic/carla_Town03_10_7.json: NOT a copy of a reference file: 24 agents drawn here (seed 7) on a ring
    110 - 230 m from the start of validation case 2, in the schema the reference reads at gym_env.py:207-216
    (location, agent_density, random_seed, agent_states[center{x,y}, orientation, speed], agent_attributes[length,
    width, rear_axis_offset, agent_type, waypoint], recurrent_states)."""
import json
import math
import os

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/torchdriveenv/data/validation_cases.yml"
OUT = os.path.join(ROOT, "tests", "golden")


def pack_sequence(rows):
    if len(rows) > 1 and all(r == rows[0] for r in rows):
        return {"repeat": len(rows), "row": rows[0]}
    return rows


def main():
    with open(SRC) as f:
        d = yaml.safe_load(f)
    suite = dict(locations=d["locations"], waypoint_suite=d["waypoint_suite"],
                 scenarios=[None if s is None else dict(agent_states=s["agent_states"], agent_attributes=s["agent_attributes"])
                            for s in d["scenarios"]],
                 car_sequence_suite=[None if c is None else {str(k): pack_sequence(v) for k, v in c.items()}
                                     for c in d["car_sequence_suite"]])
    with open(os.path.join(OUT, "validation_suite.json"), "w") as f:
        json.dump(suite, f, separators=(",", ":"))
    rng = np.random.default_rng(7)
    ego = d["waypoint_suite"][2][0]
    states, attrs = [], []
    for _ in range(24):
        r, th = rng.uniform(110.0, 230.0), rng.uniform(0.0, 2.0 * math.pi)
        states.append(dict(center=dict(x=round(ego[0] + r * math.cos(th), 2), y=round(ego[1] + r * math.sin(th), 2)),
                           orientation=round(float(rng.uniform(0.0, 2.0 * math.pi)), 2), speed=round(float(rng.uniform(0.0, 9.0)), 2)))
        attrs.append(dict(length=round(float(rng.uniform(4.2, 5.4)), 2), width=round(float(rng.uniform(1.8, 2.3)), 2),
                          rear_axis_offset=round(float(rng.uniform(1.5, 2.0)), 2), agent_type=None, waypoint=None))
    os.makedirs(os.path.join(OUT, "background_traffic"), exist_ok=True)
    with open(os.path.join(OUT, "background_traffic", "carla_Town03_10_7.json"), "w") as f:
        json.dump(dict(location="carla:Town03", agent_density=10, random_seed=7, agent_states=states, agent_attributes=attrs,
                       recurrent_states=[[0.0] for _ in states]), f, separators=(",", ":"))
    print("wrote", os.path.join(OUT, "validation_suite.json"), os.path.getsize(os.path.join(OUT, "validation_suite.json")), "bytes")


if __name__ == "__main__":
    main()
