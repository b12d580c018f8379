"""YAML / scenario-builder loaders with the reference's names and schemas (ref torchdriveenv/env_utils.py:10-123),
on PyYAML instead of omegaconf.  The reference's data files (data/validation_cases.yml, data/training_cases.yml) are
not redistributed here; point `load_waypoint_suite_data` at a copy, or set TORCHDRIVEENV_DATA to the directory that
holds them (the `torchdriveenv/data` directory of an installed reference package is searched too)."""
import json
import os
import random

import yaml

from .config import EnvConfig, RendererConfig, Scenario, SimulatorConfig, WaypointSuite


def construct_env_config(raw_config):                      # ref env_utils.py:10-12
    raw = dict(raw_config)
    sim = raw.pop("simulator", None)
    cfg = EnvConfig(**raw)
    if isinstance(sim, dict):
        rend = sim.pop("renderer", None)
        cfg.simulator = SimulatorConfig(**sim)
        if isinstance(rend, dict):
            cfg.simulator.renderer = RendererConfig(**rend)
    elif sim is not None:
        cfg.simulator = sim
    return cfg


def load_env_config(yaml_path):                            # ref env_utils.py:15-17
    with open(yaml_path) as f:
        return construct_env_config(yaml.safe_load(f) or {})


def _int_keys(d):
    return None if d is None else {int(k): v for k, v in d.items()}


def load_waypoint_suite_data(yaml_path):                   # ref env_utils.py:20-28
    with open(yaml_path) as f:
        raw = yaml.safe_load(f)
    data = WaypointSuite(**raw)
    if data.car_sequence_suite is not None:
        data.car_sequence_suite = [_int_keys(c) for c in data.car_sequence_suite]
    if data.scenarios is not None:
        data.scenarios = [Scenario(agent_states=s["agent_states"], agent_attributes=s["agent_attributes"],
                                   recurrent_states=s.get("recurrent_states")) if s is not None else None
                          for s in data.scenarios]
    return data


def _xy(state):
    return state["center"]["x"], state["center"]["y"]


def _labeled_case(doc):
    """one scenario-builder export -> (waypoints, Scenario | None, car_sequences | None)   ref env_utils.py:52-100"""
    waypoints = [list(_xy(st)) for st in doc["individual_suggestions"]["0"]["states"]]
    agents = doc.get("predetermined_agents")
    if agents is None:
        return waypoints, None, None
    rows, sizes, sequences = [], [], {}
    for key, ag in agents.items():
        first = ag["states"]["0"]
        x0, y0 = _xy(first)
        # a single recorded state means "drive on": the reference draws an integer speed in [5, 10]   (:68-71)
        speed = random.randint(5, 10) if len(ag["states"]) == 1 else 0
        rows.append([x0, y0, first["orientation"], speed])
        sa = ag["static_attributes"]
        sizes.append([sa["length"], sa["width"], sa["rear_axis_offset"]])
        if sa.get("max_speed", None) == 0:                       # parked: 200 copies of the first state   (:86-91)
            sequences[int(key)] = [[x0, y0, first["orientation"], 0] for _ in range(200)]
        elif len(ag["states"]) > 1:                              # recorded trajectory, speed column 0      (:93-98)
            sequences[int(key)] = [[*_xy(ag["states"][t]), ag["states"][t]["orientation"], 0] for t in ag["states"]]
    scenario = Scenario(agent_states=rows, agent_attributes=sizes,
                        recurrent_states=[[0] * 132 for _ in rows]) if rows else None
    return waypoints, scenario, sequences


def load_labeled_data(data_dir):
    """directory of scenario-builder JSON exports -> WaypointSuite (same result as ref env_utils.py:31-105)"""
    suite = WaypointSuite(locations=[], waypoint_suite=[], scenarios=[], car_sequence_suite=[])
    suite.traffic_light_state_suite, suite.stop_sign_suite = [], []
    for name in os.listdir(data_dir):
        if not name.endswith(".json"):
            continue
        with open(os.path.join(data_dir, name)) as f:
            waypoints, scenario, sequences = _labeled_case(json.load(f))
        suite.locations.append(name.split("_")[1])
        suite.waypoint_suite.append(waypoints)
        suite.scenarios.append(scenario)
        suite.car_sequence_suite.append(sequences)
        suite.traffic_light_state_suite.append(None)
        suite.stop_sign_suite.append(None)
    return suite


def load_background_traffic(json_path):
    """one `resources/background_traffic/*.json` file of the reference (the schema read at ref gym_env.py:207-216) ->
    dict(location, agent_density, random_seed, agent_states [[x, y, psi, v]], agent_attributes [[length, width,
    rear_axis_offset]]).  The recurrent states feed the remote model only (ref gym_env.py:216,286-287) and are dropped."""
    with open(json_path) as f:
        doc = json.load(f)
    states = [[*_xy(st), st["orientation"], st["speed"]] for st in doc["agent_states"]]
    attrs = [[at["length"], at["width"], at["rear_axis_offset"]] for at in doc["agent_attributes"]]
    return dict(location=doc["location"], agent_density=doc["agent_density"], random_seed=doc.get("random_seed"),
                agent_states=states, agent_attributes=attrs)


def _background_dirs():
    return [os.path.join(root, "background_traffic") for root in _data_path()] + \
           [os.path.join(os.path.dirname(root), "resources", "background_traffic") for root in _data_path()]


def pick_background_traffic(location, background_dir=None, rng=random):
    """the file choice of ref gym_env.py:202-220: a random file of the directory whose name carries the town of
    `location` (map name "carla_Town03", or the suite's bare "Town03", == name.split("_")[1]), redrawn until
    agents + density < 100.
    Returns the loaded dict, or None when the directory holds no file for that town."""
    dirs = [background_dir] if background_dir else _background_dirs()
    town = location[6:] if location.startswith("carla_") else location
    for d in dirs:
        if not d or not os.path.isdir(d):
            continue
        names = sorted(n for n in os.listdir(d) if n.endswith(".json") and len(n.split("_")) > 1
                       and n.split("_")[1] == town)
        docs = []
        for n in names:
            bt = load_background_traffic(os.path.join(d, n))
            if len(bt["agent_states"]) + bt["agent_density"] < 100:
                docs.append(bt)
        if docs:
            return rng.choice(docs)
    return None


def _data_path():
    roots = []
    if os.environ.get("TORCHDRIVEENV_DATA"):
        roots.append(os.environ["TORCHDRIVEENV_DATA"])
    roots.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data"))
    try:
        import importlib.util

        spec = importlib.util.find_spec("torchdriveenv")
        if spec is not None and spec.submodule_search_locations:
            roots += [os.path.join(p, "data") for p in spec.submodule_search_locations]
    except Exception:
        pass
    return roots


def _load_default_data(file_name):                          # ref env_utils.py:108-115
    for root in _data_path():
        file_path = os.path.join(root, file_name)
        if os.path.exists(file_path):
            return load_waypoint_suite_data(file_path)
    return None


def load_default_validation_data():                         # ref env_utils.py:118
    return _load_default_data(file_name="validation_cases.yml")


def load_default_train_data():                              # ref env_utils.py:122
    return _load_default_data(file_name="training_cases.yml")


def set_seeds(seed, logger=None):                           # ref helpers.py:39-49
    import numpy as np
    import torch

    if seed is None:
        seed = np.random.randint(low=0, high=2**32 - 1)
    if logger is not None:
        logger.info(f"seed: {seed}")
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    return seed
