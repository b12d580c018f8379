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


def load_labeled_data(data_dir):                           # ref env_utils.py:31-105 (scenario-builder JSON export)
    suite = WaypointSuite(locations=[], waypoint_suite=[], scenarios=[], car_sequence_suite=[])
    suite.traffic_light_state_suite = []
    suite.stop_sign_suite = []
    for json_file in os.listdir(data_dir):
        if json_file[-5:] != ".json":
            continue
        suite.locations.append(json_file.split('_')[1])
        with open(os.path.join(data_dir, json_file)) as f:
            data = json.load(f)
        suite.waypoint_suite.append([[s['center']['x'], s['center']['y']]
                                     for s in data['individual_suggestions']['0']['states']])
        scenario, car_sequences = None, None
        agents = data.get("predetermined_agents")
        if agents is not None:
            states, attrs, recur = [], [], []
            for aid in agents:
                ag = agents[aid]
                speed = random.randint(5, 10) if len(ag['states']) == 1 else 0      # :68-71
                s0 = ag['states']['0']
                states.append([s0['center']['x'], s0['center']['y'], s0['orientation'], speed])
                sa = ag['static_attributes']
                attrs.append([sa['length'], sa['width'], sa['rear_axis_offset']])
                recur.append([0] * 132)
            if states:
                scenario = Scenario(agent_states=states, agent_attributes=attrs, recurrent_states=recur)
            car_sequences = {}
            for aid in agents:
                ag = agents[aid]
                s0 = ag['states']['0']
                if ag["static_attributes"].get("max_speed", None) == 0:            # parked car, :86-91
                    car_sequences[int(aid)] = [[s0['center']['x'], s0['center']['y'], s0['orientation'], 0]
                                               for _ in range(200)]
                elif len(ag['states']) > 1:                                          # :93-98
                    car_sequences[int(aid)] = [[ag['states'][i]['center']['x'], ag['states'][i]['center']['y'],
                                                ag['states'][i]['orientation'], 0] for i in ag['states']]
        suite.scenarios.append(scenario)
        suite.car_sequence_suite.append(car_sequences)
        suite.traffic_light_state_suite.append(None)
        suite.stop_sign_suite.append(None)
    return suite


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
