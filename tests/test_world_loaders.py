import math
import os

import numpy as np
import pytest
import yaml

from torchdriveenv_amd import _abi
from torchdriveenv_amd.config import EnvConfig, Scenario, WaypointSuite, to_tde_config
from torchdriveenv_amd.loaders import construct_env_config, load_env_config, load_waypoint_suite_data
from torchdriveenv_amd.sharding import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_env_config_defaults_match_reference():
    c = EnvConfig()          # ref gym_env.py:34-54
    assert (c.ego_only, c.max_environment_steps, c.frame_stack) == (False, 200, 3)
    assert (c.waypoint_bonus, c.heading_penalty, c.distance_bonus, c.distance_cutoff) == (100., 25., 1., 0.5)
    assert c.use_background_traffic and c.terminated_at_infraction and c.seed is None
    assert c.render_mode == "rgb_array" and c.video_res == 1024 and c.video_fov == 500 and c.device is None
    assert c.simulator.renderer.left_handed_coordinates and c.simulator.renderer.highlight_ego_vehicle
    assert c.simulator.collision_metric == "nograd" and c.simulator.offroad_threshold == 0.5
    t = to_tde_config(c, seed=5, flags=_abi.F_ALL)
    assert (t.waypoint_bonus, t.heading_penalty, t.distance_cutoff, t.max_steps, t.seed) == (100., 25., 0.5, 200, 5)
    assert abs(t.dt - 0.1) < 1e-8 and t.reach_radius == 3.0


def test_yaml_loaders(tmp_path):
    (tmp_path / "env.yml").write_text(yaml.safe_dump(dict(ego_only=True, distance_cutoff=0.25, frame_stack=3,
                                                          simulator=dict(offroad_threshold=0.4))))
    c = load_env_config(str(tmp_path / "env.yml"))
    assert c.ego_only and c.distance_cutoff == 0.25 and c.simulator.offroad_threshold == 0.4
    assert construct_env_config(dict(waypoint_bonus=10.0)).waypoint_bonus == 10.0
    suite = dict(locations=["Town01", "Town02"], waypoint_suite=[[[0, 0], [10, 0], [20, 0]], [[0, 0], [0, 15]]],
                 car_sequence_suite=[{1: [[5, 3, 0, 0]] * 4}, None],
                 scenarios=[dict(agent_states=[[5, 3, 0, 0]], agent_attributes=[[5, 2, 2]], recurrent_states=[[0.0]]), None])
    (tmp_path / "suite.yml").write_text(yaml.safe_dump(suite))
    d = load_waypoint_suite_data(str(tmp_path / "suite.yml"))
    assert isinstance(d, WaypointSuite) and isinstance(d.scenarios[0], Scenario) and d.scenarios[1] is None
    assert list(d.car_sequence_suite[0].keys()) == [1] and d.locations == ["Town01", "Town02"]


def test_world_from_threeway_fixture():
    """BASELINE configs[0] inputs: validation case 0 (data extracted by oracle/gen_golden.py)"""
    import json

    from torchdriveenv_amd.env import world_from_waypoint_suite

    t = json.load(open(os.path.join(ROOT, "tests", "golden", "threeway_scenario.json")))
    parked = t["parked_replay_example"]
    sc = t["scenario"]
    # ego + the 2 scenario agents + 2 parked replay cars next to the route = "ego + 4 replay NPCs"
    wp = t["waypoints"]
    extra = [[wp[2][0] + 4.0, wp[2][1] + 3.0, 0.0, 0.0], [wp[4][0] - 3.0, wp[4][1] + 4.0, 0.0, 0.0]]
    seqs = {3: [extra[0]] * parked["length"], 4: [extra[1]] * parked["length"]}
    data = WaypointSuite(locations=[t["location"]], waypoint_suite=[wp], car_sequence_suite=[seqs],
                         scenarios=[Scenario(agent_states=sc["agent_states"] + extra,
                                             agent_attributes=sc["agent_attributes"] + [[5.0, 2.0, 2.0]] * 2)])
    w = world_from_waypoint_suite(data, agents_per_env=8)
    assert w.A == 8 and w.n_scn == 1 and w.arrays["scn"]["wp_n"][0] == len(wp)
    sp = w.arrays["spawn"][0]
    assert sp["present"].tolist() == [1, 1, 1, 1, 1, 0, 0, 0]
    assert sp["replay"][3] >= 0 and sp["replay_len"][3] == 300 and sp["route"][1] >= 0 and sp["replay"][1] == -1
    assert math.isclose(w.arrays["scn"]["start_heading"][0],
                        math.atan2(wp[1][1] - wp[0][1], wp[1][0] - wp[0][0]), rel_tol=1e-6)
    # every waypoint lies on the synthetic drivable corridor
    from oracle import oracle
    m = w.arrays["maps"][0]
    tri = w.arrays["tri"][m["tri_base"]:m["tri_base"] + m["n_tri"]]
    assert all(oracle.point_mesh_d2(np.float32(x), np.float32(y), tri) == 0.0 for x, y in wp)


def test_shard_ranges_partition_the_batch():
    for total, ws in ((65536, 8), (10, 3), (7, 8)):
        r = [shard_range(k, ws, total) for k in range(ws)]
        assert r[0][0] == 0 and r[-1][1] == total and all(a[1] == b[0] for a, b in zip(r, r[1:]))
    assert shard_range(3, 8, 65536) == (3 * 8192, 4 * 8192)


REF_DATA = "/root/reference/torchdriveenv/data"


def test_reference_suites_load_and_step_with_the_oracle(tmp_path):
    """the reference's own validation suite (its DATA, committed as tests/golden/validation_suite.json) goes through the
    loaders and the world builder unchanged; runs anywhere (no /root/reference)"""
    from oracle import oracle
    from tests.golden_util import write_validation_suite_yaml
    from torchdriveenv_amd.env import world_from_waypoint_suite
    from torchdriveenv_amd.state import EnvState

    val = load_waypoint_suite_data(write_validation_suite_yaml(str(tmp_path / "validation_cases.yml")))
    assert val.locations == ["Town07", "Town07", "Town03", "Town03", "Town01"] and len(val.waypoint_suite) == 5
    assert list(val.car_sequence_suite[1].keys()) == [1] and len(val.car_sequence_suite[1][1]) == 300
    assert len(val.scenarios[0].agent_states) == 2
    w = world_from_waypoint_suite(val, agents_per_env=8)
    assert w.n_scn == 5 and w.arrays["spawn"][1, 1]["replay_len"] == 300
    cfg = _abi.default_config(seed=1, distance_cutoff=0.25)
    st = EnvState(10, 8)
    oracle.env_reset(cfg, w, st)
    assert set(st["scn"]) <= set(range(5))
    for _ in range(60):
        st["action"][...] = [0.5, 0.0]
        oracle.env_step(cfg, w, st)
    assert np.isfinite(st["x"]).all() and st["episode"].min() >= 1
    # the parked replay car of case 1 stays where the file puts it
    e = int(np.nonzero(st["scn"] == 1)[0][0]) if (st["scn"] == 1).any() else None
    if e is not None:
        assert abs(st["x"][e * 8 + 1] - (-55.70970916748047)) < 1e-4 and st["v"][e * 8 + 1] == 0.0


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DATA, "validation_cases.yml")),
                    reason="reference data files are only present in the build container")
def test_committed_validation_fixture_equals_the_reference_file(tmp_path):
    """the committed fixture is the reference's data, value for value (and the 100-case training suite still loads)"""
    from tests.golden_util import write_validation_suite_yaml

    ref = load_waypoint_suite_data(os.path.join(REF_DATA, "validation_cases.yml"))
    got = load_waypoint_suite_data(write_validation_suite_yaml(str(tmp_path / "v.yml")))
    assert got.locations == ref.locations and got.waypoint_suite == ref.waypoint_suite
    assert got.car_sequence_suite == ref.car_sequence_suite
    for a, b in zip(got.scenarios, ref.scenarios):
        assert (a is None) == (b is None)
        if a is not None:
            assert a.agent_states == b.agent_states and a.agent_attributes == b.agent_attributes
    train = load_waypoint_suite_data(os.path.join(REF_DATA, "training_cases.yml"))
    assert len(train.waypoint_suite) == 100 and all(5 <= len(wp) <= 20 for wp in train.waypoint_suite)
    assert all(s is None for s in train.scenarios)


def test_load_labeled_data_schema(tmp_path):
    """scenario-builder export schema (ref env_utils.py:31-105): parked car -> 200 identical states, recorded
    trajectory -> one row per state with speed 0, single-state agents get an integer speed in [5, 10]"""
    import json

    from torchdriveenv_amd.loaders import load_labeled_data

    st = lambda x, y, o: {"center": {"x": x, "y": y}, "orientation": o}   # noqa: E731
    attrs = {"length": 4.5, "width": 2.0, "rear_axis_offset": 1.5}
    doc = {"individual_suggestions": {"0": {"states": [st(1, 2, 0), st(3, 4, 0)]}},
           "predetermined_agents": {
               "1": {"states": {"0": st(5, 6, 0.5)}, "static_attributes": dict(attrs, max_speed=0)},
               "2": {"states": {"0": st(7, 8, 0.1), "1": st(8, 8, 0.1)}, "static_attributes": attrs},
               "3": {"states": {"0": st(9, 9, 0.2)}, "static_attributes": attrs}}}
    (tmp_path / "case_Town03_x.json").write_text(json.dumps(doc))
    (tmp_path / "notes.txt").write_text("ignored")
    (tmp_path / "case_Town01_y.json").write_text(json.dumps({"individual_suggestions": doc["individual_suggestions"]}))
    s = load_labeled_data(str(tmp_path))
    i = s.locations.index("Town03")
    assert sorted(s.locations) == ["Town01", "Town03"] and s.waypoint_suite[i] == [[1, 2], [3, 4]]
    seq = s.car_sequence_suite[i]
    assert len(seq[1]) == 200 and seq[1][0] == [5, 6, 0.5, 0] and len(seq[2]) == 2 and 3 not in seq
    rows = s.scenarios[i].agent_states
    assert rows[1] == [7, 8, 0.1, 0] and 5 <= rows[0][3] <= 10 and 5 <= rows[2][3] <= 10
    assert s.scenarios[i].agent_attributes[0] == [4.5, 2.0, 1.5] and len(s.scenarios[i].recurrent_states[0]) == 132
    j = s.locations.index("Town01")
    assert s.scenarios[j] is None and s.car_sequence_suite[j] is None
    assert s.traffic_light_state_suite == [None, None] and s.stop_sign_suite == [None, None]


def _bg_doc(town, density, agents):
    return {"location": f"carla:{town}", "agent_density": density, "random_seed": 1,
            "agent_states": [{"center": {"x": x, "y": y}, "orientation": o, "speed": v} for x, y, o, v in agents],
            "agent_attributes": [{"length": 4.5 + 0.1 * k, "width": 2.0, "rear_axis_offset": 1.7, "agent_type": None,
                                  "waypoint": None} for k in range(len(agents))],
            "recurrent_states": [{"packed": [0.0] * 4} for _ in agents]}


def test_background_traffic_files(tmp_path):
    """background-traffic schema and file choice of ref gym_env.py:200-220: town taken from the file name, files with
    agents + density >= 100 are never drawn, recurrent states are dropped"""
    import json
    import random

    from torchdriveenv_amd.loaders import load_background_traffic, pick_background_traffic

    ag = [(0.0, 0.0, 0.1, 5.0), (150.0, 0.0, 3.1, 7.0)]
    (tmp_path / "carla_Town03_10_1.json").write_text(json.dumps(_bg_doc("Town03", 10, ag)))
    (tmp_path / "carla_Town03_99_2.json").write_text(json.dumps(_bg_doc("Town03", 99, ag)))    # 2 + 99 >= 100
    (tmp_path / "carla_Town04_10_3.json").write_text(json.dumps(_bg_doc("Town04", 10, ag[:1])))
    bt = load_background_traffic(str(tmp_path / "carla_Town03_10_1.json"))
    assert bt["agent_states"][1] == [150.0, 0.0, 3.1, 7.0] and bt["agent_attributes"][1] == [4.6, 2.0, 1.7]
    assert "recurrent_states" not in bt
    for s in range(8):
        got = pick_background_traffic("carla_Town03", str(tmp_path), rng=random.Random(s))
        assert got["agent_density"] == 10 and len(got["agent_states"]) == 2
    assert len(pick_background_traffic("Town04", str(tmp_path))["agent_states"]) == 1      # the suites' bare town name
    assert pick_background_traffic("carla_Town07", str(tmp_path)) is None
    assert pick_background_traffic("carla_Town03", str(tmp_path / "missing")) is None


def test_world_with_background_traffic(tmp_path):
    """ref gym_env.py:223-231: the ego takes the first background agent's attributes; only background agents farther
    than 100 m from the ego start are kept; they come after the scenario's own agents; ego_only drops them all"""
    import json

    from torchdriveenv_amd.config import Scenario, WaypointSuite
    from torchdriveenv_amd.env import world_from_waypoint_suite

    ag = [(5.0, 0.0, 0.0, 5.0),          # 5 m from the ego start: dropped (near field)
          (0.0, 99.0, 0.0, 5.0),         # 99 m: dropped
          (0.0, 180.0, 1.0, 6.0),        # 180 m: kept, second nearest
          (120.0, 0.0, 3.1, 7.0),        # 120 m: kept, nearest
          (0.0, 400.0, 0.0, 5.0)]        # beyond background_radius: dropped
    (tmp_path / "carla_Town03_10_1.json").write_text(json.dumps(_bg_doc("Town03", 10, ag)))
    data = WaypointSuite(locations=["carla_Town03"], waypoint_suite=[[[0.0, 0.0], [15.0, 0.0], [30.0, 0.0]]],
                         scenarios=[Scenario(agent_states=[[20.0, 0.0, 0.0, 3.0]], agent_attributes=[[4.0, 1.9, 1.5]],
                                             recurrent_states=[[0] * 132])],
                         car_sequence_suite=[None])
    w = world_from_waypoint_suite(data, agents_per_env=8, background=str(tmp_path))
    sp = w.arrays["spawn"].reshape(-1, 8)[0]
    assert list(sp["present"][:5]) == [1, 1, 1, 1, 0]
    assert np.allclose([sp["len"][0], sp["wid"][0], sp["lr"][0]], [4.5, 2.0, 1.7])          # attributes of bg agent 0
    assert np.allclose([sp["x"][1], sp["len"][1]], [20.0, 4.0])                              # scenario agent first
    assert np.allclose([sp["x"][2], sp["y"][2], sp["len"][2]], [120.0, 0.0, 4.8])            # nearest kept bg agent
    assert np.allclose([sp["x"][3], sp["y"][3], sp["v"][3]], [0.0, 180.0, 6.0])
    # fewer free slots than kept agents: the nearest ones win
    w4 = world_from_waypoint_suite(data, agents_per_env=4, background=str(tmp_path))
    sp4 = w4.arrays["spawn"].reshape(-1, 4)[0]
    assert list(sp4["present"]) == [1, 1, 1, 1] and np.allclose(sp4["x"][2:], [120.0, 0.0])
    # more scenario agents than NPC slots (the reference assembles up to ~100 agents, gym_env.py:216-237): dropped LOUDLY
    many = WaypointSuite(locations=["carla_Town03"], waypoint_suite=[[[0.0, 0.0], [15.0, 0.0], [30.0, 0.0]]],
                         scenarios=[Scenario(agent_states=[[20.0 + 8 * k, 0.0, 0.0, 3.0] for k in range(5)],
                                             agent_attributes=[[4.0, 1.9, 1.5]] * 5, recurrent_states=[[0] * 132] * 5)],
                         car_sequence_suite=[None])
    import pytest
    with pytest.warns(UserWarning, match="5 non-ego agents but only 3 NPC slots"):
        wm = world_from_waypoint_suite(many, agents_per_env=4)
    assert list(wm.arrays["spawn"].reshape(-1, 4)[0]["present"]) == [1, 1, 1, 1]
    # no file for the town -> unchanged world; ego_only -> the ego alone
    w0 = world_from_waypoint_suite(data, agents_per_env=8, background=lambda loc: None)
    assert list(w0.arrays["spawn"].reshape(-1, 8)[0]["present"][:3]) == [1, 1, 0]
    we = world_from_waypoint_suite(data, agents_per_env=8, background=str(tmp_path), ego_only=True)
    assert list(we.arrays["spawn"].reshape(-1, 8)[0]["present"][:2]) == [1, 0]
    # the background agents stand on drivable surface and step with the oracle without going offroad at once
    from oracle import oracle
    from torchdriveenv_amd import _abi
    from torchdriveenv_amd.state import EnvState

    cfg = _abi.default_config(seed=3)
    st = EnvState(1, 8)
    oracle.env_reset(cfg, w, st)
    st["action"][:] = 0.0
    oracle.env_step(cfg, w, st)
    assert st["present"][:4].tolist() == [1, 1, 1, 1] and st["offroad"][:4].tolist() == [0, 0, 0, 0]


def test_world_save_load_round_trip(tmp_path):
    from torchdriveenv_amd.synth import synthetic_world
    from torchdriveenv_amd.world import World

    w = synthetic_world(n_scn=6, A=8, seed=1, n_maps=2)
    p = str(tmp_path / "world.npz")
    w.save(p)
    w2 = World.load(p)
    assert w.ints == w2.ints and w.has_lights == w2.has_lights
    for k, a in w.arrays.items():
        assert a.dtype == w2.arrays[k].dtype and a.tobytes() == w2.arrays[k].tobytes(), k


REF_BG = "/root/reference/torchdriveenv/resources/background_traffic"


@pytest.mark.skipif(not (os.path.isdir(REF_BG) and os.path.exists(os.path.join(REF_DATA, "validation_cases.yml"))),
                    reason="reference data not present (GPU box)")
def test_reference_background_traffic_files_populate_the_world():
    """the reference's own 75 background-traffic files parse, and validation case 2 (Town03) gets its free slots
    filled with agents that are > 100 m from the ego start"""
    import random

    from torchdriveenv_amd.config import WaypointSuite
    from torchdriveenv_amd.env import world_from_waypoint_suite
    from torchdriveenv_amd.loaders import load_background_traffic, load_waypoint_suite_data, pick_background_traffic

    files = sorted(os.listdir(REF_BG))
    assert len(files) == 75
    for n in files[::9]:
        bt = load_background_traffic(os.path.join(REF_BG, n))
        assert len(bt["agent_states"]) == len(bt["agent_attributes"]) > 0 and len(bt["agent_states"][0]) == 4
    val = load_waypoint_suite_data(os.path.join(REF_DATA, "validation_cases.yml"))
    one = WaypointSuite(locations=val.locations[2:3], waypoint_suite=val.waypoint_suite[2:3],
                        scenarios=val.scenarios[2:3], car_sequence_suite=val.car_sequence_suite[2:3])
    bt = pick_background_traffic(one.locations[0], REF_BG, random.Random(0))
    w = world_from_waypoint_suite(one, agents_per_env=8, background=lambda loc: bt, background_radius=160.0)
    sp = w.arrays["spawn"].reshape(-1, 8)[0]
    n_own = len(one.scenarios[0].agent_states) if one.scenarios[0] is not None else 0
    ego = np.array(one.waypoint_suite[0][0])
    kept = [k for k in range(1 + n_own, 8) if sp["present"][k]]
    assert kept, "no background agent between 100 m and 160 m for this file"
    for k in kept:
        assert 100.0 < math.dist(ego, (sp["x"][k], sp["y"][k])) <= 160.0
    assert np.allclose([sp["len"][0], sp["wid"][0], sp["lr"][0]], bt["agent_attributes"][0])


def test_light_groups_share_the_grid_of_their_mesh():
    """assemble_world(light_groups=): a town where EVERY scenario junction is signalised (more lights than the 32 bits of one map's
    red mask) - one descriptor per group, the grid tables stored once, every scenario bound to its junction's descriptor; the
    oracle sees red lights there and none at the plain scenarios"""
    from oracle import oracle
    from torchdriveenv_amd import _abi
    from torchdriveenv_amd.state import EnvState
    from torchdriveenv_amd.synth import synthetic_town

    kw = dict(n_scn=40, A=8, seed=4, n_streets=8, spacing=100.0, cell=0.5)
    plain, lit = synthetic_town(**kw), synthetic_town(n_signals=36, **kw)
    assert plain.ints["n_maps"] == 1 and lit.ints["n_maps"] == 37 and not plain.has_lights and lit.has_lights
    for k in ("cell_word", "cell_tri", "cell_cls2", "cell_sub", "cell_coarse", "tri", "spawn", "wp_xy"):
        assert np.array_equal(np.asarray(plain.arrays[k]).view(np.uint8), np.asarray(lit.arrays[k]).view(np.uint8)), k
    m = lit.arrays["maps"]
    # a group = the scenario junction's four stop lines, then those of its signalised neighbours (3 - 5 junctions: 8 streets)
    assert m["n_stop"][0] == 0 and set(m["n_stop"][1:]) <= {12, 16, 20} and len(lit.arrays["stoplines"]) == m["n_stop"].sum()
    assert (m["stop_base"][1:] == np.cumsum(m["n_stop"])[:-1]).all() and len(set(m["cell_base"])) == 1 and len(set(m["rec_base"])) == 1
    assert list(lit.arrays["scn"]["map"]) == [1 + (s % 36) for s in range(36)] + [1 + s for s in range(4)]
    for s in range(36):
        d = m[lit.arrays["scn"]["map"][s]]
        sl = lit.arrays["stoplines"][d["stop_base"]:d["stop_base"] + d["n_stop"]]
        assert sl["light"].max() == d["n_stop"] // 2 - 1 and sorted(set(sl["light"][:4])) == [0, 1]
        c = np.array([sl["x"][:4].mean(), sl["y"][:4].mean()])                   # its own junction: the ego's route starts on an arm of it
        w0 = lit.arrays["wp_xy"][s, 0]
        assert 50.0 < np.hypot(*(w0 - c)) < 110.0
        for k in range(1, d["n_stop"] // 4):                                     # the neighbours': one street spacing away
            ck = np.array([sl["x"][4 * k:4 * k + 4].mean(), sl["y"][4 * k:4 * k + 4].mean()])
            assert 80.0 < np.hypot(*(ck - c)) < 120.0
    cfg = _abi.default_config(seed=2, distance_cutoff=0.25, flags=_abi.F_ALL | _abi.F_TRAFFIC_LIGHTS, max_steps=150)
    hs = EnvState(40, 8)
    oracle.env_reset(cfg, lit, hs)
    acts = np.zeros((160, 40, 2), np.float32)
    acts[..., 0] = 0.6
    _, done = oracle.env_rollout(cfg, lit, hs, acts)
    assert (done & 16).any()                              # some ego crossed a red stop line of ITS junction
    with pytest.raises(AssertionError, match="another mesh"):
        from torchdriveenv_amd.world import assemble_world
        sc = dict(map=0, waypoints=[(0, 0), (10, 0)], start_heading=0.0, lights=0)
        tri = np.array([[[0, -5], [20, -5], [20, 5]]], np.float32)
        assemble_world([tri, tri], [sc], 1, light_groups=[dict(map=1, stoplines=[(5, 0, 0, 1, 3, 0)], phases=[(10, [0])])])


def test_road_meshes_hook_gives_the_callers_map(tmp_path):
    """world_from_waypoint_suite(road_meshes=...) (ref gym_env.py:312, 184, 260: find_map_config(location).road_mesh handed to
    the simulator): scenarios of a location that has a mesh share ONE map built from it - dict of arrays, dict of .npy paths,
    callable, verts / faces - and a location without one keeps its synthetic corridor; the oracle's offroad flags on the
    assembled world are those of the caller's triangles (brute force)"""
    from oracle import oracle
    from torchdriveenv_amd.env import mesh_from_verts_faces, world_from_waypoint_suite
    from torchdriveenv_amd.state import EnvState
    from torchdriveenv_amd.synth import Town

    town = Town(n=3, spacing=100.0, ext=30.0)
    tri = town.mesh()
    suites = [[[float(v) for v in town.F(20.0 + 25.0 * k + 14.0 * i, 0.0)] for i in range(8)] for k in range(4)]
    data = WaypointSuite(locations=["TownA", "TownA", "TownB", "TownA"], waypoint_suite=suites, car_sequence_suite=[None] * 4,
                         scenarios=[None] * 4)
    np.save(tmp_path / "a.npy", tri.astype(np.float32))
    verts, faces = tri.reshape(-1, 2), np.arange(3 * len(tri)).reshape(-1, 3)
    forms = ({"TownA": tri}, {"TownA": str(tmp_path / "a.npy")}, lambda loc: tri if loc == "TownA" else None,
             {"TownA": mesh_from_verts_faces(verts, faces)})
    worlds = [world_from_waypoint_suite(data, agents_per_env=4, road_meshes=f) for f in forms]
    for w in worlds:
        assert w.ints["n_maps"] == 2 and list(w.arrays["scn"]["map"]) == [0, 0, 1, 0]      # TownA shared, TownB its corridor
        assert w.arrays["maps"]["n_tri"][0] == len(tri)
        assert np.array_equal(w.arrays["tri"][:len(tri)], tri.astype(np.float32).reshape(-1, 6))
        assert np.array_equal(w.arrays["cell_word"], worlds[0].arrays["cell_word"])
    with pytest.raises(ValueError):
        world_from_waypoint_suite(data, agents_per_env=4, road_meshes={"TownA": np.zeros((3, 5))})
    # an ego driving across the town's block interior is off THIS mesh (a corridor around its own waypoints would hold it)
    w = worlds[0]
    cfg = to_tde_config(EnvConfig(seed=1, terminated_at_infraction=False), 1, _abi.F_ALL & ~_abi.F_AUTORESET)
    hs = EnvState(8, 4)
    oracle.env_reset(cfg, w, hs)
    seen = 0
    for t in range(120):
        hs["action"][...] = np.array([0.5, 0.25], np.float32)
        oracle.env_step(cfg, w, hs)
        x, y, psi = hs["x"][::4], hs["y"][::4], hs["psi"][::4]
        s, c = np.sin(psi), np.cos(psi)
        hl, hw = 0.5 * hs["len"][::4], 0.5 * hs["wid"][::4]
        for e in range(8):
            m = int(w.arrays["scn"]["map"][hs["scn"][e]])
            if m != 0:
                continue
            far = any(oracle.point_mesh_d2(x[e] + sx * hl[e] * c[e] - sy * hw[e] * s[e], y[e] + sx * hl[e] * s[e] + sy * hw[e] * c[e],
                                           tri.astype(np.float32)) > 0.25 + 1e-3 for sx in (-1, 1) for sy in (-1, 1))
            near = all(oracle.point_mesh_d2(x[e] + sx * hl[e] * c[e] - sy * hw[e] * s[e], y[e] + sx * hl[e] * s[e] + sy * hw[e] * c[e],
                                            tri.astype(np.float32)) < 0.25 - 1e-3 for sx in (-1, 1) for sy in (-1, 1))
            if far:
                assert hs["offroad"][4 * e] == 1
                seen += 1
            elif near:
                assert hs["offroad"][4 * e] == 0
    assert seen > 10


def _cached_world_worker(cache_dir, q):
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from torchdriveenv_amd.synth import synthetic_world
    from torchdriveenv_amd.world import cached_world

    w, how = cached_world("race-test", lambda: synthetic_world(n_scn=2, A=4, seed=3, n_maps=1), cache_dir)
    q.put((how, int(w.arrays["cell_word"].sum() % 1000003), w.ints["n_scn"]))


@pytest.mark.timeout(300)
def test_cached_world_is_built_once_by_racing_processes(tmp_path):
    """world.cached_world: N processes that ask for the same tables at once (the ranks of a multi-GPU job) - exactly one builds and
    saves them, the others wait for the file and load identical tables"""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cached_world_worker, args=(str(tmp_path), q)) for _ in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == ["built", "loaded", "loaded"]
    assert len({r[1:] for r in res}) == 1


def _line_across(p, q, length=2.0, width=9.0, light=0):
    """a stop line centred at q across the travel direction p -> q: (x, y, psi, length, width, light)"""
    return (q[0], q[1], math.atan2(q[1] - p[1], q[0] - p[0]), length, width, light)


def test_traffic_lights_hook_groups_renumbers_and_shares(tmp_path):
    """world_from_waypoint_suite(traffic_lights=...) (ref gym_env.py:181-189: stop lines + light controller of the map config):
    a small location becomes ONE light group per map, a location with more than 32 lights is cut into per-scenario neighbourhoods
    of at most 32 lights re-numbered from 0, identical neighbourhoods share a descriptor, red masks follow the re-numbering, and
    traffic_lights_from_controller builds the description from stop-line objects + a controller's state sequence"""
    from torchdriveenv_amd.env import MAX_GROUP_LIGHTS, traffic_lights_from_controller, world_from_waypoint_suite

    # "Big": 40 lights on a 2 km street, one stop line each, every second one red in phase 0, the others in phase 1
    big_lines = [(50.0 * i, 0.0, 0.0, 2.0, 9.0, i) for i in range(40)]
    big = dict(stoplines=big_lines, phases=[(20, list(range(0, 40, 2))), (30, list(range(1, 40, 2)))])
    small = traffic_lights_from_controller(
        stoplines=[dict(actor_id=7, agent_type="traffic_light", x=30.0, y=100.0, orientation=0.0, length=2.0, width=9.0),
                   dict(actor_id=9, agent_type="traffic_light", x=60.0, y=100.0, orientation=0.0, length=2.0, width=9.0),
                   dict(actor_id=11, agent_type="stop_sign", x=90.0, y=100.0, orientation=0.0, length=2.0, width=9.0)],
        light_states=[{7: "red", 9: "green"}, {7: "green", 9: "yellow"}, {7: "green", 9: "red"}], durations=[3.0, 0.5, 2.04])
    assert small["actor_ids"] == [7, 9] and len(small["stoplines"]) == 2 and small["phases"] == [(30, [0]), (5, []), (20, [1])]
    wps = lambda x0, y0: [[x0 + 14.0 * k, y0] for k in range(6)]                     # noqa: E731
    data = WaypointSuite(locations=["Big", "Big", "Big", "Small", "Small", "None"],
                         waypoint_suite=[wps(0, 0), wps(5, 0), wps(1500, 0), wps(0, 100), wps(10, 100), wps(0, 300)],
                         car_sequence_suite=[None] * 6, scenarios=[None] * 6)
    lights = {"Big": big, "Small": small}
    world = world_from_waypoint_suite(data, agents_per_env=4, traffic_lights=lights, light_radius=400.0)
    maps, scn = world.arrays["maps"], world.arrays["scn"]
    n_mesh = 6                                                                        # no road_meshes: a corridor mesh per scenario
    assert world.has_lights and world.ints["n_maps"] == n_mesh + 5                    # 3 Big neighbourhoods (on 3 meshes) + 2 Small
    assert scn["map"][5] == 5                                                         # "None": no lights, its own mesh descriptor
    for si in range(5):
        d = maps[scn["map"][si]]
        assert scn["map"][si] >= n_mesh and d["cycle_steps"] == (50 if si < 3 else 55) and d["cell_base"] == maps[si]["cell_base"]
    st = world.arrays["stoplines"]
    for si, x0 in ((0, 0.0), (1, 5.0), (2, 1500.0)):
        d = maps[scn["map"][si]]
        mine = st[d["stop_base"]:d["stop_base"] + d["n_stop"]]
        assert 1 <= d["n_stop"] <= MAX_GROUP_LIGHTS and mine["light"].max() == d["n_stop"] - 1 and mine["light"].min() == 0
        assert (np.abs(mine["x"] - (x0 + 35.0)).max() <= 400.0 + 35.0 + 1e-3)          # within reach of the scenario's waypoints
        ph = world.arrays["phases"][d["phase_base"]:d["phase_base"] + d["n_phase"]]
        assert list(ph["end_step"]) == [20, 50]
        # the re-numbered masks: local light n is global light round(x / 50) - red in phase 0 iff that is even
        glob = np.round(mine["x"] / 50.0).astype(int)
        want0 = sum(1 << int(n) for n, g in zip(mine["light"], glob) if g % 2 == 0)
        assert ph["red_mask"][0] == want0 and ph["red_mask"][1] == ((1 << int(d["n_stop"])) - 1) ^ want0
    small_d = [maps[scn["map"][si]] for si in (3, 4)]
    assert all(d["n_stop"] == 2 and d["n_phase"] == 3 for d in small_d)
    # a per-location mesh: the two Small scenarios then share ONE mesh and ONE light group
    mesh = {"Small": np.asarray([[[-20, 90], [120, 90], [120, 110]], [[-20, 90], [120, 110], [-20, 110]]], np.float64)}
    w2 = world_from_waypoint_suite(WaypointSuite(locations=["Small", "Small"], waypoint_suite=[wps(0, 100), wps(10, 100)],
                                                 car_sequence_suite=[None, None], scenarios=[None, None]),
                                   agents_per_env=4, traffic_lights=lambda loc: lights.get(loc), road_meshes=mesh)
    assert w2.ints["n_maps"] == 2 and list(w2.arrays["scn"]["map"]) == [1, 1] and w2.arrays["maps"]["n_stop"][1] == 2
    with pytest.raises(ValueError):
        world_from_waypoint_suite(data, agents_per_env=4, traffic_lights={"Big": dict(stoplines=[(0, 0, 0, 2, 9, 0)], phases=[])})


def _normal_f32(ra, rb):
    """numpy restatement of the reset's Gaussian (oracle tde_normal / device normal_f32): Box-Muller in fp32 on the shared log / sincos"""
    from oracle import oracle

    f = np.float32
    u1 = (f(ra >> 8) + f(0.5)) * f(1.0 / 16777216.0)
    u2 = f(rb >> 8) * f(1.0 / 16777216.0)
    s, c = oracle.sincosf(np.asarray([f(6.28318530717958647692) * u2], np.float32))
    lg = f(oracle.lib().tde_oracle_logf(float(u1)))
    return f(np.sqrt(f(-2.0) * lg)) * f(c[0])


def test_start_headings_hook_fills_the_table_and_the_reset_reads_it():
    """world_from_waypoint_suite(start_headings=...) (ref gym_env.py:357-361: start_orientation = find_lanelet_directions(lanelet_map,
    x, y)[0] at the drawn start point + normal(0, 0.1)): the field is sampled at NH points along every scenario's first segment; an
    episode that starts at fraction f reads entry floor(f * NH).  The oracle's reset against a numpy restatement of the draw."""
    import ctypes as C

    from oracle import oracle
    from torchdriveenv_amd.env import world_from_waypoint_suite
    from torchdriveenv_amd.state import EnvState

    oracle.lib().tde_oracle_logf.restype = C.c_float
    oracle.lib().tde_oracle_logf.argtypes = [C.c_float]
    field = lambda loc, x, y: (0.3 if loc == "A" else -1.1) + 0.01 * x - 0.02 * y          # noqa: E731
    data = WaypointSuite(locations=["A", "B", "A"], waypoint_suite=[[[0, 0], [14, 3], [28, 3]], [[5, 5], [5, 19], [5, 33]], [[-3, 2], [9, -6], [20, -6]]],
                         car_sequence_suite=[None] * 3, scenarios=[None] * 3)
    NH = 8
    world = world_from_waypoint_suite(data, agents_per_env=2, start_headings=field, heading_samples=NH)
    assert world.ints["NH"] == NH and world.arrays["start_psi"].shape == (3, NH)
    for si, loc in enumerate(data.locations):
        p0, p1 = np.asarray(data.waypoint_suite[si][0], float), np.asarray(data.waypoint_suite[si][1], float)
        want = [field(loc, *(p0 + (j + 0.5) / NH * (p1 - p0))) for j in range(NH)]
        assert np.array_equal(world.arrays["start_psi"][si], np.asarray(want, np.float32))
    # the dict form, with one location left to its segment direction
    w2 = world_from_waypoint_suite(data, agents_per_env=2, start_headings={"A": lambda x, y: 0.25}, heading_samples=4)
    seg_b = math.atan2(19 - 5, 0.0)
    assert np.array_equal(w2.arrays["start_psi"], np.asarray([[0.25] * 4, [seg_b] * 4, [0.25] * 4], np.float32))
    assert world_from_waypoint_suite(data, agents_per_env=2).ints["NH"] == 0
    # reset: every env's start pose from the table
    cfg = _abi.default_config(seed=77)
    B = 256
    hs = EnvState(B, 2)
    oracle.env_reset(cfg, world, hs)
    oracle.env_reset(cfg, world, hs)                              # (second episode: the counters move)
    seen = set()
    for e in range(B):
        r0 = oracle.philox(77, e, 1, 0, 0x7DE)
        r1 = oracle.philox(77, e, 1, 1, 0x7DE)
        scn = (int(r0[0]) * 3) >> 32
        assert scn == hs["scn"][e]
        f = (int(r0[1]) >> 8) / 16777216.0
        idx = int(f * NH)
        seen.add((scn, idx))
        wp = np.asarray(data.waypoint_suite[scn], np.float64)
        assert hs["x"][2 * e] == np.float32(wp[0, 0] + f * (wp[1, 0] - wp[0, 0])) and hs["y"][2 * e] == np.float32(wp[0, 1] + f * (wp[1, 1] - wp[0, 1]))
        want = np.float32(float(world.arrays["start_psi"][scn, idx]) + float(_normal_f32(int(r1[2]), int(r1[3]))) * 0.1)
        assert hs["psi"][2 * e] == want, (e, scn, idx)
    assert len(seen) > 15                                          # the table's entries were really exercised
