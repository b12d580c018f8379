"""The CPU oracle's reference-owned logic vs golden vectors captured from the reference's gym_env.py
(WaypointSuiteEnv.step / get_reward / check_reach_target / is_terminated / is_truncated / get_info)."""
import numpy as np
import pytest

from oracle import oracle
from tests.golden_util import case_config, case_expected, case_inputs


def test_golden_meta(golden):
    assert golden["meta"]["n_cases"] == len(golden["cases"]) >= 16
    assert golden["meta"]["n_steps"] >= 1000


def test_oracle_reward_operator_matches_reference_bit_exact(golden):
    n = 0
    for case in golden["cases"]:
        cfg = case_config(case)
        i = case_inputs(case)
        e = case_expected(case)
        steps, target, reached = i["steps"].copy(), i["target"].copy(), i["reached"].copy()
        out = oracle.waypoint_reward(cfg, i["pre"], i["post"], i["off"], i["col"], i["tl"], i["wp"], i["wp_n"],
                                     i["scn"], steps, target, reached)
        name = case["name"]
        # reward: the reference returns r.item() of an fp32 tensor -> exactly representable; bit-exact
        assert np.array_equal(out["reward"].astype(np.float64), e["reward"]), name
        assert np.array_equal(out["terminated"], e["terminated"]), name
        assert np.array_equal(out["truncated"], e["truncated"]), name
        assert np.array_equal(target, e["target_after"]), name
        assert np.array_equal(reached, e["reached"]), name
        assert np.array_equal(out["info_reached"], e["reached"]), name
        assert np.array_equal(out["info"], e["info"]), name           # float64 info terms, bit-exact
        assert np.array_equal(steps, i["steps"] + 1), name
        assert np.array_equal(out["truncated"], e["is_success"]), name  # is_success == truncation test (:430)
        n += len(steps)
    assert n == golden["meta"]["n_steps"]


def test_oracle_sequential_episode_matches_reference(golden):
    """carry the counters step by step (instead of taking them from the record): whole-episode behaviour"""
    for case in golden["cases"]:
        cfg = case_config(case)
        i = case_inputs(case)
        e = case_expected(case)
        steps = np.zeros(1, np.int32)
        target = np.ones(1, np.int32)        # current_target_idx = 1 at reset (gym_env.py:325)
        reached = np.zeros(1, np.int32)
        for t in range(len(i["pre"])):
            out = oracle.waypoint_reward(cfg, i["pre"][t:t + 1], i["post"][t:t + 1], i["off"][t:t + 1],
                                         i["col"][t:t + 1], i["tl"][t:t + 1], i["wp"], i["wp_n"], i["scn"][:1],
                                         steps, target, reached)
            assert float(out["reward"][0]) == e["reward"][t], (case["name"], t)
            assert target[0] == e["target_after"][t] and reached[0] == e["reached"][t], (case["name"], t)
            assert out["terminated"][0] == e["terminated"][t] and out["truncated"][0] == e["truncated"][t]
