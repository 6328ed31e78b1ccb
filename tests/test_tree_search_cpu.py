"""Growmap search (sequoia_amd/tree_search.py) against the reference's own tree_search.py outputs
(tests/golden/tree_search.json, made by oracle/gen_tree_search_golden.py) and the shipped demo tree."""
import json
import os

import numpy as np
import pytest

from sequoia_amd import tree_search as ts
from sequoia_amd.growmap import GrowMap

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "tree_search.json")
with open(GOLDEN) as f:
    CASES = json.load(f)


def _config(case):
    cfg = {k: v for k, v in case["config"].items() if k != "p"}
    cfg["acceptance_rate_vector"] = case["config"]["p"]
    return cfg


@pytest.mark.parametrize("name", sorted(CASES))
def test_search_reproduces_reference_growmap(name):
    case = CASES[name]
    g, report = ts.search(_config(case))
    exp = case["expect"]
    assert [report["budget"], report["depth"]] == exp["pair"]
    assert report["time_per_token"] == pytest.approx(exp["dec_time"], rel=0, abs=0)      # same float32 quotient
    assert g["size"] == exp["size"]
    assert g["Successors"] == exp["Successors"]
    assert g["roots"] == exp["roots"] and g["branches"] == exp["branches"]
    assert g["depth"].tolist() == exp["depth"]
    assert g["mask"].sum(1).tolist() == exp["mask_rowsum"]


@pytest.mark.parametrize("name", sorted(CASES))
def test_value_table_bit_exact(name):
    case = CASES[name]
    cfg = _config(case)
    p = np.asarray(cfg["acceptance_rate_vector"], dtype=np.float32)[:-1]
    tab = ts.search_tables(p, cfg["max_budget"], cfg["max_depth"])
    exp = np.array([[-np.inf if x is None else x for x in row] for row in case["expect"]["results"]], dtype=np.float32)
    assert tab.best.shape == exp.shape
    assert np.array_equal(tab.best, exp)


def test_demo_config_gives_the_shipped_demo_tree():
    """demo-config.json + acceptance-rate-vector.pt -> demo_tree.pt (all three ship with the reference;
    the tree is in-tree as growmaps/demo_tree.json)."""
    g, _ = ts.search(_config(CASES["demo"]))
    assert g["Successors"] == GrowMap.load("demo_tree").successors


def test_growmap_loads_into_the_engine_format(tmp_path):
    g, _ = ts.search(_config(CASES["flat_verify_128"]))
    path = str(tmp_path / "tree.json")
    ts.save_growmap(g, path)
    gm = GrowMap.load(path)
    assert gm.size == 128
    ref = gm.to_reference_dict()
    assert ref["roots"] == g["roots"] and ref["branches"] == g["branches"]
    assert np.array_equal(np.asarray(ref["mask"]), g["mask"])
    assert np.asarray(ref["depth"]).tolist() == g["depth"].tolist()
    ts.save_growmap(g, str(tmp_path / "tree.pt"))
    import torch
    back = torch.load(str(tmp_path / "tree.pt"), weights_only=False)
    assert back["Successors"] == g["Successors"] and back["size"] == 128 and back["mask"].dtype == torch.int64


def test_expected_length_is_monotone_in_budget_and_depth():
    p = np.asarray(CASES["demo"]["config"]["p"], dtype=np.float32)[:-1]
    tab = ts.search_tables(p, 64, 8)
    best = tab.best[1:, 1:]
    with np.errstate(invalid="ignore"):
        assert (np.diff(best, axis=1) >= 0)[np.isfinite(best[:, 1:]) & np.isfinite(best[:, :-1])].all()
    full = tab.best[1:, 8]                       # depth 8 admits every budget up to a chain of 8, then wider trees
    assert np.isfinite(full).all() and (np.diff(full) >= -1e-6).all()
