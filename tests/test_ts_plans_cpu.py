"""Launch plans of the tall-skinny projections, host logic only (no GPU): every projection shape of the bench
configurations has a shipped plan, shipped plans respect the kernel's per-row-count register budget, and the fallback plan
for unmeasured shapes has the form the measured ones have."""
import json
import os

import pytest

from sequoia_amd.Engine.Llama_model import KNOWN_ARCHS
from sequoia_amd.Engine.ts_linear import MAX_SPLITS, SPLITS_CAP, SPLITTABLE, TsLinearSet, plan_key, shipped_plans
from sequoia_amd.growmap import GrowMap
from sequoia_amd.harness import MODELS


def _shapes(arch, vocab=32000):
    c = KNOWN_ARCHS[arch]
    h = c["hidden_size"]
    d = h // c["num_attention_heads"]
    return {"qkv": ((c["num_attention_heads"] + 2 * c["num_key_value_heads"]) * d, h, False), "o": (h, h, False),
            "gate_up": (c["intermediate_size"], h, True), "down": (h, c["intermediate_size"], False), "lm_head": (vocab, h, False)}


class _Set:      # the two attributes default_plan reads
    def __init__(self, shapes):
        self.shapes = shapes


def test_every_bench_configuration_has_shipped_plans():
    """Round 5 found configuration E at TP = 1 running three projections on the fallback plan -- a fifth of its step.  Every
    (model, row count) the bench configurations launch must be in ts_plans_gfx950.json."""
    plans = shipped_plans()
    missing = []
    growmaps = {c: [m["growmap"]] for c, m in MODELS.items()}
    growmaps["B"].append("MI355X-synthetic-68m-7b-stochastic")
    growmaps["D"].append("MI355X-synthetic-1.3b-13b-stochastic")
    for cfg, m in MODELS.items():
        for gname in growmaps[cfg]:
            g = GrowMap.load(gname)
            for arch, rows in ((m["draft"], sorted({lv.total for lv in g.levels} | {1})), (m["target"], [g.size])):
                for q in rows:
                    if q > 144:
                        continue
                    for name, (n_out, k, silu) in _shapes(arch).items():
                        key = plan_key(n_out, k, silu, (q + 15) // 16)
                        if key not in plans:
                            missing.append((cfg, gname, arch, q, name, key))
    assert not missing, missing


def test_shipped_plans_fit_the_kernel():
    for key, rec in shipped_plans().items():
        if rec == "torch":
            continue
        tiles, splits = int(rec[0]), int(rec[1])
        shape, mtp = key.split("@")
        mtp = int(mtp)
        silu = shape.endswith("s")
        n_out, k = (int(x) for x in shape.rstrip("s").split("x"))
        assert 1 <= splits <= MAX_SPLITS and k // 32 >= splits * 4, key
        units = (2 * n_out if silu and splits > 1 else n_out) // 16
        per = -(-units // min(tiles, units))
        if silu and splits == 1:
            assert per <= (4 if mtp <= 9 else 3), key           # (@9 plans with 4 units are clamped to 3 beyond 129 rows at launch)
        else:
            assert per <= 8, key


@pytest.mark.parametrize("arch", sorted(KNOWN_ARCHS))
@pytest.mark.parametrize("q_len", [1, 34, 64, 128, 129, 144])
def test_default_plan_is_launchable_and_splits_long_k(arch, q_len):
    ts = _Set(_shapes(arch))
    for name in ("qkv", "o", "gate_up", "down"):
        n_out, k, silu = ts.shapes[name]
        tiles, splits = TsLinearSet.default_plan(ts, name, q_len)
        units = n_out // 16
        per = -(-units // tiles)
        assert 1 <= tiles <= units and 1 <= splits <= SPLITS_CAP[name] and k // 32 >= splits * 8
        assert per <= ((3 if q_len > 128 else 4) if silu else 6), (arch, name, q_len, tiles, splits)
        if not silu and name in SPLITTABLE and units >= 256:
            # wide layers: a workgroup must not pull the whole activation image -- K is split until ~256 workgroups exist
            assert tiles * splits >= 192 or splits == SPLITS_CAP[name], (arch, name, tiles, splits)


def test_plans_file_is_well_formed():
    from sequoia_amd.Engine.ts_linear import PLAN_FILE
    with open(PLAN_FILE) as f:
        d = json.load(f)
    assert "plans" in d and len(d["plans"]) > 100
    for key, rec in d["plans"].items():
        assert rec == "torch" or (isinstance(rec, list) and len(rec) == 2), key
