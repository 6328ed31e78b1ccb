"""The native testbed entry point (sequoia_amd/testbed.py = tests/testbed.py's flags and loops) end to end."""
import pytest

pytestmark = pytest.mark.gpu

COMMON = ["--model", "random:JackFram/llama-68m:seed=1", "--target", "random:JackFram/llama-68m:seed=2", "--M", "384",
          "--start", "0", "--end", "1"]


@pytest.mark.parametrize("tree,growmap", [("sequoia", "demo_tree"), ("greedy", "8x8-tree"), ("specinfer", "8x8-tree"),
                                          ("greedys", "2-chain")])
def test_fast_mode_runs_every_tree(tree, growmap, capsys):
    from sequoia_amd import testbed
    avg = testbed.main(COMMON + ["--Mode", "fast", "--tree", tree, "--growmap", growmap])
    out = capsys.readouterr().out
    assert "decoding step:" in out and "large model step:" in out
    assert avg >= 1.0          # every step commits at least the bonus token


def test_benchmark_and_baseline_modes(capsys):
    from sequoia_amd import testbed
    testbed.main(COMMON + ["--Mode", "benchmark", "--growmap", "demo_tree"])
    assert "large model run:" in capsys.readouterr().out
    res = testbed.main(COMMON + ["--Mode", "baseline"])
    assert res["tokens"] >= 1 and res["ms_per_token"] > 0
