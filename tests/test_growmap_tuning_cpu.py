"""Host logic of the growmap tuner and the autoregressive baseline loop, on CPU with tiny models and the
oracle ops standing in for the HIP library (test infrastructure; the product path refuses CPU tensors)."""
import numpy as np
import pytest
import torch

from conftest import load_trace
from helpers import build_engines


@pytest.fixture
def oracle_ops():
    from oracle.ops_adapter import OracleOps
    from sequoia_amd import ops
    ops.set_ops_for_testing(OracleOps())
    yield
    ops.set_ops_for_testing(None)


def _engines(name="B_seq128"):
    z, meta = load_trace(name)
    draft, target = build_engines(z, meta, "cpu")
    prompts = [[int(t) for t in z["prompt"]]]
    return draft, target, meta, prompts


def test_acceptance_vector_layout_and_mass(oracle_ops):
    from sequoia_amd import growmap_tuning as gt
    draft, target, meta, prompts = _engines()
    cfg = dict(mode="stochastic", M=meta["M"])
    torch.manual_seed(3)
    vec = gt.measure_acceptance_vector(cfg, draft, target, "cpu", prompts, width=6, steps=12, T=meta["T"],
                                       use_graphs=False, vocab=meta["vocab"])
    assert vec.shape == (8,) and vec.dtype == np.float32
    assert vec[0] == 0 and abs(vec[1:].sum() - 1.0) < 1e-6 and (vec >= 0).all()


def test_identical_models_accept_the_first_child(oracle_ops):
    """draft == target: p == q, so `p[tok] > r * q[tok]` holds for the first child whenever r < 1."""
    from sequoia_amd import growmap_tuning as gt
    from sequoia_amd.Engine.Engine import GraphInferenceEngine, GraphInferenceEngineTG
    from helpers import dims_dict, state_dict_of
    z, meta = load_trace("B_seq128")
    spec = dict(state_dict=state_dict_of(z, "draft"), config=dims_dict(meta["draft_dims"], meta["vocab"]))
    M = meta["M"]
    draft = GraphInferenceEngine(max_length=M, model_name_or_path=spec, dtype=torch.float16, device="cpu")
    target = GraphInferenceEngineTG(max_length=M, model_name_or_path=spec, dtype=torch.float16, device="cpu")
    torch.manual_seed(5)
    vec = gt.measure_acceptance_vector(dict(mode="stochastic", M=M), draft, target, "cpu", [[int(t) for t in z["prompt"]]],
                                       width=4, steps=8, T=meta["T"], use_graphs=False, vocab=meta["vocab"])
    assert vec[1] >= 0.75          # fp16 GEMM-order noise between the q=1 and q=4 forwards may flip a rare tie


def test_star_growmap_shape():
    from sequoia_amd.growmap_tuning import star_growmap
    g = star_growmap(5)
    assert g.size == 6 and g.successors[0] == [1, 2, 3, 4, 5] and all(not s for s in g.successors[1:])
    assert g.roots == [[0], [1, 2, 3, 4, 5]] and g.branches[0] == [5]


def test_autoregressive_loop_samples_from_the_target(oracle_ops):
    """Each step's token must be the sq_sample_wor(k=1) draw from the target's logits at that position:
    replay the loop by hand with the dense engine API and compare."""
    from sequoia_amd import harness
    from oracle import ops_np
    draft, target, meta, prompts = _engines()
    cfg = dict(mode="stochastic", M=meta["M"])
    torch.manual_seed(11)
    loop = harness.AutoregressiveLoop(cfg, target, "cpu", prompts, T=meta["T"], max_steps=5, use_graphs=False,
                                      vocab=meta["vocab"])
    toks = []
    orig = loop.ops.sample_wor

    def spy(logits, rand, row_ids, k, T, out, **kw):
        r = orig(logits, rand, row_ids, k, T, out, **kw)
        toks.append((int(out[0]), logits.float().numpy().copy(), rand.float().numpy().copy()))
        return r
    loop.ops.sample_wor = spy
    dt, done = loop.run_prompt()
    assert done == len(toks) and 1 <= done <= 5
    # independent dense-mask forward of the whole sequence: logits at every position must match the loop's
    n0 = len(prompts[0][:128])
    seq = prompts[0][:128] + [t for t, _, _ in toks[:-1]]
    n = len(seq)
    mask = torch.full((n, n), torch.finfo(torch.float16).min, dtype=torch.float16).triu(1)[None, None]
    target.clear_kv()
    full = target.inference(input_ids=torch.tensor(seq)[None], storage_ids=torch.arange(n),
                            position_ids=torch.arange(n)[None], attn_mask=mask)[0].float().numpy()
    for i, (tok, lg, rnd) in enumerate(toks):
        assert np.abs(full[n0 - 1 + i] - lg[0]).max() < 4e-2
        exp = ops_np.sample_wor(lg.astype(np.float16), rnd.astype(np.float16), 1, meta["T"])
        assert tok == int(exp[0][0])


@pytest.mark.parametrize("mode", ["stochastic", "greedy"])
def test_accept_probe_classes_follow_the_reference_loop(oracle_ops, mode):
    """tests/test_accept.py's loop: a fresh SpecTreeTest / GreedyTreeTest per step with the KV lengths carried over,
    verify(benchmark=True) -> (valid_tokens, draft_kv_len, target_kv_len, b, terminate)."""
    from sequoia_amd.Tree.GreedyTree import GreedyTreeTest
    from sequoia_amd.Tree.SpecTree import SpecTreeTest
    draft, target, meta, prompts = _engines()
    M, w = meta["M"], 5
    cls = SpecTreeTest if mode == "stochastic" else GreedyTreeTest
    attn_mask = torch.full((M, M), torch.finfo(torch.float16).min, dtype=torch.float16)
    position_ids = torch.zeros(M).long()
    input_ids = torch.tensor(prompts[0][:16]).unsqueeze(0)
    draft_kv_len = target_kv_len = 0
    counts = np.zeros(w + 1)
    torch.manual_seed(2)
    for step in range(5):
        tree = cls(prefix=input_ids.squeeze(0), device="cpu", temperature=meta["T"], top_p=1.0, draft_kv_len=draft_kv_len,
                   target_kv_len=target_kv_len, draft_model_engine=draft, target_model_engine=target, max_length=M,
                   attn_mask=attn_mask, sequence=None, new_tokens_buffer=None, parents_buffer=None,
                   position_ids=position_ids, max_width=w)
        assert tree.Successors[0] == list(range(1, w + 1))
        valid, draft_kv_len, target_kv_len, b, terminate = tree.verify(benchmark=True)
        assert -1 <= b < w
        counts[b] += 1
        grew = valid.shape[0] - input_ids.shape[1]
        assert grew == (1 if b < 0 else 2) or terminate
        assert draft_kv_len == target_kv_len == valid.shape[0] - (0 if terminate else 1)
        if b >= 0:      # the accepted child is the b-th drawn token
            assert int(valid[input_ids.shape[1]]) == int(tree.tokens[input_ids.shape[1]])
        input_ids = valid.clone().unsqueeze(0)
        if terminate:
            break
    assert counts.sum() >= 1
    draft.clear_kv(); target.clear_kv()
