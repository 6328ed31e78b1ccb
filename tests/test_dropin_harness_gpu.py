"""The reference harness body (tests/testbed.py:45-95 `simulation_fast` and its setup :250-285),
re-typed here against the reference's import paths, runs unchanged on top of
sequoia_amd.dropin: engine construction, cuda_graph_for_* factories, initialize_cuda_graph,
SpecTree keyword arguments, construct_grow_map()/verify() loop, clear_kv()."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_simulation_fast_body_runs_on_dropin_modules():
    import sequoia_amd.dropin as dropin
    dropin.install(force=True)
    try:
        from Engine.Engine import GraphInferenceEngine, GraphInferenceEngineTG
        from Tree.SpecTree import SpecTree
        from utils import cuda_graph_for_residual, cuda_graph_for_sampling_without_replacement
        from sequoia_amd.growmap import GrowMap

        M, T, P = 384, 0.6, 1.0
        draft_model = GraphInferenceEngine(max_length=M, model_name_or_path="random:JackFram/llama-68m:seed=1:gain=30",
                                           dtype=torch.float16, device="cuda:0")
        target_model = GraphInferenceEngineTG(max_length=M, model_name_or_path="random:JackFram/llama-160m:seed=2:gain=30",
                                              dtype=torch.float16, device="cuda:0")
        residual_graph = cuda_graph_for_residual()
        grow_map = GrowMap.load("A100-CNN-160m-13b-stochastic").to_reference_dict()   # what torch.load(path) returns
        tree_size = grow_map["size"]
        idx_lists, branch_lists = grow_map["roots"], grow_map["branches"]
        draft_step = len(grow_map["roots"])
        graph_capture_list = [sum(x) for x in branch_lists]
        graph_capture_list.append(1)
        draft_model.initialize_cuda_graph(graph_capture_list)
        sampling_callables, sample_gather_indices = {}, {}
        for i in range(draft_step - 1):
            sampling_callables[i] = cuda_graph_for_sampling_without_replacement(
                max_length=M, idx_len=len(idx_lists[i]), num_samples=max(branch_lists[i]), temperature=T,
                tree_size=tree_size)
        for i in range(draft_step - 1):
            ith = []
            max_num_samples = max(branch_lists[i])
            for j, branch in enumerate(branch_lists[i]):
                ith.append(torch.arange(branch, device="cuda:0", dtype=torch.long) + j * max_num_samples)
            sample_gather_indices[i] = torch.cat(ith)

        # ---- simulation_fast ------------------------------------------------------------------
        dtype = torch.float16
        attn_mask = torch.full((M, M), torch.finfo(dtype).min, dtype=dtype, device="cuda:0")
        sequence = torch.tensor(list(range(M)), device="cuda:0").long().unsqueeze(-1)
        new_tokens_buffer = torch.zeros(M).long().to("cuda:0")
        parents_buffer = torch.zeros(M).long().to("cuda:0")
        position_ids = torch.zeros(M).long().to("cuda:0")
        prompts = json.load(open(os.path.join(REPO, "sequoia_amd", "growmaps", "c4_small_prompts.json")))["prompts"]
        num_decoding_steps = num_large_model_steps = 0
        with torch.no_grad():
            for step in range(2):
                input_ids = torch.tensor(prompts[step][:128]).unsqueeze(0)
                terminate = False
                attn_mask.fill_(torch.finfo(dtype).min)
                spectree = SpecTree(prefix=input_ids.squeeze(0), device="cuda:0", temperature=T, top_p=P,
                                    draft_kv_len=0, target_kv_len=0, draft_model_engine=draft_model,
                                    target_model_engine=target_model, max_length=M, max_target_seq=M,
                                    grow_map=grow_map, attn_mask=attn_mask, sequence=sequence,
                                    new_tokens_buffer=new_tokens_buffer, parents_buffer=parents_buffer,
                                    position_ids=position_ids, residual_graph=residual_graph,
                                    sampling_callables=sampling_callables, sample_gather_indices=sample_gather_indices)
                torch.cuda.synchronize()
                while input_ids.shape[1] < 256 and terminate is False:
                    spectree.construct_grow_map()
                    valid_tokens, draft_kv_len, target_kv_len, terminate = spectree.verify()
                    assert valid_tokens.shape[0] > input_ids.shape[1]
                    # committed text is append-only
                    assert torch.equal(valid_tokens[:input_ids.shape[1]].cpu(), input_ids[0].cpu())
                    num_decoding_steps += valid_tokens.shape[0] - input_ids.shape[1]
                    num_large_model_steps += 1
                    input_ids = valid_tokens.unsqueeze(0)
                    if (input_ids[0][-1] == 2) or (input_ids[0][-1] == 0):
                        terminate = True
                torch.cuda.synchronize()
                draft_model.clear_kv()
                target_model.clear_kv()
        assert num_large_model_steps > 0 and num_decoding_steps >= num_large_model_steps
        # the lengths the harness asked for were captured as implicit-mask hipGraphs and used
        # (a small draft captures every multi-row length twice: once generic, once for forwards whose rows never see each
        # other -- one tree level -- which run the fused attention block, Engine/ts_linear.py::attn_block_ok)
        lengths = set(graph_capture_list) - {0}
        assert {k[0] for k in draft_model.tree_callables} == lengths
        assert {k[0] for k in draft_model.tree_callables if len(k) == 3} == lengths - {1}
        # benchmark=True keeps the reference's 7-tuple
        spectree = SpecTree(prefix=torch.tensor(prompts[3][:128]), device="cuda:0", temperature=T, top_p=P,
                            draft_kv_len=0, target_kv_len=0, draft_model_engine=draft_model,
                            target_model_engine=target_model, max_length=M, max_target_seq=M, grow_map=grow_map,
                            attn_mask=attn_mask, sequence=sequence, new_tokens_buffer=new_tokens_buffer,
                            parents_buffer=parents_buffer, position_ids=position_ids, residual_graph=residual_graph,
                            sampling_callables=sampling_callables, sample_gather_indices=sample_gather_indices)
        a, b = spectree.construct_grow_map(benchmark=True)
        out = spectree.verify(benchmark=True)
        assert len(out) == 7 and a >= 0 and b >= 0
        # nucleus filtering (top_p < 1, utils.get_sampling_logits) runs through the native filter kernel
        draft_model.clear_kv(); target_model.clear_kv()
        st = SpecTree(prefix=torch.tensor(prompts[3][:128]), device="cuda:0", temperature=T, top_p=0.9,
                      draft_kv_len=0, target_kv_len=0, draft_model_engine=draft_model,
                      target_model_engine=target_model, max_length=M, max_target_seq=M, grow_map=grow_map,
                      attn_mask=attn_mask, sequence=sequence, new_tokens_buffer=new_tokens_buffer,
                      parents_buffer=parents_buffer, position_ids=position_ids, residual_graph=residual_graph,
                      sampling_callables=sampling_callables, sample_gather_indices=sample_gather_indices)
        st.construct_grow_map()
        vt, a1, _, _ = st.verify()
        assert vt.shape[0] > 128
        assert torch.isinf(st.target_logits).any()       # filtered logits carry -inf
    finally:
        dropin.uninstall()


def test_utils_factories_match_oracle():
    import numpy as np
    from oracle import ops_np as O
    from sequoia_amd import utils as U
    rng = np.random.RandomState(0)
    V = 32000
    logits = torch.from_numpy((rng.randn(5, V) * 3).astype(np.float16)).cuda()
    rand = torch.from_numpy((rng.randint(0, 2048, size=(5, V)) / 2048.0).astype(np.float16)).cuda()
    fn = U.cuda_graph_for_sampling_without_replacement(idx_len=5, num_samples=7, temperature=0.6)
    got = fn(logits, rand).cpu().numpy().reshape(5, 7)
    want = O.sample_wor(logits.cpu().numpy(), rand.cpu().numpy(), 7, 0.6)
    assert (got != want).sum() <= 1
    fa = U.cuda_graph_for_sampling_argmax(idx_len=5, num_samples=4)
    assert np.array_equal(fa(logits).cpu().numpy().reshape(5, 4), O.topk_ids(logits.cpu().numpy(), 4))
    m = U._make_causal_mask((1, 9), torch.float16, "cuda:0").cpu().numpy()
    want_m = np.triu(np.full((9, 9), -65504.0, dtype=np.float16), 1)
    assert np.array_equal(m, want_m)
    p = torch.softmax(logits[0].float(), -1).half(); q = torch.softmax(logits[1].float(), -1).half()
    res = U.cuda_graph_for_residual()(p, q)
    assert abs(float(res.float().sum()) - 1.0) < 2e-2


def _explain_divergence(z, meta, j, spectree, snap, prev_len, got):
    """Verify call j handed the harness other tokens than the reference's run did.  Which decision parted them, and could
    another fp16 arithmetic legitimately take it the other way?  The logits of both models are fp16 numbers of magnitude up to
    ~60 here (ulp 0.03-0.06): two correct implementations differ by an ulp or two per logit, i.e. by d = 2 ulps / T in every
    log-probability.  Evaluated by the numpy oracle on THIS GPU's own logits of the step (tokens / draft rows as they were
    before the verifier ran).  Returns (kind, x, limit); the divergence is explained when x <= limit:
      ("boundary", x)   same accepted path, another bonus token: the reference's uniform lies x (probability mass of the
                        reference's own residual) from the interval of the token drawn here; limit 2 % of mass;
      ("decision", x)   the reference's accepted path exists in this step's draft tree and the two walks part at ONE accept
                        test p > r q with x = |ln p - ln(r q)|; limit 4 d (p and q each move by up to 2 d);
      ("sampler", x)    the reference accepted a token this step's tree never drafted for that parent: x = |ln key_ref - ln
                        key_min| of the sampling keys log(u) / q (key_min: the smallest key drafted); limit 4 d -- or the
                        token's q sits at the fp16 underflow boundary (key -inf on one side only): x = |ln q - ln 2^-25|;
      ("tie", 0)        an order torch leaves unspecified decided it: (a) an exact tie of fp16 sampling keys on the accepted path
                        (torch.topk; here the lowest token id first) -- also the -inf keys of a parent with fewer non-zero-
                        probability tokens than children --, after which the tied tokens sit on swapped sibling nodes with
                        subtrees of other shapes; (b) the nucleus cut inside a class of exactly equal target logits (the
                        reference's unstable CPU sort; here by token id) containing a drafted child's token;
      ("unexplained", why, 0)."""
    import numpy as np
    from helpers import cdf_interval_distance
    from oracle import ops_np as O
    from oracle import ref_harness as RH
    from sequoia_amd.native import SQ_RES_N_TREE, SQ_RESULT_INTS
    ref = RH.record_tokens(z, j)
    succ, gt, T = meta["successors"], snap["gt"], meta["T"]
    n = len(succ)
    p_idx = int(z["verify_prompt"][j])
    u = int(z["bonus_u24"][p_idx * RH.STEPS_PER_PROMPT + int(z["verify_step"][j])])
    if len(got) == len(ref) and np.array_equal(got[:-1], ref[:-1]):
        res = RH.record_residual(z, j, meta["vocab"])
        return ("boundary", cdf_interval_distance(res, int(got[-1]), u), 2e-2) if res is not None else ("unexplained", "no residual", 0)
    tokens_pre = snap["tokens"].cpu().numpy()
    draft = snap["draft"].cpu().numpy()
    target = spectree.target_logits.cpu().numpy()[:n]
    r16 = spectree.r.cpu().numpy()

    def ulps2(row):                       # two fp16 ulps of the row's largest finite logit, in log-probability units
        fin = np.abs(row[np.isfinite(row)].astype(np.float32))
        return 2.0 * 2.0 ** -10 * max(1.0, float(fin.max()) if fin.size else 1.0) / T
    ref_acc = [int(t) for t in ref[gt:len(ref) - 1]] if len(ref) > gt else []
    lr = spectree.last_result
    got_nodes = [int(x) - (gt - 1) for x in lr[SQ_RESULT_INTS:SQ_RESULT_INTS + int(lr[SQ_RES_N_TREE])]]
    # walk the tree along the reference's accepted tokens, replaying the accept tests on this GPU's logits
    node, depth = 0, 0
    key_tie = False          # the path passed a parent where the accepted token's sampling key ties EXACTLY with a sibling's
    fragile = None           # ... or lies within the two-ulp band of a sibling's: (log distance, limit) -- the siblings may be
    #                          drafted in the other order, i.e. sit on swapped nodes with subtrees of other shapes
    while True:
        kids = succ[node]
        want_tok = ref_acc[depth] if depth < len(ref_acc) else None
        want = next((c for c in kids if int(tokens_pre[gt - 1 + c]) == want_tok), None) if want_tok is not None else None
        have = got_nodes[depth] if depth < len(got_nodes) else None
        if want_tok is not None and want is None and want_tok == int(ref[-1]) and have is not None:
            # the reference's commit order stores the bonus token BEFORE it gathers the accepted tokens (Tree/SpecTree.py:222-224):
            # an accepted node that sat at slot a shows the BONUS id in the reference's text -- the token itself is not
            # recorded; follow this GPU's own node at that depth (the walks part further down, or at the bonus draw)
            want = have
        if want is not None and len(kids) > 1:
            rand_ = spectree.rand[node].cpu().numpy()[None]
            keys_ = O.sample_keys(draft[node][None], rand_, T)[0]
            kt = [keys_[int(tokens_pre[gt - 1 + c])] for c in kids]
            kw_ = float(keys_[int(tokens_pre[gt - 1 + want])])
            key_tie = key_tie or sum(1 for k_ in kt if k_ == keys_[int(tokens_pre[gt - 1 + want])]) > 1
            if np.isfinite(kw_) and kw_ < 0:
                near = [abs(np.log(-kw_) - np.log(-float(k_))) for c_, k_ in zip(kids, kt) if c_ != want and np.isfinite(k_) and k_ < 0]
                lim_ = 4 * ulps2(draft[node])
                if near and min(near) <= lim_ and fragile is None:
                    fragile = (float(min(near)), lim_)
        if want_tok is not None and want is None:
            if key_tie or not kids:
                # an exact tie of fp16 sampling keys further up the accepted path: torch.topk leaves the order inside the tie
                # unspecified (here: lowest token id first), the tied tokens sit on swapped sibling nodes, and the growmap
                # gives those nodes subtrees of other shapes -- the reference's next accepted token has no node here
                return ("tie", 0.0, 0.0) if key_tie else ("unexplained", "the reference accepted below a leaf of this tree", 0)
            if fragile is not None:
                return "sampler", fragile[0], fragile[1]
            rand = spectree.rand[node].cpu().numpy()[None]
            keys = O.sample_keys(draft[node][None], rand, T)[0].astype(np.float64)
            kmin = min(float(keys[int(tokens_pre[gt - 1 + c])]) for c in kids)
            d = ulps2(draft[node])
            if not np.isfinite(keys[want_tok]) and not np.isfinite(kmin):
                # fewer tokens with a non-zero fp16 probability than children to draw: the rest of the draw is an exact tie
                # among keys of -inf, whose order torch.topk leaves unspecified (here: lowest token id first)
                return "tie", 0.0, 0.0
            if np.isfinite(keys[want_tok]) and np.isfinite(kmin):
                return "sampler", abs(float(np.log(-keys[want_tok]) - np.log(-kmin))), 4 * d
            q32 = np.exp((draft[node].astype(np.float32) / np.float32(T)) - np.max(draft[node].astype(np.float32) / np.float32(T)))
            q32 = q32 / q32.sum()
            return "sampler", abs(float(np.log(max(float(q32[want_tok]), 1e-30)) - np.log(2.0 ** -25))), 4 * d
        if want == have:
            if want is None:
                return "unexplained", "same path, other tokens", 0
            node, depth = want, depth + 1
            continue
        # the walks part below `node`: replay its accept tests (Tree/SpecTree.py:136-157) up to the first child either side took
        raw = snap.get("target_raw")
        if raw is not None:
            # the nucleus cut (utils.py:65-77) inside a class of exactly equal logits: WHICH of the equal tokens stay is decided
            # by torch's unstable CPU sort in the reference, by token id here; a drafted child whose token
            # is in that class has p > 0 on one side and p = 0 on the other
            rw = raw[node].astype(np.float32)
            kept = np.isfinite(target[node])
            if kept.any() and (~kept).any():
                vmin = rw[kept].min()
                cls = rw == vmin
                toks_ = {int(tokens_pre[gt - 1 + c]) for c in (want, have) if c is not None}
                if (cls & ~kept).any() and any(cls[t_] for t_ in toks_):
                    return "tie", 0.0, 0.0
        if key_tie:
            return "tie", 0.0, 0.0
        if fragile is not None:
            return "sampler", fragile[0], fragile[1]
        if want is not None and have is not None:
            # both tokens were drafted for this parent: were they drafted in the other ORDER?  The accept tests run in child
            # order (Tree/SpecTree.py:136-157) and the first accepted child wins, so two sampling keys within the two-ulp
            # band may swap the siblings and with them the outcome
            rand_ = spectree.rand[node].cpu().numpy()[None]
            keys_ = O.sample_keys(draft[node][None], rand_, T)[0].astype(np.float64)
            kw, kh = float(keys_[int(tokens_pre[gt - 1 + want])]), float(keys_[int(tokens_pre[gt - 1 + have])])
            if np.isfinite(kw) and np.isfinite(kh) and kw < 0 and kh < 0:
                xk = abs(np.log(-kw) - np.log(-kh))
                if xk <= 4 * ulps2(draft[node]):
                    return "sampler", float(xk), 4 * ulps2(draft[node])
        p = O.scaled_softmax_f16(target[node][None], T)[0]
        row = draft[node].copy()
        d = ulps2(target[node]) + ulps2(draft[node])
        worst = None
        for c in kids:
            tok = int(tokens_pre[gt - 1 + c])
            q = O.scaled_softmax_f16(row[None], T)[0]
            pt, rq = float(p[tok]), float(np.float32(r16[gt - 1 + c]) * np.float32(q[tok]))
            ok = pt > rq
            if c == want or c == have:
                x = abs(np.log(max(pt, 1e-30)) - np.log(max(rq, 1e-30)))
                return "decision", float(x), 2 * d
            if ok:
                return "unexplained", "the oracle accepts a third child on this GPU's logits", 0
            p, _ = O.residual_f16(p, q)
            row[tok] = O.F16_MIN
        return "unexplained", "no child of the parting node was taken by either side", 0


def _replay_record_on_gpu(path, prove=False):
    """The reference's harness body (tests/testbed.py:45-95 + set-up :250-285) RE-TYPED on a record's inputs -- the file itself
    cannot travel to the GPU box; tests/test_reference_harness_cpu.py executes the reference's own lines on the drop-in --
    with seeded weights, prompts, noise and bonus uniforms pinned as oracle/ref_harness.py pins them.  Returns the harness's
    value and the log [(prompt, tokens)] of what every verify() handed to the loop."""
    import numpy as np
    from oracle import ref_harness as RH
    import sequoia_amd.dropin as dropin
    z, meta = RH.load_record(path)
    dropin.install(force=True)
    try:
        from Engine.Engine import GraphInferenceEngine, GraphInferenceEngineTG
        from Tree.SpecTree import SpecTree
        from utils import cuda_graph_for_residual, cuda_graph_for_sampling_without_replacement
        from sequoia_amd.growmap import GrowMap
        M, T, P = meta["M"], meta["T"], meta["top_p"]
        sd_d, sd_t, checks = RH.seeded_pair(meta["seed"], meta["gain"], meta["share"], meta["branch"])
        assert checks == meta["weight_checksums"]
        hidden, inter, layers, heads, kv = meta["dims"]
        cfg = dict(vocab_size=meta["vocab"], hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
                   num_attention_heads=heads, num_key_value_heads=kv, max_position_embeddings=2048)
        # ---- set-up block (tests/testbed.py:250-285) --------------------------------------------------------------
        draft_model = GraphInferenceEngine(max_length=M, model_name_or_path=dict(state_dict=sd_d, config=cfg), dtype=torch.float16,
                                           device="cuda:0")
        target_model = GraphInferenceEngineTG(max_length=M, model_name_or_path=dict(state_dict=sd_t, config=cfg),
                                              dtype=torch.float16, device="cuda:0")
        residual_graph = cuda_graph_for_residual()
        grow_map = GrowMap.from_successors(meta["successors"]).to_reference_dict()
        tree_size = grow_map["size"]
        idx_lists, branch_lists = grow_map["roots"], grow_map["branches"]
        draft_step = len(grow_map["roots"])
        graph_capture_list = [sum(x) for x in branch_lists]
        graph_capture_list.append(1)
        draft_model.initialize_cuda_graph(graph_capture_list)
        sampling_callables, sample_gather_indices = {}, {}
        for i in range(draft_step - 1):
            sampling_callables[i] = cuda_graph_for_sampling_without_replacement(
                max_length=M, idx_len=len(idx_lists[i]), num_samples=max(branch_lists[i]), temperature=T, tree_size=tree_size)
        for i in range(draft_step - 1):
            ith = []
            for j, branch in enumerate(branch_lists[i]):
                ith.append(torch.arange(branch, device="cuda:0", dtype=torch.long) + j * max(branch_lists[i]))
            sample_gather_indices[i] = torch.cat(ith)
        # ---- simulation_fast (tests/testbed.py:45-95) ---------------------------------------------------------------
        dtype = torch.float16
        attn_mask = torch.full((M, M), torch.finfo(dtype).min, dtype=dtype, device="cuda:0")
        sequence = torch.tensor(list(range(M)), device="cuda:0").long().unsqueeze(-1)
        new_tokens_buffer = torch.zeros(M).long().to("cuda:0")
        parents_buffer = torch.zeros(M).long().to("cuda:0")
        position_ids = torch.zeros(M).long().to("cuda:0")
        num_decoding_steps = num_large_model_steps = 0
        u24 = z["bonus_u24"]
        log = []
        with torch.no_grad():
            for step in range(meta["n_prompts"]):
                input_ids = torch.from_numpy(z[f"prompt{step}/input_ids"])[..., :128]
                labels = torch.from_numpy(z[f"prompt{step}/labels"])[..., :128]
                terminate = False
                if labels[0][-1] == -100:
                    terminate = True
                attn_mask.fill_(torch.finfo(dtype).min)
                torch.manual_seed(RH.noise_seed(meta["seed"], step))          # the pins of oracle/ref_harness.py's SpySpecTree
                spectree = SpecTree(prefix=input_ids.squeeze(0), device="cuda:0", temperature=T, top_p=P, draft_kv_len=0,
                                    target_kv_len=0, draft_model_engine=draft_model, target_model_engine=target_model,
                                    max_length=M, max_target_seq=M, grow_map=grow_map, attn_mask=attn_mask, sequence=sequence,
                                    new_tokens_buffer=new_tokens_buffer, parents_buffer=parents_buffer,
                                    position_ids=position_ids, residual_graph=residual_graph,
                                    sampling_callables=sampling_callables, sample_gather_indices=sample_gather_indices,
                                    bonus_uniforms=[int(x) for x in u24[step * RH.STEPS_PER_PROMPT:(step + 1) * RH.STEPS_PER_PROMPT]],
                                    commit_order="reference")
                torch.cuda.synchronize()
                while input_ids.shape[1] < 256 and terminate is False:
                    spectree.construct_grow_map()
                    if prove:
                        snap = dict(gt=spectree.ground_truth_len, tokens=spectree.tokens.clone(),
                                    draft=spectree.draft_logits[:tree_size].clone())
                        real_filter = spectree.ops.top_p_filter

                        def spy_filter(logits, top_p, temperature, snap=snap, real_filter=real_filter):
                            snap["target_raw"] = logits[-tree_size:].clone().cpu().numpy()      # before the in-place cut
                            return real_filter(logits, top_p, temperature)
                        spectree.ops.top_p_filter = spy_filter
                    valid_tokens, draft_kv_len, target_kv_len, terminate = spectree.verify()
                    log.append((step, valid_tokens.cpu().numpy().copy()))
                    if prove:
                        del spectree.ops.top_p_filter                  # (the instance attribute shadowing the method)
                        jj = len(log) - 1
                        if jj >= meta["n_verify"] or int(z["verify_prompt"][jj]) != step:
                            return z, meta, None, log, ("unexplained", "another number of verify calls")
                        if not np.array_equal(log[-1][1], RH.record_tokens(z, jj)):
                            return z, meta, None, log, (jj,) + _explain_divergence(z, meta, jj, spectree, snap, input_ids.shape[1], log[-1][1])
                    num_decoding_steps += valid_tokens.shape[0] - input_ids.shape[1]
                    num_large_model_steps += 1
                    input_ids = valid_tokens.unsqueeze(0)
                    if (input_ids[0][-1] == 2) or (input_ids[0][-1] == 0):
                        terminate = True
                torch.cuda.synchronize()
                draft_model.clear_kv()
                target_model.clear_kv()
        value = num_decoding_steps / max(num_large_model_steps, 1)
        return (z, meta, value, log, None) if prove else (z, meta, value, log)
    finally:
        dropin.uninstall()


@pytest.mark.parametrize("seed", [24, 25, 28])
def test_reference_harness_record_replays_on_gpu(seed):
    """tests/golden/harness_simulation_fast_<seed>.npz: runs of the reference's own `simulation_fast` source on the reference's
    classes (oracle/ref_harness.py) whose CPU drop-in run is token-identical; the HIP library must hand the harness loop the
    reference's tokens in every verify call too."""
    from oracle import ref_harness as RH
    z, meta, value, log = _replay_record_on_gpu(os.path.join(REPO, "tests", "golden", f"harness_simulation_fast_{seed}.npz"))
    assert RH.classify_run(z, meta, log) == ("identical", None)
    assert value == meta["value"]


@pytest.mark.parametrize("tree", ["4x8", "s128"])
def test_unscreened_reference_harness_seeds_on_gpu(tree):
    """20 runs of the reference's harness per tree size, seeds 100-119 TAKEN AS THEY COME (oracle/gen_harness_golden.py): how
    many does this GPU reproduce token for token over a whole run (two prompts decoded to 256 tokens, ~40 verify calls, a few
    thousand sampled decisions)?  A run that parts from the reference's must do so at ONE decision that another arithmetic may
    legitimately take the other way (_explain_divergence, evaluated by the oracle on this GPU's own logits):
      * a bonus draw whose uniform lies within 2 % of mass of the drawn token's interval under the REFERENCE's residual (a
        normalised difference of nearly equal fp16 probabilities: one ulp of either moves every boundary);
      * an accept test p > r q, or a draft token at the top-k cut of its parent's sampling keys, that two fp16 ulps on the
        logits involved (|logit| reaches 60 here: an ulp is 0.03-0.06, i.e. 5-10 % of a probability at T = 0.6) flip.
      * an exact tie among sampling keys of -inf (fewer non-zero-probability tokens than children to draw).
      * an order torch leaves unspecified: exact ties of sampling keys (torch.topk) and the nucleus cut inside a class of
        equal logits (the unstable CPU sort) -- or two sibling keys inside the two-ulp band: the siblings are then drafted in
        the other order, the accept tests run in child order, and the swapped nodes carry subtrees of other shapes.
    Runs that part any other way are listed as unexplained and bounded (2 of 20; 0 of 40 on the round's boxes).  The counts are printed
    in the session summary (tests/conftest.py)."""
    import glob
    import helpers
    paths = sorted(glob.glob(os.path.join(REPO, "tests", "golden", f"harness_unscreened_{tree}_*.npz")))
    assert len(paths) >= 20
    identical, explained, unexplained = 0, [], []
    for path in paths:
        z, meta, value, log, why = _replay_record_on_gpu(path, prove=True)
        name = os.path.basename(path)
        if why is None:
            assert len(log) == meta["n_verify"] and value == meta["value"], name
            identical += 1
            continue
        call, kind, x, limit = why
        ok = kind != "unexplained" and x <= limit
        (explained if ok else unexplained).append((name, call, kind, x if isinstance(x, str) else float(x), limit))
        if ok:
            helpers.note_escape(f"{name} verify call {call}: {kind} (limit {limit:.3g})", x)
    helpers.HARNESS_RATE[tree] = (identical, len(paths), [(k, round(v, 4), round(lim, 3)) for _, _, k, v, lim in explained],
                                  [(nm, c, k, v if isinstance(v, str) else round(v, 3), round(lim, 3)) for nm, c, k, v, lim in unexplained])
    # the honest statement is the printed rate; the gate: most unscreened runs replay token for token, and runs that part
    # from the reference's at a decision two ulps per logit do NOT explain stay rare (they are listed, not hidden: exact ties
    # inside the nucleus cut and other orders the reference leaves to torch's unstable CPU sort end up here)
    assert identical >= len(paths) // 2, (identical, explained, unexplained)
    assert len(unexplained) <= 2, unexplained
