"""The reference's evaluation entry point (tests/testbed.py, tests/testbed_greedy.py) with its flags, runnable
without the reference checkout, tokenizers or dataset downloads:

    python -m sequoia_amd.testbed --model random:JackFram/llama-68m --target random:meta-llama/Llama-2-7b-hf \\
        --growmap A100-CNN-68m-7b-stochastic --T 0.6 --P 1.0 --M 384 --start 0 --end 8 --Mode fast

--Mode fast | greedy  : `simulation_fast` (tests/testbed.py:45-95) on SpecTree (testbed.py) / GreedyTree (testbed_greedy.py,
                        selected with --tree greedy)
--Mode baseline       : `simulation_baseline` (:99-143), the target alone
--Mode benchmark      : `simulation_benchmark` (:144-219), per-phase timers
--tree specinfer|greedys runs the paper's comparison baselines (tests/test_specinfer.py, tests/test_greedyS.py).
Models: a local HF directory or `random:<arch>[:seed=N]`; `--pair calibrated` builds the synthetic draft/target pair of
bench.py.  Prompts: rows [start, end) of the bundled c4_small extract (pre-tokenised, first 128 tokens), the only
real-text prompts available offline.  Prints the reference's summary line.
"""
from __future__ import annotations

import argparse
import random
import time

import numpy as np
import torch

from .growmap import GrowMap
from .harness import AutoregressiveLoop, load_prompts


def setup_seed(seed: int):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def build_engines(args, device):
    from .Engine.Engine import GraphInferenceEngine, GraphInferenceEngineTG
    if args.pair == "calibrated":
        from .synthetic import calibrated_pair_specs
        dspec, tspec = calibrated_pair_specs(args.model.split(":")[1] if args.model.startswith("random:") else args.model,
                                             args.target.split(":")[1] if args.target.startswith("random:") else args.target,
                                             device)
    else:
        dspec, tspec = args.model, args.target
    if args.offloading:
        from .Engine.offload_engine import OffloadEngine
        target = OffloadEngine(max_length=args.M, model_name_or_path=tspec, dtype=torch.float16, device=device)
    else:
        target = GraphInferenceEngineTG(max_length=args.M, model_name_or_path=tspec, dtype=torch.float16, device=device)
    draft = None
    if args.Mode != "baseline":
        draft = GraphInferenceEngine(max_length=args.M, model_name_or_path=dspec, dtype=torch.float16, device=device)
    return draft, target


def tree_class(name: str):
    from .Tree.GreedySTree import GreedySTree
    from .Tree.GreedyTree import GreedyTree
    from .Tree.SpecInferTree import SpecInferTree
    from .Tree.SpecTree import SpecTree
    return {"sequoia": SpecTree, "greedy": GreedyTree, "specinfer": SpecInferTree, "greedys": GreedySTree}[name]


def simulation(args, draft, target, prompts, device, benchmark: bool):
    """simulation_fast / simulation_benchmark: the loop, counters and summary line of tests/testbed.py:45-95,144-219."""
    gm = GrowMap.load(args.growmap)
    grow_map = gm.to_reference_dict()
    cls = tree_class(args.tree)
    M = args.M
    attn_mask = torch.full((M, M), torch.finfo(torch.float16).min, dtype=torch.float16, device=device)
    position_ids = torch.zeros(M, dtype=torch.long, device=device)
    lens = sorted({lv.total for lv in gm.levels} | {1})
    draft.initialize_cuda_graph(lens)
    if hasattr(target, "initialize_cuda_graph"):
        target.initialize_cuda_graph([gm.size])
    decoding_steps = large_model_steps = 0
    total_time = 0.0
    phases = np.zeros(5)                  # large model run, accept loop, kv select, small model run, sample time
    for prompt in prompts:
        input_ids = torch.tensor(prompt[:128], dtype=torch.long)
        draft.clear_kv(); target.clear_kv()
        tree = cls(prefix=input_ids, device=device, temperature=args.T, top_p=args.P, draft_kv_len=0, target_kv_len=0,
                   draft_model_engine=draft, target_model_engine=target, max_length=M, max_target_seq=M,
                   grow_map=grow_map, attn_mask=attn_mask, sequence=None, new_tokens_buffer=None, parents_buffer=None,
                   position_ids=position_ids, residual_graph=None, sampling_callables=None, sample_gather_indices=None)
        cur, terminate = input_ids.shape[0], False
        torch.cuda.synchronize()
        t1 = time.time()
        while cur < 256 and not terminate:
            if benchmark:
                s_t, c_t = tree.construct_grow_map(benchmark=True)
                valid, _, _, a, b, c, terminate = tree.verify(benchmark=True)
                phases += [a, b, c, c_t, s_t]
            else:
                tree.construct_grow_map()
                valid, _, _, terminate = tree.verify()
            decoding_steps += valid.shape[0] - cur
            cur = valid.shape[0]
            large_model_steps += 1
            if int(valid[-1]) in (0, 2):
                terminate = True
        torch.cuda.synchronize()
        total_time += time.time() - t1
    n = max(large_model_steps, 1)
    if benchmark:
        print("large model run: {:.5f}, accept loop: {:.5f}, kv select: {:.5f}, small model run: {:.5f}, sample time: {:.5f}"
              .format(*(phases / n)))
    print("total time :{:.5f}s, latency :{:.5f}s, decoding step: {}, large model step: {}, {}".format(
        total_time, total_time / max(decoding_steps, 1), decoding_steps, large_model_steps, decoding_steps / n))
    return decoding_steps / n


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", type=str, default="random:JackFram/llama-68m", help="draft model")
    ap.add_argument("--target", type=str, default="random:meta-llama/Llama-2-7b-hf", help="target model")
    ap.add_argument("--dataset", type=str, default="c4_small", help="only the bundled c4_small extract is available offline")
    ap.add_argument("--growmap", type=str, default="A100-CNN-68m-7b-stochastic", help="bundled name, .json or reference .pt")
    ap.add_argument("--start", type=int, default=0)
    ap.add_argument("--end", type=int, default=8)
    ap.add_argument("--T", type=float, default=0.6)
    ap.add_argument("--P", type=float, default=1.0)
    ap.add_argument("--M", type=int, default=384)
    ap.add_argument("--seed", type=int, default=17)
    ap.add_argument("--Mode", type=str, default="fast", choices=["fast", "greedy", "baseline", "benchmark"])
    ap.add_argument("--tree", type=str, default="sequoia", choices=["sequoia", "greedy", "specinfer", "greedys"])
    ap.add_argument("--pair", type=str, default="as-given", choices=["as-given", "calibrated"])
    ap.add_argument("--offloading", action="store_true", help="tensor-parallel target in the OffloadEngine slot")
    args = ap.parse_args(argv)
    print(args)
    setup_seed(args.seed)
    device = "cuda:0"
    prompts = load_prompts()[args.start:args.end]
    draft, target = build_engines(args, device)
    if args.Mode == "baseline":
        res = AutoregressiveLoop(dict(M=args.M), target, device, prompts, T=args.T).run(len(prompts))
        print("total time :{:.5f}s, latency :{:.5f}s, decoding step: {}".format(
            res["tokens"] * res["ms_per_token"] * 1e-3, res["ms_per_token"] * 1e-3, res["tokens"]))
        return res
    return simulation(args, draft, target, prompts, device, benchmark=args.Mode == "benchmark")


if __name__ == "__main__":
    main()
