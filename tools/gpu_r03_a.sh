# Round-3 GPU pass A: full GPU suite (new: headline-dims traces, EOS inside the pipeline, xGMI all-reduce on the 2-rank rig,
# fused small-draft forward), then short bench lines with / without the draft fusion.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" $O/tests.log | tail -15
tail -40 $O/tests.log | head -60
timeout 600 python bench.py --steps 80 --warmup 8 --no-cpu-baseline --no-autoregressive --no-tuned-growmap > $O/bench_fused.json 2> $O/bench_fused.err; tail -c 1500 $O/bench_fused.json | head -c 1500; echo
SEQUOIA_DRAFT_FUSED=0 timeout 600 python bench.py --steps 80 --warmup 8 --no-cpu-baseline --no-autoregressive --no-tuned-growmap > $O/bench_unfused.json 2> $O/bench_unfused.err
python - <<'PY'
import json
for n in ("fused","unfused"):
    try:
        d=json.loads(open(f"gpurun_out/r3/a/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, "ms/step", round(d["ms_per_step"],3), "tok/s", round(d["value"],1), "acc", round(d["mean_accepted_len"],2), "host", d.get("host_driven_loop"))
    except Exception as e:
        print(n, "failed", e)
PY
tail -5 $O/bench_fused.err
