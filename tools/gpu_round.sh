# One measurement pass of a round on the GPU box (run through gpurun; about 7 GPU-minutes):
#   bash tools/gpu_round.sh r04 [tests] [pmc] [bench] [loop] [tp2]      (no stage names = all of them, in this order)
#   tests  python -m pytest tests -m gpu -x -q + __graft_entry__.smoke()
#   pmc    tools/pmc_r04.sh: FETCH_SIZE / WRITE_SIZE / MFMA passes of the 128-row projection plans and tree attention
#          (separate rocprofv3 --pmc runs), summarised into profiles/<tag>_pmc.json stamped with the kernel-source sha --
#          bench.py refuses a record measured on other sources, so this comes BEFORE the bench stage
#   bench  the default line (with `other_configs` C / D), the driver's command, rocprofv3 --kernel-trace --stats of the default command
#   loop   rocprofv3 --kernel-trace --stats of the loop alone (--no-kernel-rooflines)
#   soak   3000 steps of the headline loop
#   tp2    configuration E tensor-parallel with two ranks on the ONE GPU (xGMI kernels over hipIpc-mapped buffers)
#   tp8    (not in the default set) the same with EIGHT ranks time-sliced on the one GPU: functional check of the TP = 8 plans / collectives
# Everything is written under gpurun_out/<tag>/ (created first: a redirect into a missing directory silently skips a stage);
# gpurun merges that directory back.  Afterwards, LOCALLY:   bash tools/gpu_round.sh r04 collect
# copies the records the judge reads into profiles/<tag>_* (profiles/ written on the GPU box does not come back).
TAG=${1:-r04}; shift
STAGES="$*"; [ -z "$STAGES" ] && STAGES="tests pmc bench loop soak tp2"
if [ "$STAGES" = "collect" ]; then          # local: gpurun_out/<tag>/ -> profiles/<tag>_*
  cd "$(dirname "$0")/.." && O=gpurun_out/$TAG
  for pair in pmc.json:pmc.json bench_default.json:bench_default.json bench_driver_cmd.json:bench_driver_cmd.json \
              kernel_stats.md:bench_kernel_stats.md \
              kernel_stats_loop_only.md:bench_kernel_stats_loop_only.md benchE_tp2.json:bench_configE_tp2_one_gpu.json benchE_tp8.json:bench_configE_tp8_functional_one_gpu.json bench_soak.json:bench_soak_3000steps.json; do
    src=$O/${pair%%:*}; [ -s $src ] && cp $src profiles/${TAG}_${pair##*:} && echo "profiles/${TAG}_${pair##*:}"
  done
  exit 0
fi
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O profiles
has() { case " $STAGES " in *" $1 "*) return 0;; *) return 1;; esac; }
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], round(d["value"], 1), d["unit"], round(d["ms_per_step"], 3), "ms/step", d.get("mean_accepted_len"),
          "roof", r.get("kernel"), r.get("frac") and round(r["frac"], 3), "traffic", r.get("traffic"),
          "step_frac", (d.get("step_roofline") or {}).get("frac"), "ref_metric", d.get("value_reference_metric"),
          "prefill_ms", d.get("prefill_step_ms"))
    for c, o in (d.get("other_configs") or {}).items():
        print("   config", c, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in o.items() if k in ("value", "ms_per_step", "mean_accepted_len", "error")},
              "roof", (o.get("roofline") or {}).get("kernel"), (o.get("roofline") or {}).get("frac"), "step_frac", (o.get("step_roofline") or {}).get("frac"))
    k = d.get("kernels") or {}
    print("   kernels us/step:", {n: round(v["per_step_us"], 1) for n, v in k.items()})
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
}
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; grep -n "passed\|failed\|error" $O/tests_gpu.log | tail -2
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
fi
if has pmc; then
  bash tools/pmc_r04.sh > $O/pmc_run.log 2>&1; tail -14 $O/pmc_run.log
  [ -f gpurun_out/r04/pmc/r04_pmc.json ] && cp gpurun_out/r04/pmc/r04_pmc.json $O/pmc.json && cp $O/pmc.json profiles/${TAG}_pmc.json   # (on the box: for the bench stage)
fi
if has bench; then
  timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; line $O/bench_default.json
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; line $O/bench_driver_cmd.json
  # (configs C and D ride in the default line since round 4: `other_configs`)
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-tuned-growmap --no-autoregressive --no-other-configs --no-reference-metric > $O/prof.log 2>&1)
  python tools/rocprof_summary.py $(find $O/prof -name "*results.db" | head -1) 45 > $O/kernel_stats.md; find $O/prof -name "*.db" -delete
fi
if has loop; then
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_loop -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 120 --warmup 8 --no-kernel-rooflines --no-cpu-baseline --no-tuned-growmap --no-autoregressive --no-other-configs --no-reference-metric > $O/prof_loop.log 2>&1)
  python tools/rocprof_summary.py $(find $O/prof_loop -name "*results.db" | head -1) 40 > $O/kernel_stats_loop_only.md; find $O/prof_loop -name "*.db" -delete
  head -16 $O/kernel_stats_loop_only.md
fi
if has soak; then      # a long run of the headline loop: long-run tokens / step of the synthetic pair, no drift, no hang
  timeout 600 python bench.py --steps 3000 --warmup 8 --no-cpu-baseline --no-autoregressive --no-tuned-growmap --no-other-configs --no-reference-metric > $O/bench_soak.json 2> $O/bench_soak.err; line $O/bench_soak.json
fi
if has tp8; then       # functional only: EIGHT tensor-parallel ranks time-sliced on the ONE GPU (times mean nothing)
  SEQUOIA_TS_EXCLUSIVE=1 SEQUOIA_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 8 --config E --backend gloo --steps 4 --warmup 2 --no-cpu-baseline --no-autoregressive > $O/benchE_tp8.json 2> $O/benchE_tp8.err; line $O/benchE_tp8.json; tail -3 $O/benchE_tp8.err
fi
if has tp2; then
  SEQUOIA_TS_EXCLUSIVE=1 SEQUOIA_BENCH_ONE_DEVICE=1 timeout 700 python bench.py --gpus 2 --config E --backend gloo --steps 8 --warmup 2 --no-cpu-baseline --no-autoregressive > $O/benchE_tp2.json 2> $O/benchE_tp2.err; line $O/benchE_tp2.json
fi
