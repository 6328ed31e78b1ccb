# GPU-box stages of round 6 (run through gpurun; everything lands under gpurun_out/r06/):
#   bash tools/gpu_r06.sh <stage> [<stage> ...]
#   block     the fused draft kernels (csrc/draft_block.hip) against the launches they replace + the oracle
#   traces    every end-to-end replay of the reference's traces
#   components  the 128-row projections on the TS_DBG experiment builds (tools/ts_dbg_build.sh first); tools/ts_components_loads_only.sh,
#             tools/ts_occ2.sh: the loads-only builds and the two-workgroups-per-CU build
#   tp8       configuration E at TP = 8, eight ranks on the one GPU (functional)
#   large     the round's new parity surface: the reference's 193- / 256- / 512-node growmaps (kernel level, host-driven loop, whole-step graphs)
#   kernels   tests/test_hip_kernels.py (every C-ABI kernel against the oracle)
#   tests     the whole GPU suite + smoke()
#   pmc_ns    PMC passes of the north-star kernels (samplers, verifier): FETCH_SIZE | WRITE_SIZE | SQ group, separate rocprofv3 runs
#   pmc_l2    L2 / TA / SQ counters of the 7B projections at the shipped 128-row plans (tools/ts_bench)
#   bench     the driver's command line
#   benchfull the default line (200 steps, other_configs, cpu_baseline)
#   loop      rocprofv3 --kernel-trace --stats of the loop alone
#   exp:<name>=<env assignments>   bench.py --steps 60 under the given environment (comma-separated VAR=VALUE), e.g. exp:ov0=SEQUOIA_OVERLAP_LAST_LEVEL=0
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], round(d["value"], 1), d["unit"], round(d["ms_per_step"], 3), "ms/step", d.get("mean_accepted_len"),
          "steady", d.get("value_steady"), "roof", r.get("kernel"), r.get("frac") and round(r["frac"], 3), "step_frac", (d.get("step_roofline") or {}).get("frac"))
    for c, o in (d.get("other_configs") or {}).items():
        print("   config", c, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in o.items() if k in ("value", "ms_per_step", "mean_accepted_len", "error", "weight_build_s")},
              "roof", (o.get("roofline") or {}).get("kernel"), (o.get("roofline") or {}).get("frac"), "step_frac", (o.get("step_roofline") or {}).get("frac"))
    k = d.get("kernels") or {}
    print("   kernels us/step:", {n: round(v["per_step_us"], 1) for n, v in k.items()})
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
}
pmc() {   # pmc <tag> <counters...> -- <command...>
  local ptag=$1; shift; local ctr=""; while [ "$1" != "--" ]; do ctr="$ctr $1"; shift; done; shift
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc/$ptag -o r -- "$@" > $O/pmc/$ptag.log 2>&1)
}
for stage in "$@"; do
  echo "=== $stage"
  case $stage in
  large)
    timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_e2e_gpu.py tests/test_step_pipeline_gpu.py -m gpu -q -k "L_ or 512 or 116 or 93 or 300 or 193 or 256" > $O/tests_large.log 2>&1
    grep -n "passed\|failed\|rror" $O/tests_large.log | tail -12 ;;
  overlap)     # the forked last draft level: whole-step graphs of the traces whose draft runs the tall-skinny path
    timeout 1200 python -m pytest tests/test_step_pipeline_gpu.py -m gpu -q -k "V32k_seq128 or B_topp09 or D_13b_w4 or E_70b_w2 or L_S256_v32k or config_b or eos" > $O/tests_overlap.log 2>&1
    grep -n "passed\|failed\|rror" $O/tests_overlap.log | tail -8 ;;
  pmc_ts)      # FETCH / WRITE / MFMA passes of the projection kernel at the shipped plans + tree attention -> profiles/r06_pmc.json
    bash tools/pmc_r06.sh > $O/pmc_ts_run.log 2>&1; tail -14 $O/pmc_ts_run.log
    [ -f gpurun_out/r06/pmc_ts/r06_pmc.json ] && cp gpurun_out/r06/pmc_ts/r06_pmc.json $O/pmc.json && cp $O/pmc.json profiles/r06_pmc.json ;;
  tsplans)     # the shipped 128-row 7B plans and the 129-row 70B plans, standalone (tools/ts_bench)
    for spec in "7b:128:qkv:128:2" "7b:128:o+res:64:4" "7b:128:gate_up+silu:230:1" "7b:128:down+res:64:4" "7b:128:o+res:128:2" "7b:128:down+res:128:2" "7b:128:down+res:32:8" "70b:129:qkv:128:2" "70b:129:o+res:128:2" "70b:129:down+res:128:4" "70b:129:down+res:64:8" "13b:64:qkv:120:2"; do
      IFS=: read arch rows shape tiles splits <<< "$spec"
      TS_ARCH=$arch TS_ONLY="$shape" TS_TILES=$tiles TS_SPLITS=$splits timeout 200 $GRAFT_REPO_ROOT/tools/ts_bench $rows 2>&1 | grep "us "
    done ;;
  loopE)       # rocprofv3 --kernel-trace --stats of configuration E's loop (TP = 1)
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_loopE -o b -- python $GRAFT_REPO_ROOT/bench.py --config E --steps 24 --warmup 4 --no-kernel-rooflines --no-cpu-baseline --no-tuned-growmap --no-autoregressive --no-other-configs --no-reference-metric > $O/prof_loopE.log 2>&1)
    python tools/rocprof_summary.py $(find $O/prof_loopE -name "*results.db" | head -1) 30 > $O/kernel_stats_loop_configE.md; find $O/prof_loopE -name "*.db" -delete
    head -22 $O/kernel_stats_loop_configE.md | cut -c1-150 ;;
  tslinear)
    timeout 900 python -m pytest tests/test_ts_linear_gpu.py -m gpu -q > $O/tests_ts_linear.log 2>&1; tail -3 $O/tests_ts_linear.log | cut -c1-300 ;;
  tunetail)    # launch plans for the 16 MT + 1 row builds: 7B at 65 rows (config C), full-width 70B at 129 rows (config E)
    rm -f $O/ts_tune_tail.log
    for shape in qkv "o+res" "gate_up+silu" "down+res"; do
      TS_ARCH=7b TS_ONLY="$shape" timeout 300 $GRAFT_REPO_ROOT/tools/ts_bench 65 >> $O/ts_tune_tail.log 2>&1
      TS_ARCH=70b TS_ONLY="$shape" timeout 300 $GRAFT_REPO_ROOT/tools/ts_bench 129 >> $O/ts_tune_tail.log 2>&1
    done
    python tools/ts_tune_pick.py $O/ts_tune_tail.log 4
    SEQUOIA_TS_TAIL=0 TS_ARCH=7b timeout 300 $GRAFT_REPO_ROOT/tools/ts_bench 65 > $O/ts_tune_notail_7b_65.log 2>&1
    python tools/ts_tune_pick.py $O/ts_tune_notail_7b_65.log 2 ;;
  tailtraces)
    timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_step_pipeline_gpu.py tests/test_tp_native_gpu.py -m gpu -q -k "C_7b or E_70b_w2 or native" > $O/tests_tail_traces.log 2>&1
    grep -n "passed\|failed\|rror" $O/tests_tail_traces.log | tail -6 | cut -c1-300 ;;
  tune70b)     # launch plans of the full-width 70B projections at 129 rows (configuration E at TP = 1): every (tiles, splits) candidate
    for shape in qkv "o+res" "gate_up+silu" "down+res"; do
      TS_ARCH=70b TS_ONLY="$shape" timeout 300 $GRAFT_REPO_ROOT/tools/ts_bench 129 >> $O/ts_tune_70b_129rows.log 2>&1
    done
    python tools/ts_tune_pick.py $O/ts_tune_70b_129rows.log ;;
  xgmi)
    timeout 1500 python -m pytest tests/test_xgmi_allreduce_gpu.py -m gpu -q > $O/tests_xgmi.log 2>&1; tail -3 $O/tests_xgmi.log | cut -c1-300 ;;
  tp2)         # configuration E tensor-parallel with two ranks on the ONE GPU (functional: xGMI kernels over hipIpc-mapped buffers)
    SEQUOIA_TS_EXCLUSIVE=1 SEQUOIA_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --config E --backend gloo --steps 8 --warmup 2 --no-cpu-baseline --no-autoregressive > $O/benchE_tp2.json 2> $O/benchE_tp2.err; line $O/benchE_tp2.json; tail -2 $O/benchE_tp2.err | cut -c1-300 ;;
  tp8)         # configuration E at TP = 8 with all eight ranks on the ONE GPU (functional: the real 70B shard shapes, xGMI kernels over
    #              hipIpc-mapped buffers, whole-step graphs; the times are time-slicing, not link measurements)
    SEQUOIA_TS_EXCLUSIVE=1 SEQUOIA_BENCH_ONE_DEVICE=1 timeout 1500 python bench.py --gpus 8 --config E --backend gloo --steps 6 --warmup 2 --no-cpu-baseline --no-autoregressive > $O/benchE_tp8.json 2> $O/benchE_tp8.err; line $O/benchE_tp8.json; tail -2 $O/benchE_tp8.err | cut -c1-300
    python -c "
import json;d=json.loads(open('$O/benchE_tp8.json').read().strip().splitlines()[-1]);a=d.get('allreduce') or {};print('n_gpus',d.get('n_gpus'),'rccl_ranks',d.get('rccl_ranks'),'allreduce',{k:a.get(k) for k in ('kind','xgmi_status','xgmi_self_check','xgmi_us','workspace')},'step_loop',(d.get('config') or {}).get('step_loop'))" ;;
  replicas2)   # the driver's N > 1 command with two replica ranks on the one GPU (gloo), incl. the tensor-parallel child job
    SEQUOIA_BENCH_ONE_DEVICE=1 timeout 1200 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_replicas2.json 2> $O/bench_replicas2.err; line $O/bench_replicas2.json; python -c "
import json;d=json.loads(open('$O/bench_replicas2.json').read().strip().splitlines()[-1]);t=d.get('tp_70b') or {};print('n_gpus',d.get('n_gpus'),'rccl_ranks',d.get('rccl_ranks'),'tp_70b',{k:t.get(k) for k in ('value','ms_per_step','allreduce_kind','xgmi_status','error','steady_ms_per_step')})" ;;
  baselines)
    timeout 900 python -m pytest tests/test_baselines_gpu.py tests/test_properties_gpu.py -m gpu -q > $O/tests_baselines.log 2>&1; tail -6 $O/tests_baselines.log | cut -c1-300 ;;
  kvonly)
    timeout 600 python -m pytest tests/test_e2e_gpu.py -m gpu -q -k "kv_only" > $O/tests_kvonly.log 2>&1; tail -3 $O/tests_kvonly.log | cut -c1-300 ;;
  lossless)
    timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -q -k "lossless" > $O/tests_lossless.log 2>&1; tail -3 $O/tests_lossless.log | cut -c1-300 ;;
  block)       # the fused draft attention block against the launch sequence it replaces + the oracle
    timeout 900 python -m pytest tests/test_draft_block_gpu.py -m gpu -q > $O/tests_block.log 2>&1; tail -15 $O/tests_block.log | cut -c1-400 ;;
  traces)      # every end-to-end replay of the reference's traces (host-driven and whole-step graphs)
    timeout 2400 python -m pytest tests/test_e2e_gpu.py tests/test_step_pipeline_gpu.py tests/test_baselines_gpu.py tests/test_probe_gpu.py -m gpu -q > $O/tests_traces.log 2>&1
    grep -n "passed\|failed\|rror\|margin" $O/tests_traces.log | tail -12 | cut -c1-300 ;;
  components)  # the 128-row 7B projections at the shipped plans, production build and TS_DBG experiment builds (tools/ts_dbg_build.sh):
    #              which part of a launch is the weight stream, the activation ingest, the MFMAs, the merge + stores
    for bits in 0 1 2 3 4 32 24 36; do
      for spec in "qkv:128:2" "o+res:64:4" "gate_up+silu:230:1" "down+res:64:4"; do
        IFS=: read shape tiles splits <<< "$spec"
        if [ $bits = 0 ]; then lib=""; else lib="LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/_dbg/$bits"; fi
        echo -n "TS_DBG=$bits $shape ${tiles}x$splits: "
        env $lib TS_ARCH=7b TS_ONLY="$shape" TS_TILES=$tiles TS_SPLITS=$splits timeout 120 $GRAFT_REPO_ROOT/tools/ts_bench 128 2>&1 | grep "us " | head -1
      done
    done > $O/ts_components.log 2>&1
    cat $O/ts_components.log | cut -c1-200 ;;
  kernels)
    timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q > $O/tests_kernels.log 2>&1; tail -3 $O/tests_kernels.log ;;
  tests)
    timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; grep -n "passed\|failed\|error" $O/tests_gpu.log | tail -3
    python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 ;;
  pmc_ns)
    mkdir -p $O/pmc
    for what in samp verify; do
      pmc ns_${what}_FETCH FETCH_SIZE -- python $GRAFT_REPO_ROOT/tools/kbench.py $what
      pmc ns_${what}_WRITE WRITE_SIZE -- python $GRAFT_REPO_ROOT/tools/kbench.py $what
      pmc ns_${what}_SQ SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python $GRAFT_REPO_ROOT/tools/kbench.py $what
    done
    args=""; for d in $O/pmc/ns_*/; do db=$(find $d -name "*results.db" | head -1); [ -n "$db" ] && args="$args $(basename $d)=$db"; done
    python tools/pmc_summary.py $O/pmc_northstar_raw.json $args > /dev/null; find $O/pmc -name "*.db" -delete
    python tools/pmc_r05_summary.py northstar $O/pmc_northstar_raw.json $O/pmc_northstar.json ;;
  pmc_l2)
    mkdir -p $O/pmc
    for spec in "qkv:qkv:128:2" "o:o+res:64:4" "gate_up:gate_up+silu:230:1" "down:down+res:64:4"; do
      IFS=: read tag shape tiles splits <<< "$spec"
      export TS_ARCH=7b TS_ONLY="$shape" TS_TILES=$tiles TS_SPLITS=$splits
      pmc l2_${tag}_TCC TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr GRBM_GUI_ACTIVE -- $GRAFT_REPO_ROOT/tools/ts_bench 128
      pmc l2_${tag}_TCP TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum -- $GRAFT_REPO_ROOT/tools/ts_bench 128
      pmc l2_${tag}_SQ SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -- $GRAFT_REPO_ROOT/tools/ts_bench 128
      unset TS_ARCH TS_ONLY TS_TILES TS_SPLITS
    done
    args=""; for d in $O/pmc/l2_*/; do db=$(find $d -name "*results.db" | head -1); [ -n "$db" ] && args="$args $(basename $d)=$db"; done
    python tools/pmc_summary.py $O/pmc_l2_raw.json $args > /dev/null; find $O/pmc -name "*.db" -delete
    python tools/pmc_r05_summary.py l2 $O/pmc_l2_raw.json $O/pmc_l2.json ;;
  bench)
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; line $O/bench_driver_cmd.json ;;
  benchfull)
    timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; line $O/bench_default.json ;;
  loop)
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_loop -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 120 --warmup 8 --no-kernel-rooflines --no-cpu-baseline --no-tuned-growmap --no-autoregressive --no-other-configs --no-reference-metric > $O/prof_loop.log 2>&1)
    python tools/rocprof_summary.py $(find $O/prof_loop -name "*results.db" | head -1) 40 > $O/kernel_stats_loop_only.md; find $O/prof_loop -name "*.db" -delete
    head -24 $O/kernel_stats_loop_only.md ;;
  exp:*)
    spec=${stage#exp:}; name=${spec%%=*}; envs=${spec#*=}
    ( IFS=,; for kv in $envs; do export "$kv"; done; unset IFS
      timeout 600 python bench.py --config ${BENCH_CONFIG:-B} --steps ${BENCH_STEPS:-200} --warmup 8 ${BENCH_ARGS:---steady-window --no-kernel-rooflines --no-cpu-baseline --no-tuned-growmap --no-autoregressive --no-other-configs --no-reference-metric} > $O/exp_$name.json 2> $O/exp_$name.err )
    line $O/exp_$name.json ;;
  *) echo "unknown stage $stage" ;;
  esac
done
