#!/bin/bash
# Runtime knobs of the HIP runtime against the steady step of configuration B (graph replay): does any of them move
# the ~1.6 us a dependent graph node costs?  One lean bench run per setting, same box, the default first and last.
#   gpurun -- bash tools/env_knobs.sh
mkdir -p gpurun_out/r06
out=gpurun_out/r06/env_knobs.log
: > $out
run() {
  echo "== $*" | tee -a $out
  env "$@" python bench.py --steps 200 --warmup 8 --no-cpu-baseline --no-kernel-rooflines --no-other-configs --no-autoregressive \
      --no-reference-metric 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('   ms_per_step', j['ms_per_step'], 'steady', j.get('steady_ms_per_step', j.get('config',{}).get('steady_ms_per_step')), 'value', j['value'])
" | tee -a $out
}
run X=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run GPU_MAX_HW_QUEUES=1
run HSA_NO_SCRATCH_RECLAIM=1
run DEBUG_HIP_GRAPH_DOT_PRINT=0 AMD_DIRECT_DISPATCH=1
run HIP_GRAPH_KERNEL_ARG_OPT=1
run X=1
