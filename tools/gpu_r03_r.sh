cd $GRAFT_REPO_ROOT
for flags in "--no-graphs" "--sync-loop" "--pair random" "--growmap 8x8-tree --config C"; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-autoregressive --no-tuned-growmap $flags 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$flags', round(d['ms_per_step'],3), round(d['value'],1), d['config']['step_loop'][:13])"
done
SEQUOIA_COMMIT_ORDER=lossless timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-autoregressive --no-tuned-growmap 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lossless', round(d['ms_per_step'],3), d['config']['commit_order'])"
python -m sequoia_amd.testbed --help 2>&1 | head -5
