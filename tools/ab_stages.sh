cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in "" "$GRAFT_REPO_ROOT/bath_amd/libbathhip_prev.so"; do
    BATH_HIP_LANES=1 BATH_HIP_LIBRARY=$lib python3 bench.py --steps 10 --warmup 2 --no-fs --no-streamed --no-cpu-baseline --no-one-part 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-current}'.split('/')[-1], 'one lane: %.3f ms/step' % d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms'].items()})"
  done
done
