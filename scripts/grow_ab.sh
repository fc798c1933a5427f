# growth per launch of the int8 candidate scan (HX_DEBUG_GROW_MAX8): fewer, larger launches against more appended rows per launch.
# usage (GPU box): bash scripts/grow_ab.sh
R=$GRAFT_REPO_ROOT
export AB_L=100
for rep in 1 2; do
for G in 16 24 50 12; do
  HX_DEBUG_GROW_MAX8=$G timeout -k 10 120 python $R/scripts/cand8_hits.py 2>&1 | grep -E "^\{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('grow_max $G', 'ms_per_call %.3f  scan %.3f  launches %.0f  rest %.3f  retries %d uncert %d' % (d['ms_per_call'], d['scan_ms'], d['launches'], d['ms_per_call']-d['scan_ms'], d['retries'], d['uncertified']))"
done
done
