# A/B: the resident query tile of the small-batch scan against the streamed one (HX_DEBUG_NO_QRES).  usage (GPU box): bash scripts/qres_ab.sh
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for V in on off; do
  echo "== resident query tile: $V"
  if [ $V = off ]; then export HX_DEBUG_NO_QRES=1; else unset HX_DEBUG_NO_QRES; fi
  timeout -k 10 200 python $R/scripts/hbm_roofline.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print(' '.join('%d:%s:%.3f' % (x['batch'], x['stage'][:4], x['frac_of_hbm_peak']) for x in d['runs']))"
done
done
