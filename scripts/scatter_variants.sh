# A/B of the log scatter's waves per log on ONE box (libraries built beforehand as scripts/ubench/build/libhx_<name>.so): dense-only
# search at B = 1024, L = 100 -- ms per call, the scan kernels excluded.  usage (GPU box): VARIANTS="wpl1 wpl2 wpl4" bash scripts/scatter_variants.sh
R=$GRAFT_REPO_ROOT
export AB_L=100
for rep in 1 2; do
for V in ${VARIANTS}; do
  HX_LIB_PATH=$R/scripts/ubench/build/libhx_$V.so timeout -k 10 120 python $R/scripts/cand8_hits.py 2>&1 | grep -E "^\{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$V', 'ms_per_call %.3f  scan %.3f  rest %.3f  retries %d' % (d['ms_per_call'], d['scan_ms'], d['ms_per_call']-d['scan_ms'], d['retries']))"
done
done
