# A/B of compile-time variants of the small-batch scan on ONE box (libraries built beforehand as
# scripts/ubench/build/libhx_<name>.so); usage (GPU box): VARIANTS="nt0 nt1" bash scripts/hbm_variants.sh
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for V in ${VARIANTS}; do
  echo "== $V"
  HX_LIB_PATH=$R/scripts/ubench/build/libhx_$V.so timeout -k 10 200 python $R/scripts/hbm_roofline.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print(' '.join('%d:%s:%.3f' % (x['batch'], x['stage'][:4], x['frac_of_hbm_peak']) for x in d['runs']))"
done
done
