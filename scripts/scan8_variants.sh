# A/B of compile-time variants of the int8 candidate scan on ONE box: builds are made beforehand (CPU side) as
# scripts/ubench/build/libhx_<name>.so; usage (GPU box): VARIANTS="base prio0 prio2" bash scripts/scan8_variants.sh
R=$GRAFT_REPO_ROOT
export AB_L=100
for rep in 1 2; do
for V in ${VARIANTS}; do
  echo "== $V"
  HX_LIB_PATH=$R/scripts/ubench/build/libhx_$V.so timeout -k 10 120 python $R/scripts/cand8_hits.py 2>&1 | grep -E "^\{" | cut -c1-220
done
done
