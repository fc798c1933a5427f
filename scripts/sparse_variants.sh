# A/B of compile-time variants of the sparse select pass on ONE box (libraries built beforehand as
# scripts/ubench/build/libhx_<name>.so); usage (GPU box): VARIANTS="acc0 acc1" bash scripts/sparse_variants.sh
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for V in ${VARIANTS}; do
  echo "== $V"
  HX_LIB_PATH=$R/scripts/ubench/build/libhx_$V.so timeout -k 10 200 python $R/scripts/sp_only.py 2>/dev/null | tail -1 | cut -c1-120
done
done
