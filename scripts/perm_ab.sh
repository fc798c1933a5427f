# does the strided tile order of the scan (kernels.hpp) cost time against the physical order?  usage (GPU box): bash scripts/perm_ab.sh
R=$GRAFT_REPO_ROOT
export AB_L=100
for rep in 1 2; do
  echo "== strided"; timeout -k 10 120 python $R/scripts/cand8_hits.py 2>&1 | grep -E "^\{" | cut -c1-200
  echo "== physical"; HX_DEBUG_NO_PERM=1 timeout -k 10 120 python $R/scripts/cand8_hits.py 2>&1 | grep -E "^\{" | cut -c1-200
done
