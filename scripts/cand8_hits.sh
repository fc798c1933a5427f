R=$GRAFT_REPO_ROOT
export HX_DEBUG_GEOMETRY=1
for S in "1 0 10" "1 40 10" "4 288 10" "4 288 100" "1 0 100"; do
  set -- $S
  HX_DEBUG_CAND8_MUL=$1 HX_DEBUG_CAND8_ADD=$2 AB_L=$3 timeout -k 10 100 python $R/scripts/cand8_hits.py 2>&1 | grep -E "cand8=1|^\{" | cut -c1-300
done
