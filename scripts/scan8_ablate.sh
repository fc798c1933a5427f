# ablations of the int8 candidate scan (timing builds, wrong results): HX_SCAN_DBG = 1 one corpus tile over and over (no HBM
# traffic), 2 no loads in the loop, 3 no MFMAs, 4 no threshold filter,
# 5 no fragment reads in the loop, 6 MFMAs + barriers only, 7 MFMAs only, 8 no loads and no filter.  usage (GPU box): bash scripts/scan8_ablate.sh
R=$GRAFT_REPO_ROOT
export HX_LIB_PATH=$R/scripts/ubench/build/libhx_dbg.so AB_L=100
for D in ${ABLATE:-0 4 1 2 8 5 6 7 3}; do
  echo "== HX_SCAN_DBG=$D"
  HX_SCAN_DBG=$D timeout -k 10 120 python $R/scripts/cand8_hits.py 2>&1 | grep -E "^\{" | cut -c1-300
done
