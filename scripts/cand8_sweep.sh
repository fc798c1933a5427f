# geometry sweep of the int8 candidate pass on the bench workload (one process per setting)
# usage (GPU box): bash scripts/cand8_sweep.sh > gpurun_out/cand8_sweep.log
R=$GRAFT_REPO_ROOT
export AB_ONLY=i8 HX_DEBUG_GEOMETRY=1
for S in "64 4 288 4096" "24 4 288 4096" "16 4 288 4096" "12 4 288 4096" "8 4 288 4096" "16 3 220 4096" "12 3 220 4096" "16 4 288 8192" "64 4 288 8192" "16 5 400 4096"; do
  set -- $S
  echo "=== grow_max=$1 mul=$2 add=$3 C=$4"
  HX_DEBUG_GROW_MAX8=$1 HX_DEBUG_CAND8_MUL=$2 HX_DEBUG_CAND8_ADD=$3 HX_DEBUG_CAND8_C=$4 timeout -k 10 120 python $R/scripts/cand8_ab.py 10000000 1024 10 2>&1 | grep -E "geometry L=100 approx=1 safe=0 cand8=1|^i8 " | cut -c1-420
done
