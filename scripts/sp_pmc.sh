# PMC passes of the sparse stage alone (scripts/sp_only.py).  usage: bash scripts/sp_pmc.sh <tag> [rows]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${1:-sp}
N=${2:-10000000}
for G in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  T=$(echo $G | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sp_$V/$T -o x -- python $R/scripts/sp_only.py $N > $R/gpurun_out/pmc_sp_${V}_$T.log 2>&1
  echo "pmc $T done"
done
python $R/scripts/pmc_sum.py $R/gpurun_out/pmc_sp_$V k_sparse_select
