# Everything behind profiles/r02_*: the bench line, its rocprofv3 kernel stats, three PMC passes, the small-batch
# kernel stats.  usage (on the GPU box): bash scripts/final_prof.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${1:-r04}
timeout -k 10 500 python $R/bench.py > $R/gpurun_out/bench_$V.json 2> $R/gpurun_out/bench_$V.err
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$V -o $V -- python $R/bench.py --steps 5 --no-cpu-baseline --no-secondary --no-alone > $R/gpurun_out/bench_${V}p.json 2> $R/gpurun_out/bench_${V}p.err
echo "kernel stats done"
for G in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  N=$(echo $G | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$V/$N -o x -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-alone > $R/gpurun_out/pmc_${V}_$N.log 2>&1
  echo "pmc $N done"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${V}_b32 -o b32 -- python $R/scripts/hbm_roofline.py 10000000 > $R/gpurun_out/hbm_$V.json 2> $R/gpurun_out/hbm_$V.err
echo "small batch done"
cut -c1-300 $R/gpurun_out/bench_$V.json
