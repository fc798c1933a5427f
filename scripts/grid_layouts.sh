# per-rank step of the layouts an N-GPU job could use for the 10M x 768, B = 1024 workload: R row shards x Q query
# groups (R * Q = N), each rank scanning 10M / R rows for 1024 / Q queries; the exchange is a local stand-in.
# usage (GPU box): bash scripts/grid_layouts.sh
R=$GRAFT_REPO_ROOT
for cfg in "1250000 8 1024" "2500000 4 512" "5000000 2 256" "10000000 1 128" "10000000 1 256" "5000000 2 512" "10000000 1 512" "5000000 2 1024"; do
  echo "== rows/world/batch: $cfg"
  timeout -k 10 200 python $R/scripts/shard_overhead.py $cfg 2>/dev/null | grep -E "one ABI|pipelined|local dense|local sparse"
done
