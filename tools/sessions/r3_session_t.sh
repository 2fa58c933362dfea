#!/bin/bash
# round 3, session t: what do the ragged streams fetch?  (FETCH_SIZE / L2 requests, padded against ragged, NT against plain)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_t
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "ragged" > $OUT/pytest_ragged.txt 2>&1
echo "ragged tests rc=$?"; tail -8 $OUT/pytest_ragged.txt | cut -c1-400
cd /tmp && export TMPDIR=/tmp
for CFG in "0 -1" "1 -1" "1 0"; do
  set -- $CFG
  for C in FETCH_SIZE "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    tag=ragged$1_variant$2_$(echo $C | cut -c1-9)
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$tag -o pmc -- python3 $R/bench.py --size 120 --knockout 0.4 --ragged $1 --spmv-variant $2 --steps 1 --warmup 0 --no-cpu --placement-tries 1 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
    python3 $R/tools/pmc_dir_summary.py $OUT/$tag
    find $OUT/$tag -name "*.csv" -size +2M -delete
  done
done
cd $R
for CFG in "0 -1" "1 -1" "1 0" "0 0"; do
  set -- $CFG
  timeout 600 python3 bench.py --size 120 --knockout 0.4 --ragged $1 --spmv-variant $2 --steps 2 --warmup 1 --no-cpu > $OUT/b.json 2>> $OUT/err.txt
  python3 - $OUT/b.json "ragged $1 variant $2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "DOF/s %.3e" % d["value"], "spmv ms %.4f" % d["roofline"]["avg_launch_ms"], "frac %.3f" % d["roofline"]["frac"], "its", d["config"]["cg_iterations"], "ragged", d["config"]["ragged_stream"])
PY
done
