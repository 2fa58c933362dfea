#!/bin/bash
# Where does the SpMV spend its time on an irregular mesh?  Cache-hierarchy counters of k_spmv on the 120^3 box
# whole / with 40 % of the elements knocked out (one counter group per pass; program directly behind `--`).
# usage (GPU box, repo root): bash tools/gather_pmc.sh <outdir>
OUT=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $R/$OUT/list_avail.txt 2>&1
i=0
for G in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
         "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum" "GRBM_GUI_ACTIVE TCC_BUSY_avr TCC_TAG_STALL_sum"; do
  i=$((i+1))
  for K in 0 0.4; do
    timeout 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/$OUT/g${i}_k$K -o pmc -- python3 $R/bench.py --size 120 --knockout $K --steps 1 --warmup 0 --no-cpu --placement-tries 1 > $R/$OUT/bench_g${i}_k$K.json 2> $R/$OUT/bench_g${i}_k$K.err
    echo "group $i [$G] knockout $K rc=$?"
  done
done
cd $R
python3 tools/gather_pmc_summary.py $OUT | tee $OUT/summary.txt
