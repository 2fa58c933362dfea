#!/bin/bash
# L2 / fabric-interface counters of the steps of tools/lab/spmv_steps_lab.cpp (one counter group per pass, the program
# directly behind `--`): what changes when the product's y is STORED?   usage (GPU box, repo root): bash tools/lab/spmv_steps_pmc.sh <outdir>
OUT=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
hipcc -O3 --offload-arch=gfx950 -o /tmp/spmv_steps_lab $R/tools/lab/spmv_steps_lab.cpp 2>/dev/null || exit 1
/tmp/spmv_steps_lab 148 2,4,12,14,15 20 > $R/$OUT/times.txt 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for G in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
         "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_sum" \
         "TCC_TAG_STALL_sum TCC_BUSY_avr TCC_CYCLE_sum" \
         "TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum TCC_WRITE_sum TCC_NORMAL_EVICT_sum" \
         "TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum TCC_IB_STALL_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
         "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/$OUT/g$i -o pmc -- /tmp/spmv_steps_lab 148 2,4,12,14,15 3 > $R/$OUT/run_g$i.txt 2>&1
  echo "group $i [$G] rc=$?"
done
cd $R
python3 tools/lab/spmv_steps_pmc_summary.py $OUT | tee $OUT/summary.txt
