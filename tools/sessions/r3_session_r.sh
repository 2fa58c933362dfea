#!/bin/bash
# round 3, session r: cache-hierarchy counters of the SpMV, whole box against perforated box
R=$GRAFT_REPO_ROOT
cd $R
timeout 2400 bash tools/gather_pmc.sh gpurun_out/r03_r/gather_pmc 2>&1 | tail -60
