#!/bin/bash
# Round 5, session c: whole GPU suite on the restructured CG (after the rank-agreement fix), k_recover's new form
# timed and counted, refine modes again (fp64 refresh every 50), 400^3 mixed honest.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 300 python3 tools/recover_time.py 148 10 > $O/recover_time_n148.json 2> $O/recover_time.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/recover_trace -o run -- python3 $R/tools/recover_time.py 148 10 > $O/recover_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/recover_pmc/$C -o pmc -- python3 $R/tools/recover_time.py 148 3 > $O/recover_pmc_$C.log 2>&1
done
cd $R
python3 tools/pmc_summary.py gpurun_out/r05c/recover_pmc > $O/recover_pmc_summary.txt 2>&1
python3 tools/trace_summary.py $O/recover_trace/run_kernel_trace.csv > $O/recover_trace_summary.txt 2>&1
timeout 900 python3 tools/mixed_refine.py 148 > $O/mixed_refine_n148.jsonl 2> $O/mixed_refine.err
timeout 3000 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1
echo "rc $?" >> $O/pytest_gpu.txt
echo done > $O/done.txt
