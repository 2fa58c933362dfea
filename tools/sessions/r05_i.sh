#!/bin/bash
# Round 5, session i: the last tree once more -- smoke, the whole GPU suite, the driver's bench command, the console driver.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05i; mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
python3 __graft_entry__.py smoke > $O/smoke.txt 2>&1
timeout 3000 python3 -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
echo "rc $?" >> $O/pytest_gpu.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_err.txt
for i in a b; do timeout 600 python3 tools/cli_scale.py 148 > $O/cli_scale_n148_$i.txt 2>&1; done
echo done > $O/done.txt
