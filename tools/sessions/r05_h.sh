#!/bin/bash
# Round 5, session h: the tests that failed in session g (sharded refinement block, device-memory growth: alone and inside
# its file), then the files around them.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05h; mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python3 -m pytest tests/test_gpu_memory.py -q -m gpu -s > $O/pytest_memory_alone.txt 2>&1
echo "rc $?" >> $O/pytest_memory_alone.txt
timeout 1800 python3 -m pytest tests/test_gpu_sharded.py tests/test_gpu_formats.py tests/test_gpu_memory.py tests/test_gpu_multi.py -q -m gpu -s > $O/pytest_subset.txt 2>&1
echo "rc $?" >> $O/pytest_subset.txt
echo done > $O/done.txt
