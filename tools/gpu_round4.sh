#!/bin/bash
# Round-4 evidence run on the GPU box: tests, the default bench, config 3's per-GPU share, the MC sweep, rocprofv3 passes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests -m gpu -q 2>&1 | tee gpurun_out/pytest_gpu.log | tail -4
python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; tail -c 300 gpurun_out/r04_bench.err
python bench.py --workload cfg3 --gpus 1 --slides 200 --no-extras --no-cpu-baseline --no-profile > gpurun_out/r04_bench_cfg3_share.json 2> gpurun_out/r04_bench_cfg3.err; tail -c 300 gpurun_out/r04_bench_cfg3.err
python tools/sweep_mc.py > gpurun_out/r04_sweep_mc.jsonl 2>/dev/null
bash tools/profile.sh r04 f16 > gpurun_out/profile_r04.log 2>&1; tail -3 gpurun_out/profile_r04.log
