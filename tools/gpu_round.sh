#!/bin/bash
# A round's evidence run on the GPU box (bash tools/gpu_round.sh r06): tests, the default bench, config 3's per-GPU share, the MC
# sweep, rocprofv3 passes (tools/profile.sh), SQ counters (tools/pmc_sq.sh); tools/collect.sh copies the results into profiles/.
T=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests -m gpu -q 2>&1 | tee gpurun_out/pytest_gpu.log | tail -4
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; tail -c 300 gpurun_out/${T}_bench.err
python bench.py --workload cfg3 --gpus 1 --slides 200 --no-extras --no-cpu-baseline --no-profile > gpurun_out/${T}_bench_cfg3_share.json 2> gpurun_out/${T}_bench_cfg3.err; tail -c 300 gpurun_out/${T}_bench_cfg3.err
python tools/sweep_mc.py > gpurun_out/${T}_sweep_mc.jsonl 2>/dev/null
bash tools/profile.sh $T f16 > gpurun_out/profile_${T}.log 2>&1; tail -3 gpurun_out/profile_${T}.log
bash tools/pmc_sq.sh > gpurun_out/pmc_sq_${T}.log 2>&1; tail -2 gpurun_out/pmc_sq_${T}.log
