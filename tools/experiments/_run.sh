set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_head.py -x -q -m gpu -k "not subprocess" 2>&1 | tail -25 > gpurun_out/r02a_head_tests.txt
tail -5 gpurun_out/r02a_head_tests.txt
timeout 600 python tools/experiments/head_rt_ab.py rowtile > gpurun_out/r02a_ab_rowtile.jsonl 2> gpurun_out/r02a_ab_rowtile.err
MTR_HEAD_F32=groups HEAD_AB_ONLY_FUSED=1 timeout 600 python tools/experiments/head_rt_ab.py groups > gpurun_out/r02a_ab_groups.jsonl 2> gpurun_out/r02a_ab_groups.err
MTR_HEAD_RTG=5 HEAD_AB_ONLY_FUSED=1 timeout 600 python tools/experiments/head_rt_ab.py rtg5 > gpurun_out/r02a_ab_rtg5.jsonl 2>&1
MTR_HEAD_RTG=2 HEAD_AB_ONLY_FUSED=1 timeout 600 python tools/experiments/head_rt_ab.py rtg2 > gpurun_out/r02a_ab_rtg2.jsonl 2>&1
cat gpurun_out/r02a_ab_*.jsonl | cut -c1-400
