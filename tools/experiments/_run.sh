cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_head.py -x -q -m gpu 2>&1 | tail -5
RT_VARIANTS=full,nbuf2,nbuf4,nodecode python tools/experiments/ablate_rt.py run 2>/dev/null
timeout 600 python tools/experiments/head_rt_ab.py rowtile4 2>/dev/null | cut -c1-400 | tee gpurun_out/r02f_head_rt_ab.jsonl
