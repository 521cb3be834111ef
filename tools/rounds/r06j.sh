# f32 head at configs[2]'s shape (B 32, 12x12, J 17): FETCH_SIZE of NCHW vs NHWC features -- is the 2.08 x "traffic" the access
# pattern or the x 2 correction (calibrated for 16 B / lane reads) applied to 4 B / lane NCHW copies?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for lay in nchw nhwc; do
  for B in 32 256; do
  timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/f_$lay$B -o p -- python $R/tools/_pmc_head.py $B f32 17 12 $lay > /dev/null 2>&1
  python $R/tools/rocprof_summary.py /tmp/f_$lay$B $O/r06j_fetch_f32_12x12_b${B}_$lay.md --ours-only > /dev/null 2>&1
  echo "== $lay B=$B"; grep -E "head_rt" $O/r06j_fetch_f32_12x12_b${B}_$lay.md | cut -c1-70,150-240
  done
done
