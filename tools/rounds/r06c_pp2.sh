# Round 6, third GPU call: the two-halves head with software-pipelined fragment reads; the same pipelining in the shipped
# early-copies kernel (a developer build); the NHWC decode with the next batch prefetched.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_head.py -x -q -m gpu -k "two_halves or equals_unfused" > $O/r06c_pp_tests.log 2>&1; echo "pp tests rc=$?: $(tail -1 $O/r06c_pp_tests.log)" | tee -a $O/r06c_summary.txt
timeout 900 python tools/experiments/head16_pp_probe.py quick > $O/r06c_head16_pp.jsonl 2> $O/r06c_head16_pp.err; echo "pp probe rc=$?" | tee -a $O/r06c_summary.txt
MTR_PROBE_LIB=$R/tools/experiments/_build/libmtr_h16pipe.so timeout 900 python tools/experiments/head16_pp_probe.py quick > $O/r06c_head16_fragpipe_in_early_copies.jsonl 2>> $O/r06c_head16_pp.err; echo "fragpipe probe rc=$?" | tee -a $O/r06c_summary.txt
timeout 900 python tools/experiments/nhwc_decode_ab.py run > $O/r06c_nhwc_decode_ab.jsonl 2> $O/r06c_nhwc.err; echo "nhwc ab rc=$?" | tee -a $O/r06c_summary.txt
timeout 300 python -m pytest tests/test_gpu_decode_recon.py -q -m gpu > $O/r06c_decode_tests.log 2>&1; echo "decode tests rc=$?: $(tail -1 $O/r06c_decode_tests.log)" | tee -a $O/r06c_summary.txt
for f in r06c_head16_pp.jsonl r06c_head16_fragpipe_in_early_copies.jsonl; do echo $f; cat $O/$f | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'][0], d['shape'][2], 'nhwc' if d['nhwc'] else 'nchw', d['opts'], d['us'], d['bit_equal_to_early_copies'])"; done
cat $O/r06c_nhwc_decode_ab.jsonl | cut -c1-200
cat $O/r06c_summary.txt
