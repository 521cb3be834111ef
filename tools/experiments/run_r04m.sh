( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -70 ) > gpurun_out/r04m_tests.log 2>&1
