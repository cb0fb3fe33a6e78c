TAG=${1:-t}
mkdir -p gpurun_out/$TAG
(time python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/$TAG/tests.log 2>&1
grep -n "passed\|failed" gpurun_out/$TAG/tests.log; grep -n "^FAILED\|^ERROR" gpurun_out/$TAG/tests.log
