#!/bin/bash
# round 5: transition latency, the odd-pitch interpolate test, the comm tests
set -o pipefail
mkdir -p gpurun_out/r5_misc
timeout -k 10 300 python tools/transition_latency.py > gpurun_out/r5_misc/transition.txt 2> gpurun_out/r5_misc/transition.err
echo "transition rc $?"; cat gpurun_out/r5_misc/transition.txt; tail -n 3 gpurun_out/r5_misc/transition.err
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "only_4_byte_aligned or row_pitch" > gpurun_out/r5_misc/tests.log 2>&1
echo "tests rc $?"; tail -n 4 gpurun_out/r5_misc/tests.log
