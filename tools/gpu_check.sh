#!/bin/bash
# one GPU-box visit: the tests named on the command line (default: the convolution / fold / model tests), then the default
# bench line.  Everything lands under gpurun_out/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
TESTS="${@:-tests/test_conv_gpu.py tests/test_conv_fold_gpu.py tests/test_bn_fold_gpu.py tests/test_model_gpu.py}"
timeout 1500 python -m pytest $TESTS -q -m gpu -x --no-header -p no:cacheprovider > gpurun_out/tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/tests.log
tail -40 gpurun_out/tests.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench rc=$?"
cat gpurun_out/bench.json
tail -5 gpurun_out/bench.err
