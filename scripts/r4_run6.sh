#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_spconv_gpu.py -m gpu -x -q -k "output_stationary or debug_pipeline" > $O/r4_gputest6.log 2>&1 || { tail -40 $O/r4_gputest6.log; exit 1; }
tail -2 $O/r4_gputest6.log
FRAMES=12 ONLY=b3,f2tr,f3tr timeout -k 10 300 python scripts/layer_bench.py > $O/r4_layers_wreg1.log 2>&1 || { tail -20 $O/r4_layers_wreg1.log; exit 2; }
cat $O/r4_layers_wreg1.log | cut -c1-400
APR_OS_WREG=0 FRAMES=12 ONLY=b3,f2tr,f3tr timeout -k 10 300 python scripts/layer_bench.py > $O/r4_layers_wreg0.log 2>&1 || exit 3
cat $O/r4_layers_wreg0.log | cut -c1-400
FRAMES=2 ONLY=b3,f2tr,f3tr timeout -k 10 300 python scripts/layer_bench.py > $O/r4_layers_wreg1_f2.log 2>&1 || exit 4
cat $O/r4_layers_wreg1_f2.log | cut -c1-400
