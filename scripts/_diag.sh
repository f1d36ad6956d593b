set -euo pipefail
timeout -k 10 900 python -m pytest tests/test_match_pose_gpu.py -m gpu -x -q > gpurun_out/prune_test.log 2>&1
timeout -k 10 300 python scripts/match_load_bench.py 2>/dev/null | tail -3
