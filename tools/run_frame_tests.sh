set -e
mkdir -p gpurun_out/golden
python -m pytest tests/test_gpu_reference.py -x -q -s -k "whole_frame or every_bounce" > gpurun_out/frame_tests.log 2>&1 || { tail -40 gpurun_out/frame_tests.log; exit 1; }
tail -30 gpurun_out/frame_tests.log
python tests/golden/make_golden.py gpurun_out/golden frames > gpurun_out/golden_frames.log 2>&1 || { tail -30 gpurun_out/golden_frames.log; exit 1; }
tail -10 gpurun_out/golden_frames.log
