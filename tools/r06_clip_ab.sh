mkdir -p gpurun_out/r06
python -m pytest tests/test_routes_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -5
python -m pytest tests/test_chan_gpu.py -x -q -m gpu -k "enlarged or planar_clips_at or finished_images or decoder" 2>&1 | tail -5
for f in yuv420p yuv422p10 nv12; do
 PH_ENLARGE_FORMAT=$f python tools/enlarge_bench.py 400 1 1280 720 1920 1080 2>/dev/null
 PH_ENLARGE_FORMAT=$f python tools/enlarge_bench.py 400 1 1920 1080 1920 1080 2>/dev/null
done
PH_ENLARGE_FORMAT=yuv420p python tools/enlarge_bench.py 300 1 1920 1080 3840 2160 2>/dev/null
PH_ENLARGE_FORMAT=yuv420p python tools/enlarge_bench.py 300 1 720 576 1920 1080 2>/dev/null
