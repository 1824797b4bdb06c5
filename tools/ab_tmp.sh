python -m pytest tests/test_hip_parity.py tests/test_boundary_gpu.py -m gpu -x -q 2>&1 | tail -2
for g in 4 6 8 4 6 8; do
  export PH_FUSED_GEOM=$g
  echo "== P $g"; python bench.py --steps 1000 --warmup 100 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"
done
