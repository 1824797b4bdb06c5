python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
for v in prev new prev new; do
  if [ $v = new ]; then unset PHANERON_HIP_LIB; else export PHANERON_HIP_LIB=/root/repo/phaneron_amd/lib/libphaneron_hip_$v.so; fi
  echo "== $v"; python bench.py --steps 1000 --warmup 100 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"
done
