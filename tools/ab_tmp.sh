for v in old new old new; do
  if [ $v = new ]; then unset PHANERON_HIP_LIB; else export PHANERON_HIP_LIB=/root/repo/phaneron_amd/lib/libphaneron_hip_$v.so; fi
  echo "== $v"; python bench.py --steps 300 --warmup 30 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"
done
