for n in 4 2; do
for v in "" pipe; do
  lib=""; [ -n "$v" ] && lib=tools/_variants/libphaneron_hip_$v.so
  echo "${v:-product} layers=$n: $(PH_UP_LAYERS=$n PHANERON_HIP_LIB=$lib python tools/up_bench.py 300 up 2>/dev/null | tail -1)"
done
done
