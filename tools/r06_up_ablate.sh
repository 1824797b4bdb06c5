# the block compositor by layer count: product, arithmetic alone (upab1: no patch loads), loads + stores alone (upab11) - timing builds, wrong pixels
for n in 1 2 3 4; do
for v in "" upab1 upab11; do
  lib=""; [ -n "$v" ] && lib=tools/_variants/libphaneron_hip_$v.so
  echo "${v:-product} layers=$n: $(PH_UP_LAYERS=$n PHANERON_HIP_LIB=$lib python tools/up_bench.py 300 up 2>/dev/null | tail -1)"
done
done
