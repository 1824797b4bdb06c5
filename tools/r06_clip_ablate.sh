# phase prices of clip_up_write_v210_kernel (timing builds, wrong pixels): 1 = phase 1 alone, 2 = nothing converted, 4 = no phase 2 work, 6 = tables + barriers only
for v in "" clip1 clip2 clip4 clip6; do
 for shape in "1280 720 1920 1080" "1920 1080 1920 1080" "1920 1080 3840 2160"; do
  lib=""; [ -n "$v" ] && lib=tools/_variants/libphaneron_hip_$v.so
  echo "$v $shape: $(PHANERON_HIP_LIB=$lib PH_ENLARGE_ONLY=routed PH_ENLARGE_FORMAT=yuv420p python tools/enlarge_bench.py 400 1 $shape 2>/dev/null | python -c 'import json,sys; print(json.loads(sys.stdin.read())["us_per_frame"])')"
 done
done
