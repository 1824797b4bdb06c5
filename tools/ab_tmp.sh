python -m pytest tests -m gpu -x -q 2>&1 | tail -2
echo "== nontemporal"; python tools/kernel_bench.py 2>&1 | grep '^{' | cut -c1-110
echo "== plain"; PHANERON_HIP_LIB=/root/repo/phaneron_amd/lib/libphaneron_hip_plain.so python tools/kernel_bench.py 2>&1 | grep '^{' | cut -c1-110
echo "== nontemporal"; python tools/config_bench.py 2>&1 | grep '^{' | cut -c1-150
echo "== plain"; PHANERON_HIP_LIB=/root/repo/phaneron_amd/lib/libphaneron_hip_plain.so python tools/config_bench.py 2>&1 | grep '^{' | cut -c1-150
