python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "compose or transform or resize" 2>&1 | tail -2
python tools/kernel_bench.py 2>&1 | grep "compose\|transform\|resize"
PH_COMPOSE_QUAD=1 python tools/kernel_bench.py 2>&1 | grep "compose"
python tools/config_bench.py 2>&1 | tail -4
