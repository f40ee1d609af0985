for wl in C2 C3 C4; do KEEP_ACT=0 python tools/probe_flow.py $wl 2>&1 | tail -1; done
timeout -k 10 600 python -m pytest tests/test_gpu_flow.py tests/test_gpu_golden.py tests/test_gpu_large_configs.py -m gpu -q -x 2>&1 | tail -3
