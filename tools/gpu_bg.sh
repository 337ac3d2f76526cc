#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_windows.py tests/test_gpu_cli.py tests/test_gpu_synth.py tests/test_gpu_baseline_configs.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
timeout 500 python tools/bg_probe.py 2>&1 | tail -2
