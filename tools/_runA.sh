python -m pytest tests -q -m gpu > gpurun_out/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED" gpurun_out/gpu_tests.log | tail -5
for i in 1 2 3; do python bench.py --workload planar7_1024x32 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('planar7', round(d['value']), min(d['rep_ms_per_step']))"; done
