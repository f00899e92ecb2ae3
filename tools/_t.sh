python -m pytest tests/test_gpu_stages.py tests/test_gpu_packed.py tests/test_gpu_longform.py tests/test_gpu_configs.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
python bench.py --cpu-sample 0 --no-host-loop --no-b1 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('ms', d['ms_per_step'], 'vo stage', d['stage_ms_fully_profiled_step']['vo'])"
