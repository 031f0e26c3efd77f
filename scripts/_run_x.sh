cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/full_gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/full_gpu_tests.log
python bench.py > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err; python -c "
import json; d=json.loads(open('gpurun_out/bench_now.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], d['serial_reference']['ms_per_step'], r['frac'], r['traffic'], d['cpu_baseline']['value'])"
python bench.py --workload detect --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('detect', d['value'], d['ms_per_step'], r['frac'], r['achieved'])"
